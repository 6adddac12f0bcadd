"""`.hpbk` capture/replay round trip (CPU): what a patched HiPhase would dump is what the solver is fed."""
import io

import numpy as np

from hiphase_amd.block_io import read_blocks, write_block
from oracle_ffi import oracle_solve, oracle_synth


def test_round_trip_and_replay():
    blocks = [oracle_synth(n, 10, 6, 0.05, 0.03, 77 + i)[0] for i, n in enumerate((40, 1, 17))]
    buf = io.BytesIO()
    exp = []
    for i, b in enumerate(blocks):
        h1, h2, st, _ = oracle_solve(b)
        exp.append((h1, h2, st))
        write_block(buf, b, block_index=i, expected=(h1, h2, st) if i != 1 else None)
    buf.seek(0)
    got = list(read_blocks(buf))
    assert len(got) == 3
    for i, (blk, meta, e) in enumerate(got):
        assert meta["block_index"] == i and meta["min_queue_size"] == 1000
        for f in ("read_start", "read_end", "row_off", "var_flags"):
            assert np.array_equal(getattr(blk, f), getattr(blocks[i], f))
        assert np.array_equal(blk.quals[:blk.n_cells], blocks[i].quals[:blk.n_cells])
        h1, h2, st, _ = oracle_solve(blk)   # replay
        assert np.array_equal(h1, exp[i][0]) and st == exp[i][2]
        if i != 1:
            assert np.array_equal(e[0], exp[i][0]) and tuple(e[2]) == exp[i][2]
        else:
            assert e is None


def test_c_writer_produces_the_same_stream(tmp_path):
    """hp_hpbk_append (the capture side a patched HiPhase links) == hiphase_amd.block_io.write_block"""
    import ctypes as C
    from hiphase_amd import _ffi
    lib = _ffi.lib()
    blocks = [oracle_synth(n, 10, 6, 0.05, 0.03, 91 + i)[0] for i, n in enumerate((23, 1, 40))]
    path = tmp_path / "cap.hpbk"
    buf = io.BytesIO()
    for i, b in enumerate(blocks):
        h1, h2, st, _ = oracle_solve(b)
        exp = (h1, h2, st) if i != 1 else None
        write_block(buf, b, block_index=i, expected=exp)
        v = b.view()
        p = _ffi.AstarParams(1000, 3, 0, i)
        stc = _ffi.PhaseStats(*st)
        rc = lib.hp_hpbk_append(str(path).encode(), C.byref(v), C.byref(p), h1.ctypes.data if exp else None, h2.ctypes.data if exp else None,
                                C.byref(stc) if exp else None)
        assert rc == 0
    assert path.read_bytes() == buf.getvalue()
