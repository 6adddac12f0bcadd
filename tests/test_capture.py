"""`.hpbr` read-bearing capture (hiphase_amd/csrc/hp_capture.cpp): a generated block set written block by block with the oracle's
results as "what the caller's own solve_block produced", read back, and the oracle's whole path on the replayed inputs equals
what the file says - every field, ASCII and BAM 4-bit. No GPU (the generator, the capture code and hpo_solve_block are host code
carried by the test oracle too)."""
import ctypes as C

import pytest

from hiphase_amd import _ffi
from hiphase_amd.blocks import _params
from hiphase_amd.synth_sets import Capture, SynthSet, default_spec
from oracle_ffi import oracle

KW = dict(total_hets=260, max_block_hets=80, seed=21, noisy_fraction=0.03, supplementary_fraction=0.06, frac_snv=0.8, frac_sv=0.04)


def write_capture(d, sset, prm, path, with_expected=True):
    exp = sset.outputs()
    for b in range(sset.n):
        assert d.hpo_solve_block(C.byref(sset.inputs[b]), C.byref(prm), C.byref(exp.arr[b])) == 0
        assert d.hp_hpbr_append(str(path).encode(), C.byref(sset.inputs[b]), C.byref(prm), C.byref(exp.arr[b]) if with_expected else None) == 0, d.hp_hpbr_last_error()
    return exp


@pytest.mark.parametrize("fmt", [_ffi.SEQ_ASCII, _ffi.SEQ_BAM4])
def test_capture_round_trip(tmp_path, fmt):
    d = oracle()
    prm = _params(2, 1000, 3, None, True)
    s = SynthSet(default_spec(d, seq_format=fmt, **KW), d)
    path = tmp_path / "blocks.hpbr"
    exp = write_capture(d, s, prm, path)
    cap = Capture(path, d)
    assert cap.n == s.n and cap.info["hets"] == s.info["hets"] and cap.info["records"] == s.info["records"] and cap.info["read_bases"] == s.info["read_bases"]
    assert cap.params[0].max_edit_distance == prm.max_edit_distance and cap.params[0].astar.min_queue_size == 1000
    got = cap.outputs()
    for b in range(cap.n):
        assert cap.inputs[b].seq_format == fmt and cap.expected[b].status == 0
        assert d.hpo_solve_block(C.byref(cap.inputs[b]), C.byref(cap.params[b]), C.byref(got.arr[b])) == 0
        assert d.hp_block_output_equal(C.byref(cap.inputs[b]), C.byref(got.arr[b]), C.byref(cap.expected[b])) == 1     # replay == file
        assert d.hp_block_output_equal(C.byref(s.inputs[b]), C.byref(exp.arr[b]), C.byref(cap.expected[b])) == 1       # file == what was written


def test_capture_without_expected_and_bad_files(tmp_path):
    d = oracle()
    prm = _params(2, 1000, 3, None, True)
    s = SynthSet(default_spec(d, total_hets=40, max_block_hets=40, seed=3), d)
    path = tmp_path / "noexp.hpbr"
    write_capture(d, s, prm, path, with_expected=False)
    cap = Capture(path, d)
    assert cap.n == s.n and all(cap.expected[b].status == -2 ** 31 for b in range(cap.n))
    bad = tmp_path / "bad.hpbr"
    bad.write_bytes(path.read_bytes()[:300])
    with pytest.raises(_ffi.HpError):
        Capture(bad, d)
    with pytest.raises(_ffi.HpError):
        Capture(tmp_path / "missing.hpbr", d)
