"""`.hpbk` capture -> replay on the HIP path: blocks written with their expected results (here by the
oracle standing in for the real HiPhase binary, through the C writer a patched HiPhase links) are read back, solved by
hp_batch_* and by `bench.py --replay`, and compared with what the file says."""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from hiphase_amd import ResidentBatch, _ffi
from hiphase_amd.block_io import read_blocks
from oracle_ffi import oracle_solve, oracle_synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def capture(path, sizes):
    lib = _ffi.lib()
    for i, n in enumerate(sizes):
        b = oracle_synth(n, 30, 20, 0.02 + 0.02 * (i % 3), 0.02, 300 + i)[0]
        h1, h2, st, _ = oracle_solve(b)
        v, p, stc = b.view(), _ffi.AstarParams(1000, 3, 0, i), _ffi.PhaseStats(*st)
        assert lib.hp_hpbk_append(str(path).encode(), C.byref(v), C.byref(p), h1.ctypes.data, h2.ctypes.data, C.byref(stc)) == 0


def test_replay_through_the_resident_batch(tmp_path):
    path = tmp_path / "blocks.hpbk"
    capture(path, [12, 75, 300, 2, 40])
    with open(path, "rb") as f:
        allb = list(read_blocks(f))
    assert len(allb) == 5
    rb = ResidentBatch([b for b, _, _ in allb])
    rb.solve()
    res, _, _ = rb.results()
    rb.close()
    for r, (_, meta, exp) in zip(res, allb):
        assert np.array_equal(r.haplotype_1, exp[0]) and np.array_equal(r.haplotype_2, exp[1]) and r.statistics.as_tuple() == tuple(exp[2])


def test_bench_replay_mode(tmp_path):
    path = tmp_path / "blocks.hpbk"
    capture(path, [30, 120, 8, 55])
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "c2", "--replay", str(path), "--steps", "1", "--warmup", "1", "--no-cpu"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["parity_vs_capture"] == {"blocks_compared": 4, "bit_identical": True}
