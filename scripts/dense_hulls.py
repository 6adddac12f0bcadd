"""How wide are the hulls a one-read-per-wavefront kernel walks? For the reads that end in the dense band on the bench workload - the 5 %-noise
tail - the oracle counts, per (round, node) visit, the diagonals that kept a wave (oracle/hp_oracle_wfa.cpp, hpo_wfa_hull_stats): with
wfa_prune_distance = 500 a wave more than 500 read bases behind the front is dropped, and a diagonal k off the front's lags by ~20 |k| bases
at 5 % noise. CPU only.  python scripts/dense_hulls.py [reads] [noise]"""
import ctypes as C, sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from hiphase_amd import _ffi
from hiphase_amd.synth_sets import SynthSet, default_spec
from hiphase_amd.blocks import _params
from oracle_ffi import oracle

n_want = int(sys.argv[1]) if len(sys.argv) > 1 else 60
noise = float(sys.argv[2]) if len(sys.argv) > 2 else 0.05
lib, d = _ffi.lib(), oracle()
d.hpo_wfa_hull_stats.argtypes = [C.POINTER(C.c_uint64), C.c_int]
d.hpo_wfa_hull_stats.restype = None
prm = _params(2, 1000, 3, None, True)
s = SynthSet(default_spec(lib, total_hets=400, seed=5, seq_format=_ffi.SEQ_ASCII, max_block_hets=200, noisy_fraction=1.0, noisy_noise=noise, supplementary_fraction=0.0))
out = s.outputs()
st = (C.c_uint64 * 8)()
d.hpo_wfa_hull_stats(st, 1)
recs = 0
for b in range(s.n):
    if recs >= n_want:
        break
    assert d.hpo_solve_block(C.byref(s.inputs[b]), C.byref(prm), C.byref(out.arr[b])) == 0
    recs += s.inputs[b].n_records
d.hpo_wfa_hull_stats(st, 0)
v, sw, mw, c64, c256, rounds, mn, wide = [int(x) for x in st]
print(f"{recs} records at {noise:.3f} noise (max_edit_distance 500, prune 500): {rounds} rounds, {v} (round, node) visits = {v / max(rounds, 1):.2f} per round (most {mn}),"
      f" hull width mean {sw / max(v, 1):.1f} widest {mw}, visits wider than 64 diagonals {100.0 * wide / max(v, 1):.2f} %;"
      f" 64-lane chunks {c64} ({c64 / max(v, 1):.3f} per visit), 256-lane chunks {c256}: four wavefronts per read would save {100.0 * (c64 - c256) / max(c64, 1):.1f} % of the chunk steps")
