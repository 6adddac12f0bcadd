"""Does the heuristic chain 'forget' its start? (feasibility study for a speculative segment-parallel heuristic)
Compare delta[v] = H[v]-H[v+1] of the full chain with a chain started at b+L with a zero state."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle_ffi import oracle_solve, oracle_synth, oracle, params
from hiphase_amd.read_segments import BlockMatrix, ReadSegment
import ctypes as C

def heur(blk):
    d = oracle(); v = blk.view(); p = params()
    h = np.zeros(blk.n_variants + 1, np.uint64)
    assert d.hpo_astar_heuristic(C.byref(v), C.byref(p), h.ctypes.data) == 0
    return h.astype(np.int64)

def truncate(blk, n):   # keep variants [0, n): rows clipped
    segs = []
    for s in blk.segments():
        if s.start >= n: continue
        e = min(s.end, n)
        al = [3] * n; q = [0] * n
        al[s.start:e] = s.alleles[:e - s.start]; q[s.start:e] = s.quals[:e - s.start]
        segs.append(ReadSegment(s.read_name, al, q))
    return BlockMatrix.from_segments(segs, n, blk.var_flags[:n])

for (N, Cc, S, e) in [(1200, 30, 20, 0.01), (1200, 30, 20, 0.15), (1200, 60, 40, 0.05)]:
    blk, _ = oracle_synth(N, Cc, S, e, 0.02, 99)
    H = heur(blk); dfull = H[:-1] - H[1:]
    for L in (40, 80, 120, 200):
        ok = tot = 0
        for b in range(200, N - 250, 97):
            t = truncate(blk, b + L)
            Ht = heur(t); dt = Ht[:-1] - Ht[1:]
            # state at b: deltas b+1..b+39 must match for the segment below to be exact
            tot += 1
            ok += int(np.array_equal(dt[b + 1:b + 40], dfull[b + 1:b + 40]) and np.array_equal(dt[:b + 1], dfull[:b + 1]))
        print(f"N={N} C={Cc} S={S} e={e} L={L}: {ok}/{tot} seams converge exactly")
