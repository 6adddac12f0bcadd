"""Extended randomised parity run (not part of the test suite): many random blocks x queue parameters x switches against
the CPU oracle — haplotypes, statistics, heuristic arrays and work counters must all be identical."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from hiphase_amd import ResidentBatch, synth_block
from hiphase_amd._ffi import HpError
from oracle_ffi import oracle_solve

seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 1
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 12
cmax = int(sys.argv[3]) if len(sys.argv) > 3 else 130   # <= 40: every batch runs the one-tile kernel variant
rng = np.random.default_rng(seed0)
bad = 0; total = 0
for rd in range(rounds):
    minq, qinc = [(1000, 3), (200, 2), (60, 1), (25, 1), (400, 0), (2000, 5)][rd % 6]
    blocks, expect = [], []
    for i in range(48):
        n = int(rng.integers(1, 700)); c = int(rng.integers(2, cmax)); s = int(rng.integers(2, 140))
        e = float(rng.choice([0.0, 0.01, 0.05, 0.15, 0.3])); a = float(rng.choice([0.0, 0.02, 0.1]))
        ign = int(rng.choice([0, 0, 40, 150]))
        blk, _ = synth_block(n, c, s, e, a, seed0 * 100000 + rd * 1000 + i, ignored_permille=ign)
        try:
            exp = oracle_solve(blk, min_queue_size=minq, queue_increment=qinc, want_heuristics=True)
        except HpError:
            continue
        blocks.append(blk); expect.append(exp)
    for env in ({}, {"HP_SEG_TARGET": "64", "HP_SEG_WARM": "160"}, {"HP_NO_SEGMENTS": "1"}):
        for k in ("HP_SEG_TARGET", "HP_SEG_WARM", "HP_NO_SEGMENTS"): os.environ.pop(k, None)
        os.environ.update(env)
        rb = ResidentBatch(blocks, min_queue_size=minq, queue_increment=qinc); rb.solve()
        res, ctrs, heur = rb.results(want_heuristics=True); rb.close()
        for k, (r, c, h, (h1, h2, st, octr, oh)) in enumerate(zip(res, ctrs, heur, expect)):
            total += 1
            ok = (np.array_equal(r.haplotype_1, h1) and np.array_equal(r.haplotype_2, h2) and r.statistics.as_tuple() == st
                  and np.array_equal(h, oh) and c.as_tuple() == octr)
            if not ok:
                bad += 1
                print("MISMATCH round", rd, "block", k, env, "n", blocks[k].n_variants, flush=True)
print(f"seed {seed0}: {total} comparisons, {bad} mismatches")
