"""Randomised CPU pin of the round-2 graph-WFA design (tests/cpp/wfa2_model.cpp vs the oracle), wider and longer than
tests/test_wfa2_model.py.   usage: wfa2_model_stress.py [seed] [seconds] [generation: 2 | 3]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_wfa2_model import model, compare
from oracle_ffi import oracle
from wfa_util import synth_wfa_job, _Rng

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 120.0
gen = int(sys.argv[3]) if len(sys.argv) > 3 else 3
r = _Rng(seed)
m, d = model(), oracle()
t0 = time.time()
tot = [0, 0, 0]
while time.time() - t0 < budget:
    prune = [0, 20, 100, 500, 500, 500][r.randint(0, 5)]
    max_ed = [8, 60, 150, 500, 500, 2000][r.randint(0, 5)]
    specs = []
    for _ in range(16):
        L = [200, 600, 2000, 6000, 17000][r.randint(0, 4)]
        specs.append(synth_wfa_job(r.next(), ref_len=max(L, 800), n_vars=r.randint(0, 40), n_homs=r.randint(0, 12),
                                   noise=[0.0, 0.002, 0.004, 0.01, 0.03][r.randint(0, 4)], multiallelic=0.3)[0])
    p = compare(specs, prune, max_ed, m, d, gen=gen)
    tot = [a + b for a, b in zip(tot, p)]
print(f"wfa2_model_stress seed {seed}, generation {gen}: compact model == oracle on {tot[0]} jobs; {tot[1]} outgrew the compact state, {tot[2]} builder->host; {time.time()-t0:.0f}s")
