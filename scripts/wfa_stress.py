"""Randomised parity run of hp_wfa_assign_batch against the CPU oracle (same comparison as tests/test_wfa_gpu.py,
wider parameter spread, longer): read lengths 200 b - 20 kb, noise up to 6 %, small prune distances and edit caps
(pruning floor, MAX_ED and band re-runs), multi-allelic sites, and batches whose jobs share ONE reference buffer with
different windows (the merged-range upload path).   usage: wfa_stress.py [seed] [seconds]"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from hiphase_amd import _ffi
from hiphase_amd.wfa_graph import WfaJobSpec, make_jobs, wfa_assign_batch
from oracle_ffi import oracle
from wfa_util import synth_wfa_job, _Rng

os.environ.setdefault("HP_WFA2_MIN_JOBS", "0")   # the compact kernel takes every batch (its leftovers still reach the dense-band one)
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 120.0
r = _Rng(seed)
d = oracle()
t0 = time.time()
n_cmp = n_maxed = n_batches = 0
while time.time() - t0 < budget:
    n_jobs = r.randint(1, 96)
    prune = [0, 20, 100, 500][r.randint(0, 3)]
    max_ed = [8, 60, 150, 500, 2000][r.randint(0, 4)]
    specs = []
    shared = r.u01() < 0.4
    if shared:   # one reference buffer, per-read windows and variant subsets (what read_parsing.rs:738-768 builds)
        L = r.randint(3000, 12000)
        base, _ = synth_wfa_job(r.next(), ref_len=L, n_vars=r.randint(4, 30), n_homs=r.randint(0, 6), noise=0.0, multiallelic=0.2)
        for _ in range(n_jobs):
            a = r.randint(0, L - 700)
            b = min(L, a + r.randint(400, 4000))
            read = bytearray(base.reference[a:b])
            for k in range(len(read)):   # noise on the window itself
                if r.u01() < 0.01:
                    read[k] = b"ACGT"[r.next() & 3]
            for v in base.hets + base.homs:   # carry some alternate alleles (same-length ones keep the coordinates simple)
                if a <= v.position and v.position + v.ref_len <= b and len(v.allele1) == v.ref_len and r.u01() < 0.5:
                    read[v.position - a:v.position - a + v.ref_len] = v.allele1
            specs.append(WfaJobSpec(reference=base.reference, ref_start=a, ref_end=b, hets=base.hets, homs=base.homs,
                                    read=bytes(read), ref_base=0))
    else:
        for _ in range(n_jobs):
            L = [200, 600, 2000, 6000, 20000][r.randint(0, 4)]
            if L == 20000 and r.u01() < 0.7:
                L = 3000
            specs.append(synth_wfa_job(r.next(), ref_len=max(L, 800), n_vars=r.randint(0, 24), n_homs=r.randint(0, 6),
                                       noise=[0.0, 0.002, 0.01, 0.03, 0.06][r.randint(0, 4)], multiallelic=0.3)[0])
    got = wfa_assign_batch(specs, prune_distance=prune, max_edit_distance=max_ed)
    for i, (spec, g) in enumerate(zip(specs, got)):
        jobs, keep = make_jobs([spec])
        out = _ffi.WfaResult()
        al = np.full(max(1, len(spec.hets)), 3, np.uint8)
        rc = d.hpo_wfa_assign(C.byref(jobs[0]), (2 ** 64 - 1) if prune == 0 else prune, max_ed, C.byref(out), al.ctypes.data)
        assert rc == 0
        exp = (out.status, out.score, out.n_nodes)
        if (g[0], g[1], g[2]) != exp or not np.array_equal(g[3], al[:len(spec.hets)]):
            print("MISMATCH", seed, n_batches, i, g[:3], exp, g[3].tolist(), al[:len(spec.hets)].tolist())
            sys.exit(1)
        n_cmp += 1
        n_maxed += out.status != 0
    n_batches += 1
print(f"wfa_stress seed {seed}: {n_batches} batches, {n_cmp} reads compared, {n_maxed} MAX_ED, 0 mismatches, {time.time() - t0:.0f}s")
