"""Local re-alignment micro-benchmark: hp_local_realign_batch (host coordinate logic on threads + one HIP edit-distance
launch for every inexact allele of the batch) vs the CPU oracle's per-record restatement, on synthetic CIGAR reads."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from hiphase_amd import _ffi
from hiphase_amd.read_parsing import local_realignment_batch
from local_util import make_local_block, oracle_local
import oracle_ffi

ref, variants, truth, records = make_local_block(7, ref_len=400000, n_vars=3000, n_reads=3000, read_len=(8000, 20000), noise=0.01)
local_realignment_batch(records[:64], variants)   # warm-up
t0 = time.perf_counter(); al, ql, st = local_realignment_batch(records, variants); dt = time.perf_counter() - t0
kms = _ffi.lib().hp_last_kernel_ms()
inexact = sum(sum(s.inexact_matches) + sum(s.failed_matches) for s in st)
d = oracle_ffi.oracle()
sample = records[:100]
t0 = time.perf_counter(); oal, oql, ost, rcs = oracle_local(d, sample, variants); dto = time.perf_counter() - t0
ok = bool(np.array_equal(al[:100], oal) and np.array_equal(ql[:100], oql))
print(json.dumps({"reads": len(records), "variants": len(variants), "call_s": dt, "reads_per_s": len(records) / dt, "edit_kernel_ms": kms,
                  "inexact_or_failed_alleles": int(inexact), "cpu_oracle_reads_per_s": len(sample) / dto, "parity_first_100": ok}))
