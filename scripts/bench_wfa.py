"""WFA stage micro-benchmark: synthetic 17 kb reads over a 24-het + 8-hom window (SURVEY.md §8d 'WFA synthetic'),
hp_wfa_assign_batch (host graph build + upload + kernel) vs the CPU oracle on a sample."""
import argparse, ctypes as C, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from hiphase_amd import _ffi
from hiphase_amd.wfa_graph import wfa_assign_batch, make_jobs, PreparedWfaBatch
from wfa_util import synth_wfa_job

ap = argparse.ArgumentParser()
ap.add_argument("--jobs", type=int, default=2048)
ap.add_argument("--distinct", type=int, default=64)
ap.add_argument("--ref-len", type=int, default=17000)
ap.add_argument("--noise", type=float, default=0.004)
ap.add_argument("--cpu-sample", type=int, default=8)
ap.add_argument("--kernel", choices=["auto", "compact", "dense"], default="auto")
args = ap.parse_args()
if args.kernel != "auto":
    os.environ["HP_WFA2_MIN_JOBS"] = "0" if args.kernel == "compact" else "1000000000"
base = [synth_wfa_job(1000 + s, ref_len=args.ref_len, n_vars=24, n_homs=8, noise=args.noise)[0] for s in range(args.distinct)]
specs = [base[i % args.distinct] for i in range(args.jobs)]
wfa_assign_batch(specs[:64])  # warm-up (module load)
pb = PreparedWfaBatch(specs)
pb.run()  # steady state: a worker thread's staging buffers and band scratch are sized by its first block
t0 = time.perf_counter(); res = pb.run(); dt = time.perf_counter() - t0
kms = _ffi.lib().hp_last_kernel_ms()
bases = sum(len(s.read) for s in specs)
out = {"jobs": args.jobs, "c_call_s": pb.last_call_s, "python_call_s": dt, "kernel_ms": kms, "kernel_reads_per_s": args.jobs / (kms * 1e-3), "reads_per_s": args.jobs / pb.last_call_s, "read_bases_per_s": bases / pb.last_call_s,
       "mean_score": float(np.mean([r[1] for r in res])), "max_score": int(max(r[1] for r in res))}
import oracle_ffi
d = oracle_ffi.oracle()
t0 = time.perf_counter()
for s in specs[:args.cpu_sample]:
    jobs, keep = make_jobs([s]); o = _ffi.WfaResult(); al = np.zeros(64, np.uint8)
    d.hpo_wfa_assign(C.byref(jobs[0]), 500, 500, C.byref(o), al.ctypes.data)
cpu = (time.perf_counter() - t0) / args.cpu_sample
out["cpu_oracle_reads_per_s"] = 1.0 / cpu
print(json.dumps(out))
