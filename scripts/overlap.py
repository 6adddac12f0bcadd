"""Overlap of the stages on the device, from a rocprofv3 kernel + memory-copy trace (scripts/prof_path.sh):
python scripts/overlap.py gpurun_out/prof_path_<tag>/trace > profiles/roundN/path_overlap.txt
Classes: wfa = the hp_wfa3_kernel (or hp_wfa2_kernel) launch sets, dense = dense-band + reference-window kernels of the late results,
astar = hp_astar_kernel + hp_heur_*, h2d = host-to-device copies, other = everything else."""
import csv, glob, sys

d = sys.argv[1]
iv = {k: [] for k in ("wfa", "dense", "astar", "h2d", "other")}
for f in glob.glob(d + "/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        k = ("wfa" if ("hp_wfa2_kernel<" in n or "hp_wfa3_kernel<" in n) else "dense" if ("hp_wfa_kernel" in n or "hp_wfa_big_kernel" in n or "hp_wfa2_bound_kernel" in n)
             else "astar" if ("hp_astar_kernel" in n or "hp_heur_" in n) else "other")
        iv[k].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
for f in glob.glob(d + "/*memory_copy_trace.csv"):
    for r in csv.DictReader(open(f)):
        if "HOST_TO_DEVICE" in r["Direction"]: iv["h2d"].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))

def union(v):
    out = []
    for a, b in sorted(v):
        if out and a <= out[-1][1]: out[-1][1] = max(out[-1][1], b)
        else: out.append([a, b])
    return out
def total(u): return sum(b - a for a, b in u)
def inter(u, v):
    i = j = 0; t = 0
    while i < len(u) and j < len(v):
        a, b = max(u[i][0], v[j][0]), min(u[i][1], v[j][1])
        if a < b: t += b - a
        if u[i][1] < v[j][1]: i += 1
        else: j += 1
    return t
U = {k: union(v) for k, v in iv.items()}
print("overlap of the stages on the device, from the rocprofv3 kernel + memory-copy trace of `python bench.py --steps 5 --warmup 2` (scripts/prof_path.sh, scripts/overlap.py):")
for k in ("other", "wfa", "dense", "astar", "h2d"): print(f"{k:6s} busy {total(U[k]) / 1e6:8.1f} ms in {len(U[k])} intervals")
for a, b in (("wfa", "h2d"), ("wfa", "astar"), ("wfa", "dense"), ("astar", "h2d")):
    t = inter(U[a], U[b]); den = total(U[b]) or 1
    print(f"{a} || {b}: {t / 1e6:8.1f} ms overlapped = {100 * t / den:.0f} % of the {b} time")
