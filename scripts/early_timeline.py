"""Timeline of the early pass's kernels against the class kernels, from a rocprofv3 kernel trace (scripts/prof_path.sh):
python scripts/early_timeline.py gpurun_out/prof_path_<tag>/trace"""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        tag = ("CLS " + n.split("hp_wfa3_kernel")[1][:10] if "hp_wfa3_kernel<" in n else "BOUND64" if "bound_kernel<64" in n else "BOUND256" if "bound_kernel<256" in n
               else "DENSE" if "hp_wfa_kernel" in n else "DENSEBIG" if "hp_wfa_big" in n else "ASTAR" if "hp_astar_kernel" in n else "SEG" if "hp_heur_seg" in n else "EDIT" if "hp_edit_kernel" in n else None)
        if tag: rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), tag, r.get("Grid_Size", r.get("Grid_Size_X", "")), r.get("LDS_Block_Size", ""), r.get("VGPR_Count", ""), r.get("Queue_Id", "")))
rows.sort()
t0 = rows[0][0]
for s, e, tag, g, lds, vg, q in rows:
    print(f"{(s - t0) / 1e6:9.2f} +{(e - s) / 1e6:7.2f} ms  {tag:16s} grid {g:>8s} lds {lds:>6s} vgpr {vg:>4s} queue {q}")
