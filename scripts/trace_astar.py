"""A* launches of a rocprofv3 kernel trace (csv): name, start (ms from the first one shown), duration, grid - the last N of them."""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if any(k in r["Kernel_Name"] for k in ("hp_astar_kernel", "hp_heur_seg", "hp_heur_stitch"))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
rows = rows[-n:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    print(r["Kernel_Name"][:64], "start", round((int(r["Start_Timestamp"]) - t0) / 1e6, 2), "dur_ms", round((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, 2), "grid", r.get("Grid_Size_X") or r.get("Grid_Size"))
