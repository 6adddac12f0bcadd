"""Last launches of a rocprofv3 kernel trace: name, start, duration, grid."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 14
t0 = int(rows[-n]["Start_Timestamp"]) if len(rows) >= n else int(rows[0]["Start_Timestamp"])
for r in rows[-n:]:
    print(r["Kernel_Name"][:70], "start", round((int(r["Start_Timestamp"]) - t0) / 1e6, 2), "dur_ms", round((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, 2), "grid", r.get("Grid_Size_X") or r.get("Grid_Size"))
