"""Print the kernel timeline between two consecutive fwd_rel launches from a rocprofv3 kernel trace: python tools/timeline.py <trace.csv> [k]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "fwd_rel_kernel" in r["Kernel_Name"]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 12
t0, t1 = int(rows[idx[k]]["Start_Timestamp"]), int(rows[idx[k + 1]]["Start_Timestamp"])
print("tick us", (t1 - t0) / 1e3)
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if t0 <= s < t1:
        print(f"{(s - t0) / 1e3:8.1f} {(e - s) / 1e3:7.1f}  q{r.get('Queue_Id', '?')} {r['Kernel_Name'][:60]}")
