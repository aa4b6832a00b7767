"""Kernel timeline of the concurrent pipeline from a rocprofv3 kernel trace: python tools/timeline.py <kernel_trace.csv> [first_step] [n_steps] [anchor]
For n consecutive training steps (a step = from one start of the anchor kernel, default fwd_rel_kernel, to the next) prints every kernel with its start offset,
duration and queue, then per queue the busy time and the idle gaps inside the window."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
anchor = sys.argv[4] if len(sys.argv) > 4 else "fwd_rel_kernel"
idx = [i for i, r in enumerate(rows) if anchor in r["Kernel_Name"]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 12
n = int(sys.argv[3]) if len(sys.argv) > 3 else 2
t0, t1 = int(rows[idx[k]]["Start_Timestamp"]), int(rows[idx[k + n]]["Start_Timestamp"])
print("window us", (t1 - t0) / 1e3, "per step", (t1 - t0) / 1e3 / n)
busy = collections.defaultdict(float)
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if t0 <= s < t1:
        q = r.get("Queue_Id", "?")
        busy[q] += (e - s) / 1e3
        name = r["Kernel_Name"].replace("void ", "").replace("kge::", "")
        print(f"{(s - t0) / 1e3:8.1f} {(e - s) / 1e3:7.1f}  q{q} {name[:48]}")
for q, b in sorted(busy.items()):
    print(f"queue {q}: busy {b:.1f} us of {(t1 - t0) / 1e3:.1f}")
