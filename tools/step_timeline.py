"""Where the time between two step1_kernel launches of the concurrent pipeline goes: from a `rocprofv3 --kernel-trace` CSV of a bench.py run
(all four streams running), per training step  step1_kernel | gap | rel_tail_kernel | gap to the next step1_kernel, with what ran on the
other streams meanwhile.  usage: python tools/step_timeline.py <kernel_trace.csv>"""
import csv
import sys

import numpy as np

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
s1 = [(a, b) for a, b, n in rows if "step1_kernel" in n]
rt = [(a, b) for a, b, n in rows if "rel_tail_kernel" in n]
if len(s1) < 50:
    print("too few step1_kernel launches in the trace:", len(s1))
    sys.exit(1)
# steady part: drop the first 30 and the last 5 steps
lo, hi = 30, len(s1) - 5
d_s1, g1, d_rt, g2, period = [], [], [], [], []
j = 0
for i in range(lo, hi - 1):
    a, b = s1[i]
    while j < len(rt) and rt[j][0] < b:
        j += 1
    if j >= len(rt) or rt[j][0] > s1[i + 1][0]:
        continue
    ra, rb = rt[j]
    d_s1.append(b - a); g1.append(ra - b); d_rt.append(rb - ra); g2.append(s1[i + 1][0] - rb); period.append(s1[i + 1][0] - a)
us = lambda x: np.asarray(x) / 1e3  # noqa: E731
print(f"{len(period)} steady steps of the concurrent pipeline (rocprofv3 kernel trace: the profiler's per-dispatch overhead is inside these numbers)")
print(f"  start-to-start of step1_kernel        median {np.median(us(period)):7.2f} us   mean {us(period).mean():7.2f}   p90 {np.percentile(us(period), 90):7.2f}")
for name, x in (("step1_kernel (beside the producer stages)", d_s1), ("gap step1 -> rel_tail (stream order)", g1), ("rel_tail_kernel", d_rt),
                ("gap rel_tail -> next step1 (plan-consumed record, wait for the next plan, launch)", g2)):
    print(f"  {name:84s} median {np.median(us(x)):7.2f} us   mean {us(x).mean():7.2f}")
print(f"  sum of the medians {np.median(us(d_s1)) + np.median(us(g1)) + np.median(us(d_rt)) + np.median(us(g2)):7.2f} us")
# what the other streams ran during the steady window
t0, t1 = s1[lo][0], s1[hi - 1][0]
busy = {}
for a, b, n in rows:
    if b < t0 or a > t1 or "step1_kernel" in n or "rel_tail_kernel" in n:
        continue
    k = n.split("(")[0].replace("void ", "").replace("kge::", "")[:48]
    busy.setdefault(k, [0, 0.0])
    busy[k][0] += 1
    busy[k][1] += (min(b, t1) - max(a, t0)) / 1e3
steps = hi - 1 - lo
print(f"other kernels in the same window ({steps} steps): launches per step, us per step")
for k, (c, t) in sorted(busy.items(), key=lambda kv: -kv[1][1])[:14]:
    print(f"  {k:50s} {c / steps:6.2f}  {t / steps:8.2f}")
