"""Start-to-start intervals of the step kernel in a rocprofv3 kernel trace: python tools/step_intervals.py <kernel_trace.csv> [kernel substring]"""
import csv
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if (sys.argv[2] if len(sys.argv) > 2 else "step1_kernel") in r["Kernel_Name"]]
ts = sorted(int(r["Start_Timestamp"]) for r in rows)
d = [(b - a) / 1e3 for a, b in zip(ts, ts[1:])]
print(len(ts), "launches; start-to-start us:", " ".join(f"{x:.0f}" for x in d))
