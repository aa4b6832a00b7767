#!/bin/bash
# usage (GPU box, repo root): bash tools/prof_any.sh <tag> <script.py> [args]  -> rocprofv3 kernel stats of any tool script, top 25 kernels printed
tag=$1; shift
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/prof_$tag -- python3 $root/"$@" > $root/gpurun_out/prof_$tag.log 2>&1
cd $root
python3 - <<PY
import csv,glob
f=sorted(glob.glob("gpurun_out/prof_$tag/*/*kernel_stats.csv"))[-1]
for r in list(csv.DictReader(open(f)))[:25]:
    print(r["Name"][:90].ljust(90), r["Calls"].rjust(5), ("%.1f"%(float(r["AverageNs"])/1e3)).rjust(10), r["Percentage"])
PY
