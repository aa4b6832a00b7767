#!/bin/bash
# usage (on the GPU box, from the repo root): bash tools/prof.sh <tag>   -> gpurun_out/prof_<tag>/ + summary print
tag=$1; shift
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/prof_$tag -- python3 $root/bench.py --steps 100 --warmup 3 --no-cpu-baseline "$@" > $root/gpurun_out/prof_$tag.log 2>&1
cd $root
python3 - <<EOF
import csv,glob
f=glob.glob("gpurun_out/prof_$tag/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    print(r["Name"][:60].ljust(60), r["Calls"].rjust(5), ("%.1f"%(float(r["AverageNs"])/1e3)).rjust(9), r["Percentage"])
EOF
