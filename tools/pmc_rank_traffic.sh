#!/bin/bash
# usage (GPU box, repo root): bash tools/pmc_rank_traffic.sh <tag> -- memory-side bytes of the 1-vs-all sweep kernel (two separate --pmc
# passes: FETCH_SIZE and WRITE_SIZE do not share a pass on gfx950; FETCH_SIZE x2 for 16-byte-per-lane reads, MI355X_MICROARCH.md)
tag=$1
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $root/gpurun_out/pmct_${tag}_$c -- python3 $root/tools/rank_bench.py > $root/gpurun_out/pmct_${tag}_$c.log 2>&1
done
cd $root
python3 - <<PY
import csv, glob, collections
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    fs = glob.glob("gpurun_out/pmct_${tag}_%s/*/*counter_collection.csv" % c)
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if "rank_sweep" in r["Kernel_Name"]:
            acc[r["Kernel_Name"][:60]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        out.setdefault(k, {})[c] = sum(v) / len(v)
for k, v in out.items():
    f, w = v.get("FETCH_SIZE", 0.0), v.get("WRITE_SIZE", 0.0)
    print(k, "fetch(raw) %.1f MB  write %.1f MB  -> corrected %.1f MB per launch" % (f / 1024, w / 1024, (2 * f + w) * 1024 / 1e6))
PY
