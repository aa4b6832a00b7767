#!/bin/bash
# usage (GPU box, repo root): bash tools_pmc_rank.sh <tag> -- SQ counters of the 1-vs-all sweep
tag=$1
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $root/gpurun_out/pmcr_${tag} -- python3 $root/tools/rank_bench.py > $root/gpurun_out/pmcr_${tag}.log 2>&1
cd $root
python3 - <<EOF
import csv, glob, collections
fs = glob.glob("gpurun_out/pmcr_${tag}/*/*counter_collection.csv")
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(fs[0])):
    acc[r["Kernel_Name"][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    if "rank" not in k: continue
    print(k, {c: round(sum(x)/len(x)) for c, x in v.items()})
EOF
tail -3 gpurun_out/pmcr_${tag}.log
