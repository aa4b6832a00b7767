#!/bin/bash
# usage (GPU box, repo root): bash tools/pmc.sh <tag>
# Two separate --pmc passes (FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950: MI355X_MICROARCH.md, PMC slots),
# each with --kernel-trace only.  Output: gpurun_out/pmc_<tag>_{fetch,write}/ + a per-kernel summary JSON.
tag=$1
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $root/gpurun_out/pmc_${tag}_fetch -- python3 $root/bench.py --steps 5 --warmup 2 --settle 0 --no-cpu-baseline --no-check --no-sharded-eval --no-hbm-leg --no-dp02 > $root/gpurun_out/pmc_${tag}_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $root/gpurun_out/pmc_${tag}_write -- python3 $root/bench.py --steps 5 --warmup 2 --settle 0 --no-cpu-baseline --no-check --no-sharded-eval --no-hbm-leg --no-dp02 > $root/gpurun_out/pmc_${tag}_write.log 2>&1
cd $root
python3 - <<EOF
import csv, glob, json, collections
out = collections.defaultdict(dict)
for kind in ("fetch", "write"):
    fs = glob.glob("gpurun_out/pmc_${tag}_%s/*/*counter_collection.csv" % kind)
    if not fs:
        print("no counter file for", kind, glob.glob("gpurun_out/pmc_${tag}_%s/*/*" % kind)); continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        out[k][kind + "_kb_per_launch"] = sum(v) / len(v)
        out[k]["launches"] = len(v)
res = {}
for k, v in out.items():
    if not k.startswith("void kge") and not k.startswith("kge"): continue
    f, w = v.get("fetch_kb_per_launch", 0.0), v.get("write_kb_per_launch", 0.0)
    # gfx950: FETCH_SIZE reports 1/2 of the bytes of wide (16 B/lane) reads -> x2; WRITE_SIZE exact for 16-B stores (MI355X_MICROARCH.md, HBM)
    v["hbm_bytes_per_launch_corrected"] = (2.0 * f + w) * 1024.0
    res[k] = v
json.dump(res, open("gpurun_out/pmc_${tag}.json", "w"), indent=1)
for k, v in sorted(res.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch_corrected"])[:12]:
    print(k[:70].ljust(70), "%.1f MB fetch(raw) %.1f MB write -> %.1f MB corrected" % (v.get("fetch_kb_per_launch",0)/1024, v.get("write_kb_per_launch",0)/1024, v["hbm_bytes_per_launch_corrected"]/1e6))
EOF
