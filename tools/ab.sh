#!/bin/bash
# tools/ab.sh NAME...: the default bench (no CPU baseline, no parity check) once per named build of the library ("default" = the in-tree
# libkge_hip.so, else kge-keras_amd/csrc/variants/libkge_hip_NAME.so); prints ms per step and the serial kernel averages.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for name in "$@"; do
  lib=""
  [ "$name" != default ] && lib=$PWD/kge-keras_amd/csrc/variants/libkge_hip_$name.so
  for rep in 1 2; do
    KGE_HIP_LIB=$lib timeout -k 10 200 python bench.py --no-cpu-baseline --no-check > gpurun_out/ab_$name.json 2> gpurun_out/ab_$name.err || { echo "$name FAILED"; tail -5 gpurun_out/ab_$name.err; continue; }
    python - "$name" <<'PY'
import json, sys
n = sys.argv[1]
j = json.loads(open(f"gpurun_out/ab_{n}.json").read().strip().splitlines()[-1])
r = j["roofline"]
k = {a.split(" ")[0]: round(b["avg_ms"] * 1e3, 1) for a, b in r["all_kernels"].items()}
print(f"    host enqueue {r.get('host_enqueue_ms_per_step')} launch {r.get('host_graph_launch_ms_per_step')}")
print(f"{n:12s} ms/step {j['ms_per_step']:.4f}  value {j['value']/1e9:.3f} G  serial us: {k} opt {r['opt_kernel_avg_ms']*1e3:.1f} ranks/s {j.get('value_ranks_per_sec', 0)/1e6:.2f} M")
PY
  done
done
