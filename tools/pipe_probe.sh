#!/bin/bash
# usage (GPU box, repo root): bash tools/pipe_probe.sh   -- step time of the native pipeline with subsets of its stages (timing experiment;
# stage bits: 1 = word rings, 2 = draw, 4 = plan, 8 = step).  Without the draw stage the plan stage keeps rebuilding the last drawn batch.
for mask in 15 12 13 14 4 1 3; do
  KGE_PIPE_STAGES=$mask python bench.py --steps 300 --warmup 10 --settle 0 --no-cpu-baseline --no-check --probe 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('stages mask $mask: ms_per_step %.4f  host launch %.4f enqueue %.4f' % (d['ms_per_step'], r['host_graph_launch_ms_per_step'], r['host_enqueue_ms_per_step']))"
done
