#!/bin/bash
# usage (GPU box, repo root): bash tools/pipe_probe.sh   -- step time of the native pipeline with subsets of its stages (timing experiment)
for mask in 7 1 2 4 3 5 6; do
  KGE_PIPE_STAGES=$mask python bench.py --steps 300 --warmup 10 --settle 0 --no-cpu-baseline --no-check 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('stages mask $mask: ms_per_step %.4f  host wait %.4f launch %.4f by stage %s' % (d['ms_per_step'], r['host_wait_ms_per_step'], r['host_graph_launch_ms_per_step'], r['host_wait_by_stage_ms_per_step']))"
done
