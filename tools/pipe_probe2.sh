#!/bin/bash
# usage (GPU box, repo root): bash tools/pipe_probe2.sh  -- graph replay vs kernel-by-kernel launches from the native loop, per stage
for direct in 0 7 1 2 4 3; do
  KGE_PIPE_DIRECT=$direct python bench.py --steps 300 --warmup 10 --settle 0 --no-cpu-baseline --no-check 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('direct mask $direct: ms_per_step %.4f  host wait %.4f launch %.4f enqueue %.4f' % (d['ms_per_step'], r['host_wait_ms_per_step'], r['host_graph_launch_ms_per_step'], r['host_enqueue_ms_per_step']))"
done
