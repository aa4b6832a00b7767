#!/bin/bash
# A/B on one box: the relation rows closed inside step1_kernel (default) against rel_tail_kernel summing them (KGE_STEP_FUSED_TAIL=0)
for rep in 1 2; do for v in 1 0; do
  KGE_STEP_FUSED_TAIL=$v python bench.py --no-check --no-cpu-baseline --no-sharded-eval --no-hbm-leg --no-dp02 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('fused_tail=$v', 'ms/step %.5f' % d['ms_per_step'], 'step1 %.2f us' % (r['avg_kernel_ms_live_hipevents']*1e3), 'opt/rel_tail %.2f us' % (r['opt_kernel_avg_ms']*1e3), 'loss', d['config']['final_loss'])"
  KGE_STEP_FUSED_TAIL=$v python bench.py --steps 20 --warmup 5 --no-check --no-cpu-baseline --no-sharded-eval --no-hbm-leg --no-dp02 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('   driver flags: ms/step %.5f' % d['ms_per_step'])"
done; done
