#!/bin/bash
# A/B of an environment switch of the library on ONE box, training step only: tools/ab_step.sh VAR A B [C ...]  (ms per step of bench.py, two rounds)
var=$1; shift
for rep in 1 2; do
for v in "$@"; do
  export $var=$v
  python bench.py --no-check --no-cpu-baseline --no-sharded-eval 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$var=$v', 'ms_per_step %.5f' % d['ms_per_step'], 'kernel us %.1f' % (d['roofline']['avg_kernel_ms'] * 1e3))"
done; done
