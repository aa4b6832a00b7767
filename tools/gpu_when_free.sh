#!/bin/bash
# usage (build container): tools/gpu_when_free.sh <log> <timeout> '<command>' -- submits a gpurun call, waiting out "no slot free" answers (exit 3:
# nothing was charged); any other answer ends the loop.  Never resubmits a call that ran.
log=$1; to=$2; cmd=$3
for i in $(seq 1 30); do
  /usr/local/graft/bin/gpurun --timeout $to -- "$cmd" > $log 2>&1
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 90
done
exit 3
