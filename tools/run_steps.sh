#!/bin/bash
# usage: bash tools/run_steps.sh <tag> "<cmd1>" "<cmd2>" ...   (GPU box, repo root)
# Runs the commands one after the other, each under its own `timeout -k 10 <secs>` (prefix "NNN::" sets the seconds, default 420),
# output of step i into gpurun_out/<tag>_step<i>.log.  An ordinary failure (non-zero exit) does not stop the chain; a step that was KILLED at
# its limit does (no further GPU step after a hang).
tag=$1; shift
mkdir -p gpurun_out
i=0
for c in "$@"; do
  i=$((i+1))
  secs=420
  if [[ "$c" =~ ^([0-9]+)::(.*)$ ]]; then secs=${BASH_REMATCH[1]}; c=${BASH_REMATCH[2]}; fi
  echo "== step $i (limit ${secs}s): $c"
  timeout -k 10 $secs bash -c "$c" > gpurun_out/${tag}_step${i}.log 2>&1
  rc=$?
  echo "   rc=$rc"
  tail -n 4 gpurun_out/${tag}_step${i}.log | cut -c1-600
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step $i hit its time limit: stopping"; exit 1; fi
done
exit 0
