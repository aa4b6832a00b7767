#!/bin/bash
# usage (GPU box, repo root): bash tools/prof_dp.sh <tag> [bench flags]  -> kernel trace of the one-rank rehearsal of the native data-parallel step
tag=$1; shift
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/prof_$tag -- python3 $root/bench.py --steps 100 --warmup 3 --no-cpu-baseline --no-sharded-eval --no-hbm-leg --no-dp02 --force-dist --dp-native on "$@" > $root/gpurun_out/prof_$tag.log 2>&1
rc=$?
cd $root
tail -3 gpurun_out/prof_$tag.log
f=$(ls gpurun_out/prof_$tag/*/*kernel_trace.csv | head -1)
python3 tools/timeline.py $f 40 2 pull_self_kernel > gpurun_out/prof_${tag}_timeline.txt 2>&1
tail -60 gpurun_out/prof_${tag}_timeline.txt
exit $rc
