#!/bin/bash
# usage (GPU box, repo root): bash tools/profile_round.sh <tag>  -- everything profiles/ holds for one build, into gpurun_out/<tag>_*
tag=$1
root=$GRAFT_REPO_ROOT
cd $root
python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
python bench.py --model TransE --no-cpu-baseline > gpurun_out/${tag}_bench_transe_l2.json 2>> gpurun_out/${tag}_bench.err
python bench.py --steps 20 --warmup 5 > gpurun_out/${tag}_bench_driver_flags.json 2>> gpurun_out/${tag}_bench.err
python bench.py --gpus 2 --backend gloo --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/${tag}_bench_2ranks_gloo.json 2>> gpurun_out/${tag}_bench.err
echo "bench lines done"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/prof_${tag}_serial -- python3 $root/bench.py --steps 100 --warmup 5 --settle 0 --no-cpu-baseline --no-check --no-sharded-eval --serial > $root/gpurun_out/prof_${tag}_serial.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/prof_${tag} -- python3 $root/bench.py --steps 100 --warmup 5 --settle 0 --no-cpu-baseline --no-check --no-sharded-eval > $root/gpurun_out/prof_${tag}.log 2>&1
cd $root
cp gpurun_out/prof_${tag}_serial/*/*kernel_stats.csv gpurun_out/${tag}_serial_kernel_stats.csv
cp gpurun_out/prof_${tag}/*/*kernel_stats.csv gpurun_out/${tag}_kernel_stats.csv
echo "kernel stats done"
bash tools/pmc.sh ${tag} > gpurun_out/${tag}_pmc_summary.txt 2>&1
bash tools/pmc_rank.sh ${tag} > gpurun_out/${tag}_rank_sq_counters.txt 2>&1
bash tools/pmc_rank_traffic.sh ${tag} > gpurun_out/${tag}_rank_traffic.txt 2>&1
python tools/pipe_trace.py 40 > gpurun_out/${tag}_pipeline_stage_timeline.txt 2>/dev/null
for s in "14541 200 20466 DistMult" "40943 400 3134 ComplEx" "123182 200 5000 DistMult" "14541 200 20466 TransE"; do python tools/rank_bench.py $s; done > gpurun_out/${tag}_rank_bench.txt 2>/dev/null
echo "all done"
