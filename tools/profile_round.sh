#!/bin/bash
# usage (GPU box, repo root): bash tools/profile_round.sh <tag>  -- everything profiles/ holds for one build, into gpurun_out/<tag>_*
tag=$1
root=$GRAFT_REPO_ROOT
cd $root
python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
python bench.py --model TransE --no-cpu-baseline > gpurun_out/${tag}_bench_transe_l2.json 2>> gpurun_out/${tag}_bench.err
python bench.py --steps 20 --warmup 5 > gpurun_out/${tag}_bench_driver_flags.json 2>> gpurun_out/${tag}_bench.err
python bench.py --gpus 2 --backend gloo --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/${tag}_bench_2ranks_gloo.json 2>> gpurun_out/${tag}_bench.err
python bench.py --gpus 2 --backend gloo --train rows --steps 20 --warmup 5 --no-cpu-baseline --no-check --no-sharded-eval > gpurun_out/${tag}_bench_2ranks_gloo_rows.json 2>> gpurun_out/${tag}_bench.err
echo "bench lines done"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/prof_${tag}_serial -- python3 $root/bench.py --steps 100 --warmup 5 --settle 0 --no-cpu-baseline --no-check --no-sharded-eval --no-hbm-leg --no-dp02 --serial > $root/gpurun_out/prof_${tag}_serial.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/prof_${tag} -- python3 $root/bench.py --steps 100 --warmup 5 --settle 0 --no-cpu-baseline --no-check --no-sharded-eval --no-hbm-leg --no-dp02 > $root/gpurun_out/prof_${tag}.log 2>&1
# the HBM leg alone, as the bench runs it (the benched pipeline on the 10 M entity table: a pipelined pass, then kernel by kernel): step1_kernel average on THAT shape (a serial pass alone, from cold, reads 6 - 12 % longer), + its PMC traffic
rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/prof_${tag}_hbm -- python3 $root/bench.py --leg hbm --steps 20 > $root/gpurun_out/prof_${tag}_hbm.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $root/gpurun_out/pmc_${tag}_hbm_fetch -- python3 $root/bench.py --leg hbm --serial --steps 6 > $root/gpurun_out/pmc_${tag}_hbm_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $root/gpurun_out/pmc_${tag}_hbm_write -- python3 $root/bench.py --leg hbm --serial --steps 6 > $root/gpurun_out/pmc_${tag}_hbm_write.log 2>&1
# kernel trace of the concurrent pipeline at the driver's flags: where the time between two step1_kernel launches goes
rocprofv3 --kernel-trace --output-format csv -d $root/gpurun_out/prof_${tag}_trace -- python3 $root/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-check --no-sharded-eval --no-hbm-leg --no-dp02 > $root/gpurun_out/prof_${tag}_trace.log 2>&1
cd $root
cp gpurun_out/prof_${tag}_hbm/*/*kernel_stats.csv gpurun_out/${tag}_hbm_kernel_stats.csv
python tools/pmc_hbm.py ${tag} > gpurun_out/${tag}_hbm_pmc_summary.txt 2>&1
python tools/step_timeline.py gpurun_out/prof_${tag}_trace/*/*kernel_trace.csv > gpurun_out/${tag}_step_timeline.txt 2>&1
cp gpurun_out/prof_${tag}_serial/*/*kernel_stats.csv gpurun_out/${tag}_serial_kernel_stats.csv
cp gpurun_out/prof_${tag}/*/*kernel_stats.csv gpurun_out/${tag}_kernel_stats.csv
echo "kernel stats done"
bash tools/pmc.sh ${tag} > gpurun_out/${tag}_pmc_summary.txt 2>&1
bash tools/pmc_rank.sh ${tag} > gpurun_out/${tag}_rank_sq_counters.txt 2>&1
bash tools/pmc_rank_traffic.sh ${tag} > gpurun_out/${tag}_rank_traffic.txt 2>&1
python tools/pipe_trace.py 40 > gpurun_out/${tag}_pipeline_stage_timeline.txt 2>/dev/null
for s in "14541 200 20466 DistMult" "40943 400 3134 ComplEx" "123182 200 5000 DistMult" "14541 200 20466 TransE"; do python tools/rank_bench.py $s; done > gpurun_out/${tag}_rank_bench.txt 2>/dev/null
python tools/pairlog_bench.py > gpurun_out/${tag}_pairwize_logistic_bench.txt 2>/dev/null
echo "all done"
