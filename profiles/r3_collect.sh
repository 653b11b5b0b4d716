#!/bin/bash
# Evidence in one GPU job: bench profile (kernel trace + PMC passes), BASELINE configurations, memory
# counters of the configurations read through the caches, scheduler statistics, operation ticks.
# Usage (GPU box, repo root): bash profiles/r3_collect.sh
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
bash profiles/run_profile.sh r3c > gpurun_out/r3c_run_profile.log 2>&1 || { echo "run_profile failed"; tail -3 gpurun_out/r3c_run_profile.log; exit 1; }
echo "bench profile done"
timeout -k 10 900 python profiles/bench_configs.py c1 c3 c4 c5s c5s:10000000 c5i c5 2> gpurun_out/r3_configs.err | grep "^{" > gpurun_out/r3_configs.jsonl || { echo "configs failed"; tail -3 gpurun_out/r3_configs.err; exit 2; }
echo "configs done"
bash profiles/pmc_configs.sh r3 c3 c5s:1e7:nocheck c5i:1e7:nocheck > gpurun_out/r3_pmc_configs.log 2>&1 || { echo "pmc configs failed"; tail -3 gpurun_out/r3_pmc_configs.log; exit 3; }
echo "pmc configs done"
RT_TWO_PASS=1 timeout -k 10 100 python profiles/sched_stats.py 16 > gpurun_out/r3_sched_stats.txt 2>&1 || exit 4
RT_TWO_PASS=0 timeout -k 10 100 python profiles/sched_stats.py 16 > gpurun_out/r3_sched_stats_single_pass.txt 2>&1 || exit 4
timeout -k 10 100 python profiles/op_ticks.py cover 16 > gpurun_out/r3_op_ticks.txt 2>&1 || exit 5
timeout -k 10 300 python profiles/op_ticks.py cornell 16 >> gpurun_out/r3_op_ticks.txt 2>&1 || exit 5
timeout -k 10 600 python profiles/bench_configs.py build:1000000 build:10000000 2> gpurun_out/r3_build_compare.err | grep "^{" > gpurun_out/r3_build_compare.jsonl || exit 6
echo "all done"
