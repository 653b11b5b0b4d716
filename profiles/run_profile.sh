#!/bin/bash
# Collects the rocprofv3 kernel trace + stats and two PMC passes for the bench workload.
# Usage (on the GPU box, from the repo root): bash profiles/run_profile.sh <tag>
set -u
TAG=${1:-r1}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/trace.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_LDS -d $OUT/pmc1 -o pmc1 -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/pmc1.log 2>&1 || exit 2
timeout -s KILL 200 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM -d $OUT/pmc2 -o pmc2 -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/pmc2.log 2>&1 || exit 3
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc3 -o pmc3 -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/pmc3.log 2>&1 || exit 4
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE -d $OUT/pmc4 -o pmc4 -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/pmc4.log 2>&1 || exit 5
ls -R $OUT | head -50
