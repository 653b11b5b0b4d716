#!/bin/bash
# L2 / HBM / wait counters of the configurations whose scene does not fit LDS: the Cornell box with the
# brass vase (three 4096^2 maps) and the 10 M-triangle instanced scene.  One rocprofv3 run per counter set.
# Usage (GPU box, repo root): bash profiles/pmc_configs.sh <tag> <bench_configs.py case> [more cases]
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for CASE in "$@"; do
  NAME=$(echo $CASE | tr ':' '_')
  OUT=$ROOT/gpurun_out/prof_${TAG}_$NAME
  mkdir -p $OUT
  i=0
  for set in "FETCH_SIZE SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM" \
             "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU" \
             "TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCC_REQ_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
    i=$((i+1))
    timeout -s KILL 400 rocprofv3 --pmc $set -d $OUT/pmc$i -o pmc$i -- python3 -u $ROOT/profiles/bench_configs.py $CASE > $OUT/pmc$i.log 2>&1 || { echo "$CASE pass $i failed"; tail -3 $OUT/pmc$i.log; exit 1; }
    echo "$CASE pass $i done"
  done
  python3 $ROOT/profiles/summarize.py $OUT > $ROOT/gpurun_out/prof_${TAG}_${NAME}_summary.txt
  grep -h "^{" $OUT/pmc1.log > $ROOT/gpurun_out/prof_${TAG}_${NAME}_case.json
done
echo all done
