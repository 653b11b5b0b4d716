#!/bin/bash
# rocprofv3 kernel trace + stats of a short bench run.  Usage: bash profiles/trace_only.sh <tag>
set -u
TAG=${1:-t}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/trace.log 2>&1 || exit 1
cd $ROOT && python3 profiles/summarize.py $OUT
