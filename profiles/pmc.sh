#!/bin/bash
# One rocprofv3 PMC pass per argument (a quoted counter list) over a short bench run.
# Usage: bash profiles/pmc.sh <tag> "<counters pass 1>" "<counters pass 2>" ...
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "$@"; do
  i=$((i+1))
  timeout -s KILL 200 rocprofv3 --pmc $set -d $OUT/pmc$i -o pmc$i -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/pmc$i.log 2>&1 || { echo "pass $i failed"; tail -3 $OUT/pmc$i.log; exit 1; }
done
echo done $i passes
