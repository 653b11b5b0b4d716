#!/bin/bash
# HBM/L2 fetch traffic of the 10 M-triangle configuration: one FETCH_SIZE pass and one WRITE_SIZE pass
# over profiles/bench_configs.py c5s:1e7:nocheck.  Usage (GPU box, repo root): bash profiles/pmc_c5.sh <tag>
set -u
TAG=${1:-c5}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc1 -o pmc1 -- python3 -u $ROOT/profiles/bench_configs.py c5s:1e7:nocheck > $OUT/pmc1.log 2>&1 || { tail -3 $OUT/pmc1.log; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc2 -o pmc2 -- python3 -u $ROOT/profiles/bench_configs.py c5s:1e7:nocheck > $OUT/pmc2.log 2>&1 || { tail -3 $OUT/pmc2.log; exit 2; }
echo done
