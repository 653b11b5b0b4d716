#!/bin/bash
# Runs bench.py under a list of scheduling / launch settings and prints ms per step for each.
# Usage: bash profiles/sweep.sh "RT_THREADS=256 RT_BLOCKS_PER_CU=5" "RT_THREADS=512 ..." ...
for cfg in "$@"; do
  out=$(env $cfg timeout -k 10 120 python -u bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1)
  rc=$?
  ms=$(echo "$out" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernel_ms'])" 2>/dev/null)
  echo "$cfg => $ms (rc=$rc)"
  if [ -z "$ms" ]; then echo "FAILED: $out" | cut -c1-400; exit 1; fi
done
