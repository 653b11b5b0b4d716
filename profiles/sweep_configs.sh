#!/bin/bash
# Runs profiles/bench_configs.py cases under a list of launch settings.
# Usage: bash profiles/sweep_configs.sh "<cases>" "<env settings 1>" "<env settings 2>" ...
cases=$1; shift
for cfg in "$@"; do
  echo "== $cfg"
  env $cfg timeout -k 10 300 python -u profiles/bench_configs.py $cases 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('  ', d['case'][:40], d['kernel_ms'], 'ms', d['msamples_per_s'], 'Msamples/s', 'exact' if d.get('oracle_rows_bit_exact') else 'MISMATCH')
" || exit 1
done
