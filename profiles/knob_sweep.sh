#!/bin/bash
# Sweeps the wavefront loop's vote thresholds on the configurations that read the scene through the caches.
# usage: bash profiles/knob_sweep.sh <case> [<case> ...]   (cases as profiles/bench_configs.py takes them)
set -u
cd "$(dirname "$0")/.."
for c in "$@"; do
  for bk in 4 8 16 24; do for pk in 8 16 24 32; do for sm in 16 32 48; do
    ms=$(RT_BOX_KEEP=$bk RT_PRIM_KEEP=$pk RT_SHADE_MIN=$sm timeout -k 10 200 python -u profiles/bench_configs.py "$c" 2>&1 | grep -E '^\{' | grep -o '"kernel_ms": [0-9.]*') || exit 1
    echo "$c box_keep=$bk prim_keep=$pk shade_min=$sm $ms"
  done; done; done
done
