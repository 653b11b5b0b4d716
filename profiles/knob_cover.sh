#!/bin/bash
# Sweeps the wavefront loop's vote thresholds on the bench workload (cover scene, 1080p, 64 spp).
# usage (GPU box, repo root): bash profiles/knob_cover.sh   (RT_TWO_PASS=0 for the single-pass launch)
for bk in 4 8 12 16; do for pk in 8 16 24; do for sm in 24 32 40 48; do
  ms=$(RT_BOX_KEEP=$bk RT_PRIM_KEEP=$pk RT_SHADE_MIN=$sm timeout -k 10 100 python profiles/ab_time.py 3 | grep -o "median *[0-9.]*") || exit 1
  echo "box_keep=$bk prim_keep=$pk shade_min=$sm $ms"
done; done; done
