"""Kernel time of the bench workload (cover scene, 1920x1080, 64 spp) for the library named by
RTIOW_LIB (default: the in-tree build): median of N launches by HIP events.  For A/B runs of
experimental builds.  Usage: RTIOW_LIB=build_abl/x.so python profiles/ab_time.py [launches] [spp]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

rt = load_package()
from rtiow_amd import scenes  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 7
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 64
W, H = 1920, 1080
a = rt.Scene()
scenes.cover_scene(a, W, H)
ctx = rt.Context(0)
ctx.upload_scene(a)
ctx.set_target(W, H)
ms = []
for i in range(n + 2):
    ctx.clear()
    ctx.render(0, spp, 50)
    ctx.sync()
    if i >= 2:
        ms.append(ctx.last_render_ms())
img = ctx.read_accum()
print(f"{os.environ.get('RTIOW_LIB', 'in-tree'):40s} median {np.median(ms):7.3f} ms  min {min(ms):7.3f}  max {max(ms):7.3f}  "
      f"checksum {int(img[..., :3].view(np.uint32).astype(np.uint64).sum()):d}", flush=True)
ctx.close()
