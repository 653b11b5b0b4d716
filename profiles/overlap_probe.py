"""Two contexts on one GPU render the bench frame side by side (each on its own stream): what overlapping the tail of
one launch with the head of the next could gain.  Usage: python profiles/overlap_probe.py [spp] [launches]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

rt = load_package()
from rtiow_amd import scenes  # noqa: E402

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
W, H = 1920, 1080
a = rt.Scene()
scenes.cover_scene(a, W, H)
ctxs = []
for _ in range(2):
    c = rt.Context(0)
    c.upload_scene(a)
    c.set_target(W, H)
    c.render(0, spp, 50)
    c.sync()
    ctxs.append(c)
for mode in ("one context", "two contexts"):
    t0 = time.perf_counter()
    for i in range(n):
        c = ctxs[i & 1] if mode == "two contexts" else ctxs[0]
        c.render(0, spp, 50)
    for c in ctxs:
        c.sync()
    dt = time.perf_counter() - t0
    print(f"{mode}: {1e3 * dt / n:.3f} ms per {spp}-spp launch", flush=True)
for c in ctxs:
    c.close()
