"""Wall time of rt_run() at 1080p, 64 spp, for several samples-per-pass settings (the build-up
mode of the reference renders 1 sample per pass).  Usage: python profiles/progressive.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package

rt = load_package()
from rtiow_amd import scenes

W, H = 1920, 1080
s = rt.Scene()
scenes.cover_scene(s, W, H)
ctx = rt.Context(0)
ctx.upload_scene(s)
ctx.set_target(W, H)
ctx.run(4, 4, 50)
for per_pass in (64, 16, 4, 1):
    t0 = time.perf_counter()
    ctx.run(64, per_pass, 50)
    ctx.sync()
    dt = time.perf_counter() - t0
    print(f"{per_pass:3d} samples per pass: {1e3 * dt:7.2f} ms for 64 spp ({64 // per_pass} passes)", flush=True)

# one-sample passes: device time of a pass (HIP events) against host wall time per pass
ctx.clear()
ms, t0 = [], time.perf_counter()
for i in range(32):
    ctx.render(i, 1, 50)
    ms.append(ctx.last_render_ms())
ctx.sync()
wall = (time.perf_counter() - t0) / 32
print(f"1-spp pass: device {sum(ms) / len(ms):.3f} ms (min {min(ms):.3f}), host wall {1e3 * wall:.3f} ms per pass", flush=True)
t0 = time.perf_counter()
for i in range(32):
    ctx.render(32 + i, 1, 50)
ctx.sync()
print(f"1-spp passes enqueued back to back: {1e3 * (time.perf_counter() - t0) / 32:.3f} ms per pass", flush=True)

# build-up display: after every 1-spp pass the RGBA8 frame is resolved and copied to the host, as the
# reference application re-uploads its byte buffer every frame (R/src/RaytracingApplication.cpp:193-194)
import numpy as np
frame = np.zeros((H, W, 4), np.uint8)
t_read = []


def show(done):
    t = time.perf_counter()
    frame[...] = ctx.read_rgba8(done)
    t_read.append(time.perf_counter() - t)


t0 = time.perf_counter()
ctx.run(64, 1, 50, on_pass=show)
dt = time.perf_counter() - t0
print(f"build-up display, 64 one-sample passes each followed by resolve + RGBA8 read-back: {1e3 * dt:.2f} ms "
      f"({1e3 * dt / 64:.3f} ms per pass, of which read-back {1e3 * sum(t_read) / 64:.3f} ms)", flush=True)
