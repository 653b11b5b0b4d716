"""Kernel time of ONE rank's share of the bench frame for several world sizes, on one GPU:
rank 0 of `world` owns 1/world of the 8x8 tiles and renders 64*world samples per pixel (the
bench's weak-scaling step).  Shows that a rank's time does not depend on how few pixels it owns.
Usage: python profiles/rank_share.py [world ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package

rt = load_package()
from rtiow_amd import scenes

W, H = 1920, 1080
worlds = [int(x) for x in sys.argv[1:]] or [1, 2, 4, 8]
s = rt.Scene()
scenes.cover_scene(s, W, H)
ctx = rt.Context(0)
ctx.upload_scene(s)
for world in worlds:
    ctx.set_target(W, H, 0, world)
    spp = 64 * world
    ms = []
    for _ in range(3):
        ctx.clear()
        ctx.render(0, spp, 50)
        ctx.sync()
        ms.append(ctx.last_render_ms())
    print(f"world {world}: {spp} spp on 1/{world} of the tiles: {min(ms):.2f} ms", flush=True)
