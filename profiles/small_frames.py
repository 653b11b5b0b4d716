"""Launch time of small frames (cover scene): where the fixed cost of a launch shows.  Usage: python profiles/small_frames.py [WxHxSPP ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

rt = load_package()
from rtiow_amd import scenes  # noqa: E402

cases = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or [(64, 64, 1), (1200, 675, 1), (1200, 675, 10), (1920, 1080, 1), (1920, 1080, 4), (1920, 1080, 16), (1920, 1080, 64)]
for (W, H, spp) in cases:
    a = rt.Scene()
    scenes.cover_scene(a, W, H)
    ctx = rt.Context(0)
    ctx.upload_scene(a)
    ctx.set_target(W, H)
    ms = []
    for i in range(6):
        ctx.clear()
        ctx.render(0, spp, 50)
        ctx.sync()
        ms.append(ctx.last_render_ms())
    print(f"{W}x{H}x{spp} RT_TWO_PASS={os.environ.get('RT_TWO_PASS', 'default')}", " ".join(f"{m:.3f}" for m in ms),
          f"-> {W * H * spp / min(ms) / 1e3:.0f} Msamples/s", flush=True)
    ctx.close()
