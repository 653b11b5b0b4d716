"""Times the split loop over its knobs.  Usage: python profiles/split_sweep.py "RT_SPLIT_TRACE_WAVES=7,8,9" "RT_SPLIT_SWAP_MIN=8,16" ..."""
import itertools
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package

rt = load_package()
from rtiow_amd import scenes

W, H, SPP = 1920, 1080, 64
s = rt.Scene()
scenes.cover_scene(s, W, H)
ctx = rt.Context(0)
ctx.upload_scene(s)
ctx.set_target(W, H)


def timed():
    best = 1e9
    for _ in range(3):
        ctx.clear()
        ctx.render(0, SPP, 50)
        ctx.sync()
        best = min(best, ctx.last_render_ms())
    return best


os.environ.pop("RT_KERNEL", None)
print(f"default loop: {timed():.3f} ms", flush=True)
os.environ["RT_KERNEL"] = "split"
axes = [(a.split("=")[0], a.split("=")[1].split(",")) for a in sys.argv[1:]]
for combo in itertools.product(*[v for _, v in axes]):
    for (k, _), v in zip(axes, combo):
        os.environ[k] = v
    print(" ".join(f"{k}={v}" for (k, _), v in zip(axes, combo)), f"{timed():.3f} ms", flush=True)
