"""Split loop (RT_KERNEL=split) against the default loop: identical accumulation buffers, and timing.
Usage: python profiles/split_check.py [width height spp]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package

rt = load_package()
from rtiow_amd import scenes

W, H, SPP = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (256, 256, 4)
s = rt.Scene()
scenes.cover_scene(s, W, H)
ctx = rt.Context(0)
ctx.upload_scene(s)
ctx.set_target(W, H)


def render(kernel):
    if kernel:
        os.environ["RT_KERNEL"] = kernel
    else:
        os.environ.pop("RT_KERNEL", None)
    ctx.clear()
    ctx.render(0, SPP, 50)
    ctx.sync()
    img = ctx.read_accum().copy()
    times = []
    for _ in range(3):
        ctx.clear()
        t0 = time.perf_counter()
        ctx.render(0, SPP, 50)
        ctx.sync()
        times.append((time.perf_counter() - t0) * 1e3)
    return img, min(times), ctx.last_render_ms()


a, ta, ka = render(None)
print(f"default: {ta:.3f} ms wall, {ka:.3f} ms kernels", flush=True)
b, tb, kb = render("split")
print(f"split:   {tb:.3f} ms wall, {kb:.3f} ms kernels", flush=True)
same = np.array_equal(a.view(np.uint32), b.view(np.uint32))
print("identical:", same, "nonzero pixels:", int((b[..., :3].sum(-1) > 0).sum()), "of", W * H)
if not same:
    d = np.argwhere(a.view(np.uint32) != b.view(np.uint32))
    print("differing entries:", len(d), "first:", d[:5].tolist())
    sys.exit(1)
