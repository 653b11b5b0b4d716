"""One world of profiles/fuzz_parity.py, with variations, to narrow a mismatch down.
Usage: python profiles/fuzz_one.py <seed>"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import fuzz_parity as fz

seed = int(sys.argv[1])
rt, orc = fz.rt, fz.orc
build = fz.make_builder(seed)
rr = np.random.RandomState(seed + 7)
w, h = int(rr.randint(2, 97)), int(rr.randint(2, 71))
spp, depth = int(rr.randint(1, 5)), int(rr.choice([1, 3, 50]))
ctx = rt.Context(0)
for label, nm, ref in (("as fuzzed", None, 0), ("normal mapping off", False, 0), ("reference-order walk", None, 1)):
    a, b = rt.Scene(), orc.OracleScene()
    build(a)
    build(b)
    if nm is not None:
        a.set_normal_mapping(nm)
        b.set_normal_mapping(nm)
    ctx.set_reference_traversal(ref)
    ctx.upload_scene(a)
    ctx.set_target(w, h)
    ctx.render(0, spp, depth)
    ctx.sync()
    got = ctx.read_accum()
    want, _, _ = b.render(w, h, 0, spp, depth, rng_mode=1, order=1, threads=8)
    diff = (got[..., :3].view(np.uint32) != want[..., :3].view(np.uint32)).any(axis=-1)
    print(label, ":", int(diff.sum()), "pixels differ;", ctx.count(0, spp, depth))
    for y, x in np.argwhere(diff)[:4]:
        print("   pixel", x, y, "gpu", got[y, x, :3], "oracle", want[y, x, :3])
        for s in range(spp):
            ctx.clear()
            ctx.render(s, 1, depth)
            ctx.sync()
            g1 = ctx.read_accum()[y, x, :3].copy()
            w1 = b.render(w, h, s, 1, depth, rng_mode=1, order=1, threads=8)[0][y, x, :3]
            if not np.array_equal(g1.view(np.uint32), w1.view(np.uint32)):
                print("      sample", s, "gpu", g1, "oracle", w1)

# the differing samples, ray by ray: the oracle's path against the device's closest hit for the same rays
a, b = rt.Scene(), orc.OracleScene()
build(a)
build(b)
ctx.set_reference_traversal(0)
ctx.upload_scene(a)
ctx.set_target(w, h)
want = b.render(w, h, 0, spp, depth, rng_mode=1, order=1, threads=8)[0]
ctx.clear()
ctx.render(0, spp, depth)
ctx.sync()
got = ctx.read_accum()
diff = (got[..., :3].view(np.uint32) != want[..., :3].view(np.uint32)).any(axis=-1)
for y, x in np.argwhere(diff)[:4]:
    for s in range(spp):
        rays, hits, t = b.path_rays(w, h, int(x), int(y), s, depth)
        kind, pid, tg, _, _, _ = ctx.trace(rays)
        bad = np.nonzero((kind != hits[:, 0]) | (pid != hits[:, 1]) | (tg.view(np.uint32) != t.view(np.uint32)))[0]
        if len(bad):
            k = int(bad[0])
            print(f"pixel {x},{y} sample {s}: ray {k} of {len(rays)} differs")
            print("   ray", rays[k].tolist())
            print("   oracle kind/id/t", hits[k].tolist(), float(t[k]), " device", int(kind[k]), int(pid[k]), float(tg[k]))
            sp = a.flatten()
            print("   scene:", {kk: (vv if not hasattr(vv, "shape") else vv.shape) for kk, vv in list(sp.items())[:6]} if isinstance(sp, dict) else type(sp))
