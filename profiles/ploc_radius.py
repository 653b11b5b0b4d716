"""Search radius of the device build's clustering against the frame time of the tree it gives (vase scene, device
build only).  python profiles/ploc_radius.py <triangles> <radius>[:<top>[:<collapse>]] [...]"""
import json
import numpy as np
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package

rt = load_package()
from rtiow_amd import scenes

G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests/golden/brass_vase/brass_vase_04_4k.gltf")
n_tris = int(float(sys.argv[1]))
mesh = scenes.load_gltf_mesh(G)
copies = max(1, n_tris // mesh["indices"].shape[0])
tex = {"diffuse": scenes.noise_texture(1024, 1), "roughness": scenes.noise_texture(1024, 2), "normal": None}
a = rt.Scene()
scenes.instanced_mesh_scene(a, mesh, copies, tex)
for radius in sys.argv[2:]:
    parts = radius.split(":")   # "radius[:clusters the host builds the top over[:left-out-entry threshold]]"
    radius = parts[0]
    os.environ["RT_PLOC_RADIUS"] = radius
    if len(parts) > 1 and parts[1]:
        os.environ["RT_DEVICE_TOP"] = parts[1]
    if len(parts) > 2:
        os.environ["RT_DEVICE_COLLAPSE"] = parts[2]
    ctx = rt.Context(0)
    ctx.set_device_build(True)
    ctx.upload_scene(a)
    ctx.set_target(1920, 1080)
    ctx.render(0, 2, 50)
    ctx.sync()
    times = []
    for _ in range(3):
        ctx.clear()
        ctx.render(0, 16, 50)
        ctx.sync()
        times.append(ctx.last_render_ms())
    cnt = ctx.count(0, 2, 50)
    ent = ctx.read_search_structure()[0]
    f = ent.view(np.float32)
    d = np.maximum(0.0, np.stack([f[:, 1] - f[:, 0], f[:, 3] - f[:, 2], f[:, 5] - f[:, 4]], 1).astype(np.float64))
    area = d[:, 0] * d[:, 1] + d[:, 1] * d[:, 2] + d[:, 2] * d[:, 0]
    area = area[np.isfinite(area)]
    cost = float(area.sum() / area.max())
    print(json.dumps({"triangles": copies * mesh["indices"].shape[0], "radius": int(radius), "top": os.environ.get("RT_DEVICE_TOP", "default"), "collapse": os.environ.get("RT_DEVICE_COLLAPSE", "default"), "build_ms": round(ctx.last_build_ms(), 2),
                      "kernel_ms": [round(t, 2) for t in times], "node_tests_per_ray": round(cnt["node_tests"] / cnt["rays"], 2), "entries": int(ent.shape[0]), "area_cost": round(cost, 2)}), flush=True)
    ctx.close()
