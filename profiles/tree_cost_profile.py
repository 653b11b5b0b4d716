"""Where the surface-area cost of a search structure sits: entries sorted by box area, cumulative cost at ranks
10, 100, ... for the host-built and the device-built structure of the same vase scene.
python profiles/tree_cost_profile.py <triangles> [<radius> ...]"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package

rt = load_package()
from rtiow_amd import scenes

G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests/golden/brass_vase/brass_vase_04_4k.gltf")


def profile(entries):
    f = entries.view(np.float32)
    d = np.stack([f[:, 1] - f[:, 0], f[:, 3] - f[:, 2], f[:, 5] - f[:, 4]], 1).astype(np.float64)
    d = np.abs(d)
    area = d[:, 0] * d[:, 1] + d[:, 1] * d[:, 2] + d[:, 2] * d[:, 0]
    area = np.sort(area[np.isfinite(area)])[::-1]
    cum = np.cumsum(area) / area[0]
    ranks = [10, 100, 1000, 10000, 100000, len(area)]
    return {str(r): round(float(cum[min(r, len(area)) - 1]), 2) for r in ranks}


n_tris = int(float(sys.argv[1]))
mesh = scenes.load_gltf_mesh(G)
copies = max(1, n_tris // mesh["indices"].shape[0])
tex = {"diffuse": scenes.noise_texture(1024, 1), "roughness": scenes.noise_texture(1024, 2), "normal": None}
a = rt.Scene()
scenes.instanced_mesh_scene(a, mesh, copies, tex)
for mode in ["host"] + sys.argv[2:]:
    if mode != "host":
        os.environ["RT_PLOC_RADIUS"] = mode
    ctx = rt.Context(0)
    ctx.set_device_build(mode != "host")
    ctx.upload_scene(a)
    ent = ctx.read_search_structure()[0]
    print(json.dumps({"build": mode, "entries": int(ent.shape[0]), "cumulative_area_cost_at_rank": profile(ent)}), flush=True)
    ctx.close()
