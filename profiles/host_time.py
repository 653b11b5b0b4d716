import sys, time, os
sys.path.insert(0,'/root/repo')
from __graft_entry__ import load_package
rt = load_package()
from rtiow_amd import scenes
import numpy as np
G = "/root/repo/tests/golden/brass_vase/brass_vase_04_4k.gltf"
mesh = scenes.load_gltf_mesh(G)
tex = scenes.load_gltf_textures(G)
copies = int(sys.argv[1]) if len(sys.argv) > 1 else 3623
t0=time.perf_counter()
a = rt.Scene()
scenes.instanced_mesh_scene(a, mesh, copies, tex)
t1=time.perf_counter()
print("scene + reference BVH", round(t1-t0,3), flush=True)
ctx = rt.Context(0)
t1=time.perf_counter()
ctx.upload_scene(a)
t2=time.perf_counter()
print("upload_scene", round(t2-t1,3), "build ms", ctx.last_build_ms(), flush=True)
ctx.set_target(1920,1080); ctx.render(0,2,50); ctx.sync()
print("ok")
