"""Where the persistent kernel's waves spend their time: clock ticks per operation, from the
diagnostic timing instantiation (RT_TIMING=1 turns a counting launch into a timing one).
Usage: python profiles/op_ticks.py [cover|cornell] [spp]"""
import os
import sys

os.environ["RT_TIMING"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

rt = load_package()
from rtiow_amd import scenes  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "cover"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 8
W, H = 1920, 1080
s = rt.Scene()
if which == "cover":
    scenes.cover_scene(s, W, H)
else:
    mesh = scenes.load_gltf_mesh(os.path.join(ROOT, "tests", "golden", "brass_vase", "brass_vase_04_4k.gltf"))
    tex = {"diffuse": scenes.noise_texture(4096, 1), "roughness": scenes.noise_texture(4096, 2),
           "normal": scenes.noise_texture(4096, 3)}
    scenes.cornell_mesh_scene(s, W, H, mesh, tex)
ctx = rt.Context(0)
ctx.upload_scene(s)
ctx.set_target(W, H)
ctx.count(0, spp, 50)
st = ctx.sched_stats()
vals = [x for pair in st.values() for x in pair]   # the 14 raw slots
names = ["box bursts", "primitive runs", "shade: guard test", "shade: surface", "shade: scatter",
         "shade: next ray / end of path", "newpath: finish missed path", "newpath: queue refill",
         "newpath: camera ray", "reference-order box steps", "votes and the rest", "shade: set-up of the next ray"]
ticks = vals[:len(names)]
waves = vals[13]
total = sum(ticks)
print(f"{which}: {waves} waves, {total / max(waves, 1):.0f} ticks per wave")
for n, t in zip(names, ticks):
    print(f"  {n:32s} {100.0 * t / max(total, 1):5.1f} %   {t / max(waves, 1):12.0f} ticks per wave")
