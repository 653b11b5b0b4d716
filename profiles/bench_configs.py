"""Timing of the other BASELINE.json configurations (parity-test cases, not the bench line):

  C1  cover scene 1200x675, 10 spp                      (the reference's CPU-runnable case)
  C3  Cornell box + brass vase (glTF fixture, 2,760 triangles) with its three 4096^2 JPEG maps, 1080p 256 spp
  C4  cover scene 3840x2160, 1024 spp
  C5s instanced vases on a jittered grid, N triangles (default 1e6), 1080p 16 spp, unlimited-depth reference BVH
  C5  the same scene at 3840x2160, 256 spp

Each case: upload, one timed launch (HIP events), path counters, and a bit-exact spot check of a few
image rows against the CPU oracle.  Prints one JSON line per case.

Usage: python profiles/bench_configs.py [c1] [c3] [c4] [c5s[:triangles]] [c5[:triangles]] [c5i[:triangles]]
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_oracle, load_package  # noqa: E402

rt = load_package()
orc = load_oracle()
from rtiow_amd import scenes  # noqa: E402

G = os.path.join(ROOT, "tests", "golden", "brass_vase", "brass_vase_04_4k.gltf")


def run(name, build, w, h, spp, check_rows, oracle_spp=None):
    t0 = time.perf_counter()
    a = rt.Scene()
    build(a)
    info = a.flatten()
    t_host = time.perf_counter() - t0
    ctx = rt.Context(0)
    ctx.upload_scene(a)
    ctx.set_target(w, h)
    ctx.render(0, min(spp, 2), 50)   # warm-up launch
    ctx.sync()
    ctx.clear()
    ctx.render(0, spp, 50)
    ctx.sync()
    ms = ctx.last_render_ms()
    img = ctx.read_accum()
    print(f"[{name}] host build {t_host:.1f} s, kernel {ms:.2f} ms", file=sys.stderr, flush=True)
    cnt = ctx.count(0, min(spp, 4), 50)
    out = {"case": name, "size": [w, h], "spp": spp, "kernel_ms": round(ms, 3),
           "msamples_per_s": round(w * h * spp / ms / 1e3, 1), "host_build_s": round(t_host, 2),
           "n_nodes": info.n_nodes, "n_triangles": info.n_triangles, "n_spheres": info.n_spheres,
           "scene_in_lds": bool(info.fits_lds),
           "rays_per_sample": round(cnt["rays"] / cnt["samples"], 3),
           "node_tests_per_ray": round(cnt["node_tests"] / cnt["rays"], 2),
           "prim_tests_per_ray": round((cnt["sphere_tests"] + cnt["tri_tests"]) / cnt["rays"], 2),
           "fallbacks_per_ray": cnt["fallbacks"] / cnt["rays"]}
    if check_rows:
        b = orc.OracleScene()
        t0 = time.perf_counter()
        build(b)
        print(f"[{name}] oracle build {time.perf_counter() - t0:.1f} s", file=sys.stderr, flush=True)
        ospp = oracle_spp or spp
        if ospp != spp:
            ctx.clear()
            ctx.render(0, ospp, 50)
            ctx.sync()
            img = ctx.read_accum()
        ok = True
        for r0 in check_rows:
            want, _, _ = b.render(w, h, 0, ospp, 50, rng_mode=1, order=1, threads=16, row0=r0, row1=r0 + 2)
            ok = ok and np.array_equal(img[r0:r0 + 2, :, :3].view(np.uint32), want[r0:r0 + 2, :, :3].view(np.uint32))
        out["oracle_rows_bit_exact"] = bool(ok)
    ctx.close()
    print(json.dumps(out), flush=True)


def c1():
    run("C1 cover 1200x675x10", lambda s: scenes.cover_scene(s, 1200, 675), 1200, 675, 10, [0, 300, 600])


def c3():
    mesh = scenes.load_gltf_mesh(G)
    tex = scenes.load_gltf_textures(G)     # the vase's own three 4096^2 JPEG maps
    run("C3 cornell+vase (real 4096^2 maps) 1080p x256", lambda s: scenes.cornell_mesh_scene(s, 1920, 1080, mesh, tex),
        1920, 1080, 256, [100, 540, 900], oracle_spp=4)


def c4():
    run("C4 cover 3840x2160x1024", lambda s: scenes.cover_scene(s, 3840, 2160), 3840, 2160, 1024, [1080], oracle_spp=2)


def c5s(n_tris, w=1920, h=1080, spp=16, check=True):
    mesh = scenes.load_gltf_mesh(G)
    per = mesh["indices"].shape[0]
    copies = max(1, n_tris // per)
    tex = {"diffuse": scenes.noise_texture(1024, 1), "roughness": scenes.noise_texture(1024, 2), "normal": None}
    run(f"C5 {copies} vases = {copies * per} triangles, {w}x{h}x{spp}",
        lambda s: scenes.instanced_mesh_scene(s, mesh, copies, tex), w, h, spp, [h // 2] if check else [], oracle_spp=1)


def c5i(n_tris, w=1920, h=1080, spp=16, check=True):
    """The large-mesh scene with true instances (rt_scene_add_instance): ONE vase held in its own space, placed
    n_tris / 2,760 times.  Rendered along the reference-order entries (two levels)."""
    mesh = scenes.load_gltf_mesh(G)
    per = mesh["indices"].shape[0]
    copies = max(1, n_tris // per)
    tex = {"diffuse": scenes.noise_texture(1024, 1), "roughness": scenes.noise_texture(1024, 2), "normal": None}
    run(f"C5 placed: {copies} placements of one vase ({per} triangles held, {copies * per} shown), {w}x{h}x{spp}",
        lambda s: scenes.placed_mesh_scene(s, mesh, copies, tex), w, h, spp, [h // 2] if check else [], oracle_spp=1)


def build_compare(n_tris, w=1920, h=1080, spp=16):
    """Host (surface-area heuristic, threaded) against device (linear BVH) build of the search structure
    for the instanced-vase scene: seconds until the scene is resident, and the render time each gives."""
    mesh = scenes.load_gltf_mesh(G)
    per = mesh["indices"].shape[0]
    copies = max(1, n_tris // per)
    tex = {"diffuse": scenes.noise_texture(1024, 1), "roughness": scenes.noise_texture(1024, 2), "normal": None}
    out = {"case": f"search-structure build, {copies * per} triangles", "size": [w, h], "spp": spp}
    images = {}
    for mode in ("host", "device"):
        t0 = time.perf_counter()
        a = rt.Scene()
        scenes.instanced_mesh_scene(a, mesh, copies, tex)
        t_scene = time.perf_counter() - t0          # triangle list + reference-order BVH (host either way)
        ctx = rt.Context(0)
        ctx.set_device_build(mode == "device")
        t0 = time.perf_counter()
        ctx.upload_scene(a)                         # flatten (+ host build) + upload (+ device build)
        t_upload = time.perf_counter() - t0
        ctx.set_target(w, h)
        ctx.render(0, 2, 50)
        ctx.sync()
        ctx.clear()
        ctx.render(0, spp, 50)
        ctx.sync()
        ms = ctx.last_render_ms()
        images[mode] = ctx.read_accum()
        cnt = ctx.count(0, 2, 50)
        out[mode] = {"scene_and_reference_bvh_s": round(t_scene, 2), "flatten_build_upload_s": round(t_upload, 2),
                     "device_build_ms": round(ctx.last_build_ms(), 2), "kernel_ms": round(ms, 2),
                     "node_tests_per_ray": round(cnt["node_tests"] / cnt["rays"], 1),
                     "prim_tests_per_ray": round(cnt["tri_tests"] / cnt["rays"], 2),
                     "entries": int(ctx.read_search_structure()[0].shape[0])}
        ctx.close()
        print(f"[build {mode}] {out[mode]}", file=sys.stderr, flush=True)
    out["images_identical"] = bool(np.array_equal(images["host"].view(np.uint32), images["device"].view(np.uint32)))
    print(json.dumps(out), flush=True)


def heartbeat():
    import threading

    def tick():
        t0 = time.perf_counter()
        while True:
            time.sleep(45)
            print(f"... {time.perf_counter() - t0:.0f} s", file=sys.stderr, flush=True)
    threading.Thread(target=tick, daemon=True).start()


if __name__ == "__main__":
    heartbeat()   # long host-side BVH builds print nothing themselves
    args = sys.argv[1:] or ["c1", "c3"]
    for a in args:
        if a == "c1":
            c1()
        elif a == "c3":
            c3()
        elif a == "c4":
            c4()
        elif a.startswith("build"):
            build_compare(int(float(a.split(":")[1])) if ":" in a else 1_000_000)
        elif a.startswith("c5i"):
            parts = a.split(":")
            c5i(int(float(parts[1])) if len(parts) > 1 else 10_000_000, check=not (len(parts) > 2 and parts[2] == "nocheck"))
        elif a.startswith("c5s"):
            parts = a.split(":")   # c5s[:triangles[:nocheck]] -- nocheck skips the oracle rows (profiling runs)
            c5s(int(float(parts[1])) if len(parts) > 1 else 1_000_000, check=not (len(parts) > 2 and parts[2] == "nocheck"))
        elif a.startswith("c5"):
            c5s(int(float(a.split(":")[1])) if ":" in a else 10_000_000, 3840, 2160, 256)
