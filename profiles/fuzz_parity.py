"""Randomised parity run: many small random worlds (spheres of all four materials, loose triangles,
lathe meshes with smooth or flat normals and PBR (a third of them normal-mapped) or Lambertian materials -- baked, or in
every fourth world held once and placed by transforms --, lights), random cameras
(inside and outside the world, with and without a lens), with and without the reference's top-level
BVH; every accumulation image of the GPU path must equal the CPU oracle's bit for bit.

Usage (GPU box): python profiles/fuzz_parity.py [first_seed] [n_seeds] [device]   (device: the search structure is built on the GPU)
Prints one line per mismatch and a summary; exit code 1 if anything differed.
"""
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


def make_builder(seed):
    rr = np.random.RandomState(seed)
    n_sph = int(rr.choice([0, 1, 3, 20, 120]))
    if seed % 10 == 0:
        n_sph = int(rr.choice([3000, 9000]))  # too many for LDS: the tables are read from global memory
    n_tri = int(rr.choice([0, 1, 7, 40]))
    n_mesh = int(rr.choice([0, 0, 1, 2]))
    if n_sph + n_tri + n_mesh == 0:
        n_sph = 2
    extent = float(rr.choice([1.0, 30.0, 600.0]))
    sph = [((rr.rand(3) * extent).astype(np.float32), float(np.float32(extent * (0.01 + 0.12 * rr.rand())))) for _ in range(n_sph)]
    if n_sph and rr.rand() < 0.4:   # a ground sphere far larger than the rest
        sph.append((np.array([extent / 2, -50 * extent, extent / 2], np.float32), float(np.float32(50 * extent))))
    if n_sph > 2 and rr.rand() < 0.3:  # a duplicate: coincident primitives, decided by visit order
        sph.append(sph[0])
    shell = None
    if seed % 5 == 1:
        # everything inside one big sphere (glass or mirror, chosen below): paths bounce off its inside and come
        # back to the small primitives from far away, where the reference's hit tests are least accurate
        shell = (np.array([extent / 2] * 3, np.float32), float(np.float32(extent * float(rr.choice([3.0, 20.0, 150.0])))),
                 int(rr.choice([1, 2])))
    tris = [(rr.rand(9) * extent).astype(np.float32) for _ in range(n_tri)]
    meshes = []
    for _ in range(n_mesh):
        m = scenes.lathe_mesh(int(rr.randint(4, 14)), int(rr.randint(3, 9)), seed=int(rr.randint(1 << 30)))
        model = scenes.translate_scale(tuple((rr.rand(3) * extent * 0.5).astype(np.float32)),
                                       (extent * float(0.5 + rr.rand()),) * 3)
        meshes.append((m, model, int(rr.choice([-1, 0, 1, 3])), rr.rand() < 0.5, rr.rand() < 0.5, rr.rand() < 0.5))
    # every fourth world places its meshes as instances (held once in their own space, 1-4 placements each by a
    # rotation, an uneven scale and a shift) instead of baking them
    placed = seed % 4 == 2 and n_mesh > 0
    placements = []
    if placed:
        r2 = np.random.RandomState(seed + 991)
        for _ in range(n_mesh):
            ms = []
            for _ in range(int(r2.randint(1, 5))):
                axis = r2.randn(3)
                axis /= np.linalg.norm(axis)
                ang = float(r2.rand() * 2 * np.pi)
                K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
                R = np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * (K @ K)
                M = np.eye(4)
                M[:3, :3] = R @ np.diag(0.5 + r2.rand(3))
                M[:3, 3] = (r2.rand(3) - 0.25) * extent * 0.5
                ms.append(M.astype(np.float32).T.reshape(16).copy())  # column-major
            placements.append(ms)
    mat_draws = rr.rand(64, 5)
    wrap = rr.rand() < 0.6
    inside = rr.rand() < 0.5
    lookfrom = ((rr.rand(3) * extent) if inside else (rr.rand(3) * 4 - 1.5) * extent).astype(np.float32)
    lookat = (rr.rand(3) * extent).astype(np.float32)
    if seed % 7 == 0:
        # looking straight along an axis from a point with equal coordinates: direction components that
        # are exactly zero for the centre rays (the reference-order walk takes those)
        lookfrom = np.array([extent * 0.5, extent * 0.5, -extent], np.float32)
        lookat = np.array([extent * 0.5, extent * 0.5, extent], np.float32)
    aperture = float(rr.choice([0.0, 0.1 * extent / 30.0]))
    if seed % 7 == 0:
        aperture = 0.0
    vfov = float(rr.choice([20.0, 40.0, 90.0]))
    bg = tuple(float(x) for x in rr.choice([0.0, 0.7, 1.0], 3))
    tex_seed = int(rr.randint(1 << 20))
    # PBR meshes carry a normal map, and the scene's switch is on (not with placements: the library refuses that)
    normal_maps = seed % 3 == 0 and not placed

    def build(s):
        mats = []
        for d in mat_draws[:12]:
            kind = int(d[0] * 4)  # 0 Lambertian, 1 Metal, 2 Dielectric, 3 light
            param = float(np.float32(d[4] * (1.2 if kind == 1 else 1.0) + (1.0 if kind == 2 else 0.0)))
            col = tuple(float(np.float32(x * (8.0 if kind == 3 else 1.0))) for x in d[1:4])
            mats.append(s.add_material(kind, col, param))
        for i, (c, r) in enumerate(sph):
            s.add_sphere(c, r, mats[i % len(mats)])
        if shell is not None:
            s.add_sphere(shell[0], shell[1], s.add_material(shell[2], (0.9, 0.9, 0.9), 0.0 if shell[2] == 1 else 1.5))
        for i, p in enumerate(tris):
            s.add_triangle(p, mats[(i * 5 + 1) % len(mats)])
        for k, (m, model, depth, smooth, pbr, rough) in enumerate(meshes):
            if pbr:
                td = s.add_texture(scenes.noise_texture(16, tex_seed + k))
                tr = s.add_texture(scenes.noise_texture(16, tex_seed + 100 + k)) if rough else -1
                tn = s.add_texture(scenes.noise_texture(24, tex_seed + 200 + k)) if normal_maps else -1
                mat = s.add_pbr(td, tr, tn)
            else:
                mat = mats[k % len(mats)]
            if placed:
                proto = s.begin_prototype()
            s.add_mesh(model, m["positions"], m["indices"], mat, normals=m["normals"] if smooth else None,
                       uvs=m["uvs"] if pbr else None, max_depth=depth)
            if placed:
                s.end_prototype()
                for M in placements[k]:
                    s.add_instance(proto, M)
        if wrap:
            s.wrap_bvh()
        s.set_camera(lookfrom, lookat, (0.0, 1.0, 0.0), vfov, 1.0, aperture, float(max(1e-3, np.linalg.norm(lookat - lookfrom))))
        s.set_background(bg)
        s.set_normal_mapping(normal_maps)

    return build


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    ctx = rt.Context(0)
    if len(sys.argv) > 3 and sys.argv[3] == "device":
        ctx.set_device_build(True)
    bad = 0
    t0 = time.perf_counter()
    fallbacks = rays = hits = 0
    for seed in range(first, first + count):
        build = make_builder(seed)
        a, b = rt.Scene(), orc.OracleScene()
        build(a)
        build(b)
        rr = np.random.RandomState(seed + 7)
        w, h = int(rr.randint(2, 97)), int(rr.randint(2, 71))
        spp, depth = int(rr.randint(1, 5)), int(rr.choice([1, 3, 50]))
        ctx.upload_scene(a)
        ctx.set_target(w, h)
        ctx.render(0, spp, depth)
        ctx.sync()
        got = ctx.read_accum()
        want, _, _ = b.render(w, h, 0, spp, depth, rng_mode=1, order=1, threads=8)
        same = np.array_equal(got[..., :3].view(np.uint32), want[..., :3].view(np.uint32))
        c = ctx.count(0, spp, depth)
        fallbacks += c["fallbacks"]
        rays += c["rays"]
        hits += c["hits"]
        if not same:
            bad += 1
            n = int(np.count_nonzero((got[..., :3].view(np.uint32) != want[..., :3].view(np.uint32)).any(axis=-1)))
            print(f"seed {seed}: {n} of {w * h} pixels differ ({w}x{h}, {spp} spp, depth {depth})", flush=True)
        if (seed - first) % 25 == 24:
            print(f"... {seed - first + 1} worlds, {bad} mismatches, {time.perf_counter() - t0:.0f} s", flush=True)
    print(f"{count} random worlds from seed {first}: {bad} mismatches; {rays} rays, {hits} hits, {fallbacks} re-traced in reference order")
    ctx.close()
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
