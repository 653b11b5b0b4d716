"""Instancing (rt_scene_begin_prototype / rt_scene_add_instance): an opt-in mode the reference does not have (it
bakes every placement into world-space triangles, R/src/Core/Mesh.cpp:106-208).  Its meaning is defined by the CPU
oracle's Instance class; the GPU follows it bit for bit.  There is no reference output to compare with."""
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
VASE = os.path.join(HERE, "golden", "brass_vase", "brass_vase_04_4k.gltf")


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def rotation_scale(angle_deg, axis, scale3, translate3):
    """Column-major T * R * S."""
    a = np.radians(angle_deg)
    x, y, z = np.asarray(axis, np.float64) / np.linalg.norm(axis)
    c, s = np.cos(a), np.sin(a)
    R = np.array([[c + x * x * (1 - c), x * y * (1 - c) - z * s, x * z * (1 - c) + y * s],
                  [y * x * (1 - c) + z * s, c + y * y * (1 - c), y * z * (1 - c) - x * s],
                  [z * x * (1 - c) - y * s, z * y * (1 - c) + x * s, c + z * z * (1 - c)]])
    M = np.eye(4)
    M[:3, :3] = R @ np.diag(scale3)
    M[:3, 3] = translate3
    return np.ascontiguousarray(M.T, dtype=np.float32).reshape(16)   # column-major


def three_placements(s):
    """A prototype of two spheres (glass, metal) and a two-triangle quad with smooth normals, placed three times --
    translated, rotated with a non-uniform scale, and mirrored -- beside a sphere and a triangle of the world's own,
    under one BVHNode(world)."""
    ground = s.add_material(0, (0.5, 0.5, 0.5), 0.0)
    glass = s.add_material(2, (0.0, 0.0, 0.0), 1.5)
    metal = s.add_material(1, (0.8, 0.6, 0.2), 0.1)
    red = s.add_material(0, (0.7, 0.2, 0.2), 0.0)
    light = s.add_material(3, (4.0, 4.0, 4.0), 0.0)
    s.add_sphere((0.0, -100.5, -1.0), 100.0, ground)
    s.add_sphere((0.0, 3.0, -1.0), 0.7, light)
    s.add_triangle([-3.0, -0.5, -4.0, 3.0, -0.5, -4.0, 0.0, 2.5, -4.0], red)
    p = s.begin_prototype()
    s.add_sphere((0.0, 0.0, 0.0), 0.5, glass)
    s.add_sphere((0.9, 0.1, 0.0), 0.3, metal)
    n = [0.0, 0.6, 0.8] * 3
    s.add_triangle([-0.6, -0.5, 0.4, 0.6, -0.5, 0.4, 0.6, 0.6, 0.4], red, n, [0, 0, 1, 0, 1, 1])
    s.add_triangle([-0.6, -0.5, 0.4, 0.6, 0.6, 0.4, -0.6, 0.6, 0.4], red, n, [0, 0, 1, 1, 0, 1])
    assert s.end_prototype() == p
    s.add_instance(p, rotation_scale(0.0, (0, 1, 0), (1, 1, 1), (-1.6, 0.0, -1.5)))
    s.add_instance(p, rotation_scale(37.0, (0.2, 1.0, 0.1), (1.3, 0.7, 1.0), (0.2, 0.1, -1.2)))
    s.add_instance(p, rotation_scale(190.0, (1.0, 0.3, 0.0), (-0.8, 0.8, 0.8), (1.7, 0.0, -0.9)))
    s.wrap_bvh()
    s.set_camera((0.3, 1.0, 2.5), (0.0, 0.0, -1.0), (0.0, 1.0, 0.0), 50.0, 16.0 / 9.0, 0.05, 3.5)
    s.set_background((0.7, 0.8, 1.0))


def test_flatten_of_a_scene_with_instances(rt):
    s = rt.Scene()
    three_placements(s)
    info = s.flatten()
    flat = s.get_flat()
    nodes = flat["nodes"].reshape(-1, 8)
    marked = np.flatnonzero(nodes[:, 3] & np.uint32(1 << 29))
    assert len(marked) == 3 and sorted(nodes[marked, 7].tolist()) == [0, 1, 2]      # one entry per placement
    assert info.n_spheres == 4 and info.n_triangles == 3                            # the prototype is held once
    stride, _ = s.search_structure_octants()
    assert stride == 0                                                              # no search structure


def test_misuse_is_refused(rt):
    s = rt.Scene()
    m = s.add_material(0, (0.5, 0.5, 0.5), 0.0)
    ident = rotation_scale(0.0, (0, 1, 0), (1, 1, 1), (0, 0, 0))
    with pytest.raises(rt.RtError):
        s.add_instance(0, ident)                 # no such prototype
    p = s.begin_prototype()
    with pytest.raises(rt.RtError):
        s.begin_prototype()                      # no nesting
    with pytest.raises(rt.RtError):
        s.add_instance(p, ident)                 # not finished yet
    assert s.end_prototype() == p
    with pytest.raises(rt.RtError):
        s.add_instance(p, ident)                 # empty
    with pytest.raises(rt.RtError):
        s.end_prototype()
    q = s.begin_prototype()
    s.add_sphere((0.0, 0.0, 0.0), 1.0, m)
    s.end_prototype()
    bad = ident.copy()
    bad[5] = np.nan
    with pytest.raises(rt.RtError):
        s.add_instance(q, bad)
    assert s.add_instance(q, ident) == 0


@pytest.mark.gpu
def test_three_placements_trace_and_render(rt, orc, gpu_ctx):
    a, b = rt.Scene(), orc.OracleScene()
    three_placements(a)
    three_placements(b)
    gpu_ctx.upload_scene(a)
    rng = np.random.default_rng(5)
    n = 40000
    rays = np.zeros((n, 6), np.float32)
    rays[:, :3] = rng.uniform(-2.5, 2.5, (n, 3)) + np.array([0.0, 0.5, 1.0])
    rays[:, 3:] = rng.normal(size=(n, 3))
    k1, p1, t1, n1, f1, uv1 = gpu_ctx.trace(rays)
    k2, p2, t2, n2, f2, uv2 = b.trace(rays)
    assert (k1 >= 0).sum() > n // 4 and (k1 == 1).sum() > 500
    assert np.array_equal(k1, k2) and np.array_equal(p1, p2) and np.array_equal(bits(t1), bits(t2))
    assert np.array_equal(bits(n1), bits(n2))
    # two HitRecord fields the reference leaves unwritten are compared where it writes them: frontFace for spheres
    # and flat triangles (the prototype's quad has vertex normals, R/src/Core/Hittable.h:93-96), (u, v) for triangles
    tri = k1 == 1
    front_defined = (k1 == 0) | (tri & (p1 == 0))
    assert np.array_equal(f1[front_defined], f2[front_defined])
    assert np.array_equal(bits(uv1)[tri], bits(uv2)[tri])
    w, h, spp = 192, 108, 3
    gpu_ctx.set_target(w, h)
    gpu_ctx.render(0, spp, 50)
    gpu_ctx.sync()
    got = gpu_ctx.read_accum()
    want, want8, _ = b.render(w, h, 0, spp, 50, rng_mode=1, order=1, threads=8)
    assert got[..., :3].max() > 0
    assert np.array_equal(bits(got[..., :3]), bits(want[..., :3]))
    assert np.array_equal(gpu_ctx.read_rgba8(spp), want8)
    # path counters of the two-level walk equal the oracle's
    cnt = gpu_ctx.count(0, 1, 50)
    _, _, ocnt = b.render(w, h, 0, 1, 50, rng_mode=1, order=1, threads=1, want_counters=True)
    for key in ("rays", "node_tests", "sphere_tests", "tri_tests", "hits", "misses"):
        assert cnt[key] == ocnt[key], key
    # a rank of a split frame renders its tiles of the same image
    gpu_ctx.set_target(w, h, 1, 2)
    gpu_ctx.render(0, spp, 50)
    gpu_ctx.sync()
    part = gpu_ctx.read_accum()
    own = part[..., :3] != 0
    assert own.any() and np.array_equal(bits(part[..., :3][own]), bits(got[..., :3][own]))


@pytest.mark.gpu
def test_vases_placed_instead_of_baked(rt, orc, scenes, gpu_ctx):
    """The large-mesh configuration with true instances: 3,623 placements of ONE brass vase (2,760 triangles held
    once, against 9,999,480 baked).  GPU rows equal the oracle's bit for bit; the image is the baked scene's to
    within what the different trees and arithmetic allow (same camera, same placements)."""
    mesh = scenes.load_gltf_mesh(VASE)
    tex = {"diffuse": scenes.noise_texture(256, 1), "roughness": scenes.noise_texture(256, 2)}
    w, h, copies = 960, 540, 3623
    a, b = rt.Scene(), orc.OracleScene()
    assert scenes.placed_mesh_scene(a, mesh, copies, tex) == 2760
    info = a.flatten()
    assert info.n_triangles == 2760 + 12
    gpu_ctx.upload_scene(a)
    gpu_ctx.set_target(w, h)
    gpu_ctx.render(0, 1, 50)
    gpu_ctx.sync()
    got = gpu_ctx.read_accum()
    assert np.isfinite(got).all() and got[..., :3].max() > 0
    scenes.placed_mesh_scene(b, mesh, copies, tex)
    for row0 in (100, 270, 450):
        want, _, _ = b.render(w, h, 0, 1, 50, rng_mode=1, order=1, threads=16, row0=row0, row1=row0 + 2)
        assert np.array_equal(bits(got[row0:row0 + 2, :, :3]), bits(want[row0:row0 + 2, :, :3])), row0
    # against the baked scene (its own trees, world-space arithmetic): the same picture, not the same bits
    c = rt.Scene()
    scenes.instanced_mesh_scene(c, mesh, 300, tex)
    d = rt.Scene()
    scenes.placed_mesh_scene(d, mesh, 300, tex)
    imgs = []
    for sc in (c, d):
        gpu_ctx.upload_scene(sc)
        gpu_ctx.set_target(320, 180)
        gpu_ctx.render(0, 16, 50)
        gpu_ctx.sync()
        imgs.append(gpu_ctx.read_accum()[..., :3] / 16.0)
    assert abs(imgs[0].mean() - imgs[1].mean()) < 0.02 * imgs[0].mean()
