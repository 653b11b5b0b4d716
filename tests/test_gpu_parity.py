"""Parity of the HIP path with the CPU oracle, through the C ABI.  All of these need a GPU.

Bar: bit-exact.  Hit indices, t, normals, scatter outputs and whole accumulation images
(hash RNG stream, front-to-back throughput order) must equal the oracle's to the last bit.
The one documented deviation from the reference's own evaluation order -- the attenuation
product is formed front to back instead of inside-out -- is bounded separately (rel 2e-6).
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
G = os.path.join(HERE, "golden")


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def both(rt, orc, builder):
    a, b = rt.Scene(), orc.OracleScene()
    builder(a)
    builder(b)
    return a, b


def rays_cover(n, seed):
    from golden.make_golden import golden_rays
    return golden_rays(n, seed)


def rays_cornell(n, seed):
    from golden.make_golden import golden_rays
    return golden_rays(n, seed, target=(278, 250, 278), spread=400.0, origin_box=((100, 100, -800), (450, 450, -100)))


def assert_trace_equal(ctx, oscene, rays, smooth_tri=None):
    """Hit kind, id, t and normal must be identical everywhere.  Two HitRecord fields are left
    unwritten by the reference and are compared only where it writes them: (u, v) is written by
    Triangle::Hit only (Sphere::Hit leaves whatever an earlier candidate stored), and frontFace
    is not written for triangles with vertex normals (R/src/Core/Hittable.h:93-96)."""
    k, p, t, n, f, uv = ctx.trace(rays)
    ok, op, ot, on, of, ouv = oscene.trace(rays)
    assert np.array_equal(k, ok)
    assert np.array_equal(p, op)
    assert np.array_equal(bits(t), bits(ot))
    assert np.array_equal(bits(n), bits(on))
    tri = k == 1
    front_defined = k == 0
    if smooth_tri is not None:
        front_defined |= tri & ~smooth_tri[np.where(tri, p, 0)]
    else:
        front_defined |= tri
    assert np.array_equal(f[front_defined], of[front_defined])
    assert np.array_equal(bits(uv)[tri], bits(ouv)[tri])
    return k


def smooth_flags(scene):
    """True for triangles whose first vertex normal is non-zero (interpolated shading normal)."""
    return (scene.get_flat()["tri_normal"][:, 0:3] != 0).any(axis=1)


@pytest.mark.parametrize("wrap", [True, False])
def test_trace_cover_scene(rt, orc, scenes, gpu_ctx, wrap):
    a, b = both(rt, orc, lambda s: scenes.cover_scene(s, 1920, 1080, wrap_bvh=wrap))
    gpu_ctx.upload_scene(a)
    rays = rays_cover(50000, 7)
    k = assert_trace_equal(gpu_ctx, b, rays)
    assert 0.2 < (k >= 0).mean() < 1.0
    if wrap:  # and against the committed golden table
        g = np.load(os.path.join(G, "cover_hits.npz"))
        k, p, t, n, f, _ = gpu_ctx.trace(rays_cover(4096, 1234))
        assert np.array_equal(k, g["kind"]) and np.array_equal(p, g["pid"]) and np.array_equal(bits(t), g["t"])
        assert np.array_equal(bits(n), g["normal"]) and np.array_equal(f, g["front"])


def test_trace_edge_rays(rt, orc, scenes, gpu_ctx):
    """Axis-parallel directions (infinite inverse components), rays starting on surfaces,
    grazing rays and rays that start inside spheres."""
    a, b = both(rt, orc, lambda s: scenes.cover_scene(s, 1920, 1080, wrap_bvh=True))
    gpu_ctx.upload_scene(a)
    r = np.random.RandomState(3)
    rays = []
    for _ in range(2000):
        o = (r.rand(3) * [20, 3, 20] - [10, 0, 10]).astype(np.float32)
        axis = r.randint(3)
        d = np.zeros(3, np.float32)
        d[axis] = r.choice([-1.0, 1.0]) * (0.5 + r.rand())
        if r.rand() < 0.5:
            d[(axis + 1) % 3] = r.choice([-1.0, 1.0]) * r.rand()
        rays.append(np.concatenate([o, d]))
    for c in ([0, 1, 0], [-4, 1, 0], [4, 1, 0]):          # from the centres of the big spheres
        for _ in range(200):
            rays.append(np.concatenate([np.asarray(c, np.float32), (r.rand(3) - 0.5).astype(np.float32)]))
    for _ in range(500):                                   # from the ground plane's surface, grazing
        x, z = r.rand(2) * 16 - 8
        d = np.array([r.rand() - 0.5, 1e-3 * r.rand(), r.rand() - 0.5], np.float32)
        rays.append(np.concatenate([np.array([x, 0.0, z], np.float32), d]))
    assert_trace_equal(gpu_ctx, b, np.asarray(rays, np.float32))


def test_trace_cornell_vase(rt, orc, scenes, gpu_ctx):
    mesh = scenes.load_gltf_mesh(os.path.join(G, "brass_vase", "brass_vase_04_4k.gltf"))
    tex = {"diffuse": scenes.noise_texture(64, 1), "roughness": scenes.noise_texture(64, 2), "normal": None}
    a, b = both(rt, orc, lambda s: scenes.cornell_mesh_scene(s, 640, 360, mesh, tex))
    gpu_ctx.upload_scene(a)
    k = assert_trace_equal(gpu_ctx, b, rays_cornell(50000, 21), smooth_flags(a))
    assert (k == 1).mean() > 0.9
    g = np.load(os.path.join(G, "vase_scene.npz"))
    k, p, t, n, f, uv = gpu_ctx.trace(rays_cornell(4096, 99))
    assert np.array_equal(k, g["kind"]) and np.array_equal(p, g["pid"]) and np.array_equal(bits(t), g["t"])
    assert np.array_equal(bits(n), g["normal"]) and np.array_equal(bits(uv), g["uv"])


@pytest.mark.parametrize("max_depth", [0, 2, -1])
def test_trace_mixed_world(rt, orc, scenes, gpu_ctx, max_depth):
    """Spheres + loose triangles + a mesh (list leaves at max_depth 0) under a top-level BVH."""
    mesh = scenes.lathe_mesh(12, 9, seed=4)

    def build(s):
        m = [s.add_material(0, (0.5, 0.4, 0.3)), s.add_material(1, (0.9, 0.8, 0.7), 0.2), s.add_material(2, (0, 0, 0), 1.5)]
        rr = np.random.RandomState(5)
        for i in range(40):
            s.add_sphere((rr.rand(3) * 400 + 50).astype(np.float32), float(np.float32(5 + 30 * rr.rand())), m[i % 3])
        for i in range(9):
            s.add_triangle((rr.rand(9) * 500).astype(np.float32), m[i % 2])
        td = s.add_texture(scenes.noise_texture(8, 1))
        s.add_mesh(scenes.VASE_MODEL, mesh["positions"], mesh["indices"], s.add_pbr(td, -1, -1),
                   normals=mesh["normals"], uvs=mesh["uvs"], max_depth=max_depth)
        s.wrap_bvh()

    a, b = both(rt, orc, build)
    gpu_ctx.upload_scene(a)
    assert_trace_equal(gpu_ctx, b, rays_cornell(30000, 5), smooth_flags(a))


def test_scatter_table(rt, orc, scenes, gpu_ctx):
    """Material::scatter for every kind with injected hit records and RNG positions."""
    def build(s):
        s.add_material(0, (0.8, 0.3, 0.2))
        s.add_material(1, (0.9, 0.8, 0.7), 0.35)
        s.add_material(1, (0.7, 0.6, 0.5), 0.0)
        s.add_material(2, (0, 0, 0), 1.5)
        s.add_material(3, (15, 15, 15))
        td, tr = s.add_texture(scenes.noise_texture(32, 1)), s.add_texture(scenes.noise_texture(32, 2))
        s.add_pbr(td, tr, -1)
        s.add_pbr(td, -1, -1)
        s.add_sphere((0, 0, 0), 1.0, 0)

    a, b = both(rt, orc, build)
    gpu_ctx.upload_scene(a)
    r = np.random.RandomState(17)
    n = 4000
    mat = r.randint(0, 7, n).astype(np.int32)
    nrm = r.randn(n, 3).astype(np.float32)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True).astype(np.float32)
    din = r.randn(n, 3).astype(np.float32) * np.float32(3.0)
    flip = (np.sum(din * nrm, axis=1) > 0)
    din[flip] *= -1                                   # incoming rays face the normal
    p = (r.rand(n, 3) * 10).astype(np.float32)
    ray_in = np.concatenate([p - din, din], axis=1).astype(np.float32)
    front = r.randint(0, 2, n).astype(np.int32)
    uv = (0.02 + 0.96 * r.rand(n, 2)).astype(np.float32)
    key = np.stack([r.randint(0, 1920, n), r.randint(0, 1080, n), r.randint(0, 64, n), r.randint(0, 300, n)], 1).astype(np.uint32)
    att, out, alive, cnt = gpu_ctx.scatter_test(mat, ray_in, p, nrm, front, uv, key)
    for i in range(n):
        oa, oo, oal, oc = b.scatter(int(mat[i]), ray_in[i], p[i], nrm[i], front[i], uv[i, 0], uv[i, 1],
                                    key[i, 0], key[i, 1], key[i, 2], key[i, 3])
        assert alive[i] == oal, i
        assert cnt[i] == oc, i
        if mat[i] == 4:   # a light neither attenuates nor scatters; outputs are not defined
            continue
        assert np.array_equal(bits(att[i]), bits(oa)), (i, mat[i])
        assert np.array_equal(bits(out[i]), bits(oo)), (i, mat[i])
    assert set(np.unique(mat)) == set(range(7))


def render_gpu(ctx, w, h, first, n, depth=50):
    ctx.set_target(w, h)
    ctx.render(first, n, depth)
    ctx.sync()
    return ctx.read_accum()


@pytest.mark.parametrize("wrap", [True, False])
def test_render_cover_bit_exact(rt, orc, scenes, gpu_ctx, wrap):
    w, h, spp = 200, 112, 3
    a, b = both(rt, orc, lambda s: scenes.cover_scene(s, w, h, wrap_bvh=wrap))
    gpu_ctx.upload_scene(a)
    got = render_gpu(gpu_ctx, w, h, 0, spp)
    want, want8, _ = b.render(w, h, 0, spp, 50, rng_mode=1, order=1, threads=8)
    assert np.array_equal(bits(got[..., :3]), bits(want[..., :3]))
    assert (got[..., 3] == 1.0).all()
    assert np.array_equal(gpu_ctx.read_rgba8(spp), want8)
    # the reference's inside-out product order: same paths, last-bit differences only
    ref, _, _ = b.render(w, h, 0, spp, 50, rng_mode=1, order=0, threads=8)
    rel = np.abs(got[..., :3] - ref[..., :3]) / np.maximum(np.abs(ref[..., :3]), 1e-6)
    assert rel.max() < 2e-6
    if wrap:
        g = np.load(os.path.join(G, "cover_hash_64x36x4.npz"))
        a2 = rt.Scene()
        scenes.cover_scene(a2, 64, 36, wrap_bvh=True)
        gpu_ctx.upload_scene(a2)
        got = render_gpu(gpu_ctx, 64, 36, 0, 4)
        assert np.array_equal(bits(got[..., :3]), g["throughput"].reshape(36, 64, 4)[..., :3])


def test_render_ragged_size_and_depths(rt, orc, scenes, gpu_ctx):
    """Width and height that are not multiples of the 8x8 tile; max depth 0 (rayColor returns black before any
    ray is cast, R/src/Core/Raytracer.cpp:5-6), 1, 2 and 5."""
    w, h = 101, 59
    a, b = both(rt, orc, lambda s: scenes.cover_scene(s, w, h, wrap_bvh=True))
    gpu_ctx.upload_scene(a)
    for depth in (0, 1, 2, 5):
        got = render_gpu(gpu_ctx, w, h, 0, 2, depth)
        want, _, _ = b.render(w, h, 0, 2, depth, rng_mode=1, order=1, threads=8)
        assert np.array_equal(bits(got[..., :3]), bits(want[..., :3])), depth


def test_empty_world_and_smallest_frame(rt, orc, scenes, gpu_ctx):
    """No objects at all: every path ends on the background colour; and a 2x2 frame (the smallest
    the reference's `x / (W - 1)` pixel mapping allows)."""
    def build(s):
        s.set_camera((0.0, 0.0, 0.0), (0.0, 0.0, -1.0), (0.0, 1.0, 0.0), 60.0, 1.0, 0.1, 1.0)
        s.set_background((0.25, 0.5, 0.75))
    a, b = both(rt, orc, build)
    gpu_ctx.upload_scene(a)
    got = render_gpu(gpu_ctx, 33, 17, 0, 3)
    want, _, _ = b.render(33, 17, 0, 3, 50, rng_mode=1, order=1, threads=2)
    assert np.array_equal(bits(got[..., :3]), bits(want[..., :3]))
    assert np.array_equal(bits(got[..., :3]), bits(np.broadcast_to(np.float32([0.75, 1.5, 2.25]), (17, 33, 3))))
    k, p, t, n, f, uv = gpu_ctx.trace(np.float32([[0, 0, 0, 0, 0, -1]]))
    assert k[0] == -1 and np.isinf(t[0])
    a, b = both(rt, orc, lambda s: scenes.cover_scene(s, 2, 2))
    gpu_ctx.upload_scene(a)
    got = render_gpu(gpu_ctx, 2, 2, 0, 8)
    want, _, _ = b.render(2, 2, 0, 8, 50, rng_mode=1, order=1, threads=1)
    assert np.array_equal(bits(got[..., :3]), bits(want[..., :3]))


def test_coincident_primitives_resolve_like_the_reference(rt, orc, scenes, gpu_ctx):
    """Exactly equal hit distances: the reference keeps the primitive it visits LAST (its rejection
    test is `tMax < t`).  Coincident spheres and coincident triangles with different materials,
    flat list and BVH-wrapped, searched in any order by the kernel's own structure."""
    for wrap in (False, True):
        def build(s):
            m = [s.add_material(0, (0.1 * (i + 1), 0.5, 0.9 - 0.1 * i)) for i in range(6)]
            rr = np.random.RandomState(11)
            for i in range(12):
                c = (rr.rand(3) * 10 - 5).astype(np.float32)
                rad = float(np.float32(0.5 + rr.rand()))
                for j in range(3):                      # three copies of every sphere
                    s.add_sphere(c, rad, m[(i + j) % 6])
            for i in range(8):
                tri = (rr.rand(9) * 10 - 5).astype(np.float32)
                for j in range(2):                      # two copies of every triangle
                    s.add_triangle(tri, m[(i + 3 * j) % 6])
            if wrap:
                s.wrap_bvh()
            s.set_camera((0.0, 0.0, 14.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), 50.0, 1.0, 0.0, 10.0)
            s.set_background((0.7, 0.8, 1.0))
        a, b = both(rt, orc, build)
        gpu_ctx.upload_scene(a)
        from golden.make_golden import golden_rays
        rays = golden_rays(20000, 3, target=(0, 0, 0), spread=6.0, origin_box=((-8, -8, 9), (8, 8, 14)))
        k = assert_trace_equal(gpu_ctx, b, rays)
        assert (k >= 0).mean() > 0.3
        got = render_gpu(gpu_ctx, 96, 64, 0, 2)
        want, _, _ = b.render(96, 64, 0, 2, 50, rng_mode=1, order=1, threads=4)
        assert np.array_equal(bits(got[..., :3]), bits(want[..., :3])), wrap


def test_huge_coordinates_take_the_reference_walk(rt, orc, scenes, gpu_ctx):
    """Coordinates beyond the range the padded search structure is specified for (1e7): the
    library switches to the reference-order walk by itself and results stay bit-exact."""
    def build(s):
        m = s.add_material(0, (0.5, 0.5, 0.5))
        s.add_sphere((0.0, -3.0e7, 0.0), 3.0e7, m)
        s.add_sphere((0.0, 1.0, 0.0), 1.0, s.add_material(1, (0.8, 0.8, 0.8), 0.1))
        s.add_sphere((2.5e7, 1.0, 0.0), 4.0e6, s.add_material(2, (0, 0, 0), 1.5))
        s.wrap_bvh()
        s.set_camera((0.0, 2.0, 8.0), (0.0, 1.0, 0.0), (0.0, 1.0, 0.0), 40.0, 1.0, 0.0, 8.0)
        s.set_background((0.7, 0.8, 1.0))
    a, b = both(rt, orc, build)
    gpu_ctx.upload_scene(a)
    got = render_gpu(gpu_ctx, 80, 48, 0, 2)
    want, _, _ = b.render(80, 48, 0, 2, 50, rng_mode=1, order=1, threads=4)
    assert np.array_equal(bits(got[..., :3]), bits(want[..., :3]))
    cnt = gpu_ctx.count(0, 2, 50)
    _, _, ocnt = b.render(80, 48, 0, 2, 50, rng_mode=1, order=1, threads=4, want_counters=True)
    assert cnt["fallbacks"] == 0 and cnt["node_tests"] == ocnt["node_tests"]   # reference policy throughout


def test_cornell_box_lights_only(rt, orc, scenes, gpu_ctx):
    """cornellBox() alone (R/src/RaytracingApplication.cpp:407-432): triangles, the emitting material, black sky."""
    def build(s):
        s.cornell_box()
        s.set_camera((278.0, 278.0, -800.0), (278.0, 278.0, 0.0), (0.0, 1.0, 0.0), 40.0, 1.0, 0.0, 10.0)
        s.set_background((0.0, 0.0, 0.0))
    a, b = both(rt, orc, build)
    gpu_ctx.upload_scene(a)
    got = render_gpu(gpu_ctx, 120, 120, 0, 4)
    want, _, ocnt = b.render(120, 120, 0, 4, 50, rng_mode=1, order=1, threads=8, want_counters=True)
    assert np.array_equal(bits(got[..., :3]), bits(want[..., :3]))
    assert got[..., :3].max() > 0 and ocnt["scatter_light"] > 0
    gpu_ctx.set_reference_traversal(True)
    try:
        cnt = gpu_ctx.count(0, 4, 50)
    finally:
        gpu_ctx.set_reference_traversal(False)
    assert {k: cnt[k] for k in ocnt} == ocnt


def test_colour_buffer_budget_cuts_launches_without_changing_bits(rt, scenes, gpu_ctx, monkeypatch):
    """A render call whose per-sample colour buffer would exceed the budget is run as several
    launches over consecutive sample ranges; the ordered per-pixel sum gives the same bits."""
    w, h, spp = 200, 112, 100
    a = rt.Scene()
    scenes.cover_scene(a, w, h)
    gpu_ctx.upload_scene(a)
    whole = render_gpu(gpu_ctx, w, h, 0, spp)
    monkeypatch.setenv("RT_COLORS_MB", "16")     # 16 MB / (22,400 pixel slots x 12 B) = 62 samples per launch
    cut = render_gpu(gpu_ctx, w, h, 0, spp)
    assert np.array_equal(bits(cut), bits(whole))


def test_progressive_accumulation_is_split_invariant(rt, scenes, gpu_ctx):
    w, h = 160, 90
    a = rt.Scene()
    scenes.cover_scene(a, w, h)
    gpu_ctx.upload_scene(a)
    whole = render_gpu(gpu_ctx, w, h, 0, 6)
    gpu_ctx.set_target(w, h)
    for first, n in ((0, 1), (1, 2), (3, 3)):
        gpu_ctx.render(first, n, 50)
    gpu_ctx.sync()
    assert np.array_equal(bits(gpu_ctx.read_accum()), bits(whole))
    gpu_ctx.clear()
    gpu_ctx.sync()
    assert not gpu_ctx.read_accum()[..., :3].any()
    # rt_run = clear + passes (+ callback after each)
    seen = []
    gpu_ctx.run(6, 4, 50, on_pass=seen.append)
    assert seen == [4, 6]
    assert np.array_equal(bits(gpu_ctx.read_accum()), bits(whole))


def test_render_cornell_vase_bit_exact(rt, orc, scenes, gpu_ctx):
    w, h, spp = 96, 54, 2
    mesh = scenes.load_gltf_mesh(os.path.join(G, "brass_vase", "brass_vase_04_4k.gltf"))
    tex = {"diffuse": scenes.noise_texture(64, 1), "roughness": scenes.noise_texture(64, 2), "normal": None}
    a, b = both(rt, orc, lambda s: scenes.cornell_mesh_scene(s, w, h, mesh, tex))
    gpu_ctx.upload_scene(a)
    got = render_gpu(gpu_ctx, w, h, 0, spp)
    want, _, _ = b.render(w, h, 0, spp, 50, rng_mode=1, order=1, threads=8)
    assert np.array_equal(bits(got[..., :3]), bits(want[..., :3]))
    assert got[..., :3].max() > 0          # the light is seen
    g = np.load(os.path.join(G, "vase_scene.npz"))
    got = render_gpu(gpu_ctx, 64, 36, 0, 2)
    # golden was rendered with the 64x36 camera (same aspect: 64 // 36 == 96 // 54 == 1)
    assert np.array_equal(bits(got[..., :3]), g["accum"].reshape(36, 64, 4)[..., :3])


def test_render_lambertian_mesh_without_roughness(rt, orc, scenes, gpu_ctx):
    w, h = 64, 48
    mesh = scenes.lathe_mesh(16, 12, seed=2)
    tex = {"diffuse": scenes.noise_texture(32, 3), "roughness": None, "normal": scenes.noise_texture(32, 4)}
    a, b = both(rt, orc, lambda s: scenes.cornell_mesh_scene(s, w, h, mesh, tex, max_depth=3))
    gpu_ctx.upload_scene(a)
    got = render_gpu(gpu_ctx, w, h, 0, 3)
    want, _, _ = b.render(w, h, 0, 3, 50, rng_mode=1, order=1, threads=8)
    assert np.array_equal(bits(got[..., :3]), bits(want[..., :3]))


def test_path_counters_equal_the_oracle(rt, orc, scenes, gpu_ctx):
    w, h, spp = 120, 68, 2
    a, b = both(rt, orc, lambda s: scenes.cover_scene(s, w, h))
    _, _, want = b.render(w, h, 0, spp, 50, rng_mode=1, order=1, threads=8, want_counters=True)
    # walking the reference-order array: every counter, node and primitive tests included
    gpu_ctx.set_reference_traversal(True)
    gpu_ctx.upload_scene(a)
    gpu_ctx.set_target(w, h)
    got = gpu_ctx.count(0, spp, 50)
    assert got.pop("fallbacks") == 0
    assert got == want
    assert got["samples"] == w * h * spp and got["rays"] >= got["samples"]
    assert got["hits"] + got["misses"] == got["rays"]
    # default traversal (search structure first): same paths, far fewer box tests
    gpu_ctx.set_reference_traversal(False)
    gpu_ctx.upload_scene(a)
    fast = gpu_ctx.count(0, spp, 50)
    for k in ("samples", "rays", "hits", "misses", "depth_cutoffs", "scatter_lambertian", "scatter_metal",
              "scatter_dielectric", "scatter_light", "scatter_pbr"):
        assert fast[k] == want[k], k
    assert fast["node_tests"] < 0.7 * want["node_tests"]
    assert fast["fallbacks"] < 1e-3 * fast["rays"]


def test_reference_order_traversal_gives_identical_results(rt, orc, scenes, gpu_ctx):
    """The diagnostic switch that makes every ray walk the reference-order array: same hits,
    same image bits as the default (search structure + guard test + re-trace)."""
    w, h, spp = 200, 112, 2
    a, b = both(rt, orc, lambda s: scenes.cover_scene(s, w, h))
    rays = rays_cover(30000, 11)
    images, hits = [], []
    for on in (True, False):
        gpu_ctx.set_reference_traversal(on)
        gpu_ctx.upload_scene(a)
        hits.append(gpu_ctx.trace(rays))
        images.append(render_gpu(gpu_ctx, w, h, 0, spp))
    gpu_ctx.set_reference_traversal(False)
    for x, y in zip(hits[0], hits[1]):
        assert np.array_equal(bits(x), bits(y)) if x.dtype == np.float32 else np.array_equal(x, y)
    assert np.array_equal(bits(images[0]), bits(images[1]))
    want, _, _ = b.render(w, h, 0, spp, 50, rng_mode=1, order=1, threads=8)
    assert np.array_equal(bits(images[1][..., :3]), bits(want[..., :3]))


def test_two_rank_partition_gathers_to_the_single_rank_image(rt, scenes, tiling, gpu_ctx):
    """Both ranks of a world of 2 rendered on this one GPU, payloads laid end to end as an
    all-gather would leave them, resolved on the device: identical to the 1-rank image."""
    import torch
    w, h, spp = 200, 120, 2
    a = rt.Scene()
    scenes.cover_scene(a, w, h)
    gpu_ctx.upload_scene(a)
    whole = render_gpu(gpu_ctx, w, h, 0, spp)
    for world in (2, 3):
        n4 = tiling.payload_float4s(w, h, world)
        assert rt.accum_bytes(w, h, world) == n4 * 16
        gathered = torch.zeros(world, n4, 4, dtype=torch.float32, device="cuda:0")
        for rank in range(world):
            gpu_ctx.set_target(w, h, rank, world, external_ptr=gathered[rank].data_ptr())
            gpu_ctx.render(0, spp, 50)
            gpu_ctx.sync()
            part = gpu_ctx.read_accum()
            own = tiling.payload_from_image(whole, rank, world)
            assert np.array_equal(bits(tiling.payload_from_image(part, rank, world)), bits(own))
        out = torch.zeros(h, w, 4, dtype=torch.float32, device="cuda:0")
        gpu_ctx.resolve_gathered(gathered.data_ptr(), world, out.data_ptr())
        gpu_ctx.sync()
        assert np.array_equal(bits(out.cpu().numpy()), bits(whole))
        host = tiling.image_from_payloads(gathered.cpu().numpy(), w, h, world)
        assert np.array_equal(bits(host[..., :3]), bits(whole[..., :3]))


def test_full_size_frame_properties(rt, orc, scenes, gpu_ctx):
    """1920x1080 (the benchmark size): oracle parity on sampled rows, split invariance, and a
    second render of the same samples reproducing the same bits."""
    w, h = 1920, 1080
    a, b = both(rt, orc, lambda s: scenes.cover_scene(s, w, h))
    gpu_ctx.upload_scene(a)
    one = render_gpu(gpu_ctx, w, h, 0, 2)
    again = render_gpu(gpu_ctx, w, h, 0, 2)
    assert np.array_equal(bits(one), bits(again))
    gpu_ctx.set_target(w, h)
    gpu_ctx.render(0, 1, 50)
    gpu_ctx.render(1, 1, 50)
    gpu_ctx.sync()
    assert np.array_equal(bits(gpu_ctx.read_accum()), bits(one))
    for row0 in (0, 300, 537, 1072):
        want, _, _ = b.render(w, h, 0, 2, 50, rng_mode=1, order=1, threads=8, row0=row0, row1=row0 + 8)
        assert np.array_equal(bits(one[row0:row0 + 8, :, :3]), bits(want[row0:row0 + 8, :, :3])), row0
    assert np.isfinite(one).all()


def test_converged_image_matches_the_sequential_stream_render(rt, orc, scenes, gpu_ctx):
    """The hash-stream GPU image against the reference-order render on the sequential mt19937
    stream (different random numbers, same estimator): PSNR of the gamma-2 images >= 30 dB at
    256 spp, and the mean colour within 1 %."""
    w, h, spp = 64, 36, 256
    a, b = both(rt, orc, lambda s: scenes.cover_scene(s, w, h))
    gpu_ctx.upload_scene(a)
    got = render_gpu(gpu_ctx, w, h, 0, spp)[..., :3] / spp
    ref, _, _ = b.render(w, h, 0, spp, 50, rng_mode=0, order=0)
    ref = ref[..., :3] / spp
    g, r = np.sqrt(np.clip(got, 0, 1)), np.sqrt(np.clip(ref, 0, 1))
    mse = float(np.mean((g - r) ** 2))
    psnr = 10 * np.log10(1.0 / mse)
    assert psnr >= 30.0, psnr
    assert abs(got.mean() - ref.mean()) / ref.mean() < 0.01


def test_reference_interface_mirror(rt, orc, scenes):
    """RaytracerHIP(image, scene, H, W, spp, maxDepth, buildUp).Run() fills the caller's RGBA8
    buffer; Cancel() from a pass callback stops it."""
    w, h, spp = 80, 45, 4
    a, b = both(rt, orc, lambda s: scenes.cover_scene(s, w, h))
    img = np.zeros(h * w * 4, np.uint8)
    r = rt.RaytracerHIP(img, a, h, w, spp, 50, build_up=False)
    r.Run()
    _, want8, _ = b.render(w, h, 0, spp, 50, rng_mode=1, order=1, threads=8)
    assert np.array_equal(img.reshape(h, w, 4), want8)
    img2 = np.zeros(h * w * 4, np.uint8)
    r2 = rt.RaytracerHIP(img2, a, h, w, 64, 50, build_up=True)
    orig = r2._on_pass

    def stop_after_two(done):
        orig(done)
        if done >= 2:
            r2.Cancel()

    r2._on_pass = stop_after_two
    r2.Run()
    assert r2.cancelled
    _, want2, _ = b.render(w, h, 0, 2, 50, rng_mode=1, order=1, threads=8)
    assert np.array_equal(img2.reshape(h, w, 4), want2)
    r.ctx.close()
    r2.ctx.close()


def test_device_error_paths(rt, scenes, gpu_ctx):
    c = rt.Context(0)
    with pytest.raises(rt.RtError):
        c.render(0, 1)                       # nothing uploaded
    a = rt.Scene()
    scenes.cover_scene(a, 64, 36)
    c.upload_scene(a)
    with pytest.raises(rt.RtError):
        c.render(0, 1)                       # no target
    with pytest.raises(rt.RtError):
        c.set_target(1, 1)                   # W-1 == 0
    with pytest.raises(rt.RtError):
        c.set_target(64, 36, 2, 2)           # rank out of range
    c.set_target(64, 36)
    with pytest.raises(rt.RtError):
        c.render(-1, 1)
    with pytest.raises(rt.RtError):
        c.render(0, 1 << 25)                 # beyond the RNG key's exact range
    c.render(0, 0)                           # an empty sample range is a no-op
    c.sync()
    assert not c.read_accum()[..., :3].any()
    c.close()


def test_cancel_stops_a_long_single_pass(rt, scenes, gpu_ctx):
    """Raytracer::Cancel() from another thread while ONE pass of 2048 spp is running (no pass boundary
    to wait for: the reference polls per scanline, R/src/Core/Raytracer.cpp:54,78).  rt_run returns
    RT_ERR_CANCELLED long before the pass could have finished, every pixel holds the same whole number
    of samples, and they are exactly the first samples of the sequence."""
    import threading
    import time
    w, h, spp = 1280, 720, 2048
    a = rt.Scene()
    scenes.cover_scene(a, w, h)
    gpu_ctx.upload_scene(a)
    gpu_ctx.set_target(w, h)
    gpu_ctx.render(0, 64, 50)           # time 64 spp to know what the whole pass would cost
    gpu_ctx.sync()
    full_estimate = gpu_ctx.last_render_ms() * (spp / 64) / 1e3
    assert full_estimate > 0.25
    seen, result = [], {}

    def run():
        t0 = time.perf_counter()
        try:
            gpu_ctx.run(spp, spp, 50, on_pass=seen.append)
            result["code"] = 0
        except rt.RtError as e:
            result["code"] = e.code
        result["t_end"] = time.perf_counter()
        result["t0"] = t0

    th = threading.Thread(target=run)
    th.start()
    time.sleep(0.06)
    t_cancel = time.perf_counter()
    gpu_ctx.cancel()
    th.join(timeout=30)
    assert not th.is_alive()
    assert result["code"] == rt.RT_ERR_CANCELLED
    latency = result["t_end"] - t_cancel
    assert latency < 0.1, latency                      # the paths in flight finish; nothing new starts
    assert result["t_end"] - result["t0"] < 0.6 * full_estimate
    img = gpu_ctx.read_accum()
    done = seen[-1] if seen else 0
    assert 0 <= done < spp
    assert (img[..., 3] == 1.0).all()
    if done:
        want = render_gpu(gpu_ctx, w, h, 0, done)
        assert np.array_equal(bits(img), bits(want))
    else:
        assert not img[..., :3].any()
    # the flag was cleared on return: the next run completes
    gpu_ctx.set_target(64, 36)
    gpu_ctx.run(2, 1, 50)
    # and a Cancel() that arrives BEFORE Run() stops that run at once
    gpu_ctx.cancel()
    with pytest.raises(rt.RtError) as e:
        gpu_ctx.run(4, 1, 50)
    assert e.value.code == rt.RT_ERR_CANCELLED
    gpu_ctx.sync()
    assert not gpu_ctx.read_accum()[..., :3].any()
    gpu_ctx.run(2, 1, 50)               # cleared again


def test_every_slot_carries_the_sample_count(rt, scenes, tiling, gpu_ctx):
    """A narrow frame shared by many ranks: the first slot of rank 5 of 8 at width 32 lies outside the image
    (its tile column would be 5 of 4), and that slot is where the host reads how many samples a cancelled
    run completed.  Every slot, inside the image or not, holds the count."""
    import torch
    w, h, world, rank = 32, 24, 8, 5
    a = rt.Scene()
    scenes.cover_scene(a, w, h)
    gpu_ctx.upload_scene(a)
    n4 = tiling.payload_float4s(w, h, world)
    buf = torch.zeros(n4, 4, dtype=torch.float32, device="cuda:0")
    gpu_ctx.set_target(w, h, rank, world, external_ptr=buf.data_ptr())
    gpu_ctx.render(0, 3, 50)
    gpu_ctx.render(3, 2, 50)
    gpu_ctx.sync()
    got = buf.cpu().numpy()
    assert (got[:, 3] == 5.0).all()
    assert got[:, :3].any()
    gpu_ctx.release_target()


def test_ingested_tree_renders_like_the_built_one(rt, orc, scenes, gpu_ctx):
    """A scene whose BVH was handed in node by node (rt_scene_begin_bvh_node ...) renders the same bits
    as the scene whose BVH the library built, which in turn equals the oracle."""
    from test_host_flatten import rebuild_through_group_calls
    w, h, spp = 160, 90, 2
    a, o = both(rt, orc, lambda s: scenes.cover_scene(s, w, h))
    b = rt.Scene()
    fa = a.get_flat()
    for k in range(fa["info"].n_materials):
        b.add_material(int(fa["mat_kind"][k]), fa["mat_rgbp"][k][:3], float(fa["mat_rgbp"][k][3]))
    rebuild_through_group_calls(rt, a, b)
    b.set_camera((13.0, 2.0, 3.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), 20.0, scenes.aspect_ratio(w, h), 0.1, 10.0)
    b.set_background(scenes.COVER_BACKGROUND)
    gpu_ctx.upload_scene(b)
    got = render_gpu(gpu_ctx, w, h, 0, spp)
    want, _, _ = o.render(w, h, 0, spp, 50, rng_mode=1, order=1, threads=8)
    assert np.array_equal(bits(got[..., :3]), bits(want[..., :3]))


def test_device_group_of_one_gathers_over_rccl(rt, scenes, gpu_ctx):
    """rt_group_*: the in-library multi-GPU helper with as many devices as this box has (one): contexts,
    the RCCL communicator (ncclCommInitAll), the all-gather on the render stream and the resolve on
    device 0 all run; the frame equals the plain single-context render bit for bit."""
    w, h, spp = 200, 120, 3
    a = rt.Scene()
    scenes.cover_scene(a, w, h)
    gpu_ctx.upload_scene(a)
    want = render_gpu(gpu_ctx, w, h, 0, spp)
    want8 = gpu_ctx.read_rgba8(spp)
    g = rt.Group([0])
    g.upload_scene(a)
    g.set_target(w, h)
    g.render(0, 1, 50)
    g.render(1, spp - 1, 50)          # progressive: consecutive sample ranges
    assert np.array_equal(bits(g.read_accum()), bits(want))
    assert np.array_equal(g.read_rgba8(spp), want8)
    g.clear()
    g.render(0, spp, 50)
    assert np.array_equal(bits(g.read_accum()), bits(want))
    g.set_target(64, 36)              # re-targeting frees and re-creates the per-device buffers
    g.render(0, 1, 50)
    assert g.read_accum()[..., :3].max() > 0
    # a refused size leaves the old target in place ...
    with pytest.raises(rt.RtError) as e:
        g.set_target(1, 1)
    assert e.value.code == -1
    g.render(1, 1, 50)
    assert g.read_accum()[..., :3].max() > 0
    g.close()
    with pytest.raises(rt.RtError):
        rt.Group([99])                # no such device


def test_released_target_is_refused_not_dereferenced(rt, scenes, gpu_ctx):
    """rt_release_target(): render, clear and the read calls report RT_ERR_STATE afterwards (this is what the
    device group does to its contexts before it frees their payloads)."""
    a = rt.Scene()
    scenes.cover_scene(a, 64, 36)
    gpu_ctx.upload_scene(a)
    gpu_ctx.set_target(64, 36)
    gpu_ctx.render(0, 1, 50)
    gpu_ctx.sync()
    gpu_ctx.release_target()
    for call in (lambda: gpu_ctx.render(0, 1, 50), gpu_ctx.clear):
        with pytest.raises(rt.RtError) as e:
            call()
        assert e.value.code == -4
    assert rt.lib.rt_read_rgba8(gpu_ctx.h, 1, None) < 0
    gpu_ctx.set_target(64, 36)
    gpu_ctx.render(0, 1, 50)
    gpu_ctx.sync()
    assert gpu_ctx.read_accum()[..., :3].max() > 0
