"""The large configurations of the workload list and edge cases of the texture / dielectric
arithmetic, on the GPU against the oracle.  Bar: bit-exact."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
G = os.path.join(HERE, "golden")
VASE = os.path.join(G, "brass_vase", "brass_vase_04_4k.gltf")


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def test_million_triangle_instanced_scene(rt, orc, scenes, gpu_ctx):
    """The large-mesh configuration at a million triangles: 363 baked instances of the brass vase in
    one Mesh with an unlimited-depth reference BVH, at 1920x1080.  The scene is far too large for LDS,
    so the kernels read every table through the caches and walk the direction-ordered copies of the
    search structure.  Oracle rows bit-exact, both walks identical, counters sane."""
    mesh = scenes.load_gltf_mesh(VASE)
    tex = {"diffuse": scenes.noise_texture(256, 1), "roughness": scenes.noise_texture(256, 2)}
    w, h = 1920, 1080
    a, b = rt.Scene(), orc.OracleScene()
    n = scenes.instanced_mesh_scene(a, mesh, 363, tex)
    assert n == 363 * 2760 and n > 1_000_000
    info = a.flatten()
    assert info.n_triangles == n + 12 and not info.fits_lds
    assert info.n_nodes > 1_000_000                      # unlimited depth: about one entry per triangle
    stride, _ = a.search_structure_octants()
    assert stride > 0                                    # eight direction-ordered copies are in use
    gpu_ctx.upload_scene(a)
    gpu_ctx.set_target(w, h)
    gpu_ctx.render(0, 1, 50)
    gpu_ctx.sync()
    got = gpu_ctx.read_accum()
    assert np.isfinite(got).all() and got[..., :3].max() > 0
    scenes.instanced_mesh_scene(b, mesh, 363, tex)
    for row0 in (120, 540, 800):
        want, _, _ = b.render(w, h, 0, 1, 50, rng_mode=1, order=1, threads=16, row0=row0, row1=row0 + 2)
        assert np.array_equal(bits(got[row0:row0 + 2, :, :3]), bits(want[row0:row0 + 2, :, :3])), row0
    cnt = gpu_ctx.count(0, 1, 50)
    assert cnt["samples"] == w * h and cnt["hits"] + cnt["misses"] == cnt["rays"]
    assert 2.0 < cnt["rays"] / cnt["samples"] < 20.0
    assert cnt["scatter_pbr"] > 0.3 * cnt["hits"]        # most hits are on the vases
    assert cnt["fallbacks"] < 1e-3 * cnt["rays"]
    assert cnt["node_tests"] / cnt["rays"] < 80 and cnt["tri_tests"] / cnt["rays"] < 20
    # the reference-order walk over the same million-entry array returns the same image
    gpu_ctx.set_reference_traversal(True)
    try:
        gpu_ctx.upload_scene(a)
        gpu_ctx.set_target(w, h)
        gpu_ctx.render(0, 1, 50)
        gpu_ctx.sync()
        assert np.array_equal(bits(gpu_ctx.read_accum()), bits(got))
    finally:
        gpu_ctx.set_reference_traversal(False)


def test_ten_million_triangle_scene_at_its_size(rt, orc, scenes, gpu_ctx):
    """The large-mesh configuration at its full size: 3,623 baked instances of the brass vase = 9,999,480
    triangles (+ the Cornell box) under one unlimited-depth reference BVH, at 1920x1080.  The search structure
    has more than 2 M entries, so it is held ONCE (no direction-ordered copies) and the loop walks it with the
    general box test -- the only configuration that takes that path.  Three oracle row pairs bit-exact, the
    reference-order walk and the device-built structure give the same image."""
    mesh = scenes.load_gltf_mesh(VASE)
    tex = {"diffuse": scenes.noise_texture(256, 1), "roughness": scenes.noise_texture(256, 2)}
    w, h, copies = 1920, 1080, 3623
    a = rt.Scene()
    n = scenes.instanced_mesh_scene(a, mesh, copies, tex)
    assert n == copies * 2760 == 9_999_480
    info = a.flatten()
    assert info.n_triangles == n + 12 and not info.fits_lds and info.n_nodes > 9_000_000
    stride, entries = a.search_structure_octants()
    assert stride == 0                                   # single copy: beyond the 2 M-entry limit of the copies
    del entries
    gpu_ctx.upload_scene(a)                              # the scene carries its host-built structure: that is walked
    assert gpu_ctx.last_build_ms() == 0
    gpu_ctx.set_target(w, h)
    gpu_ctx.render(0, 1, 50)
    gpu_ctx.sync()
    got = gpu_ctx.read_accum()
    assert np.isfinite(got).all() and got[..., :3].max() > 0
    b = orc.OracleScene()
    scenes.instanced_mesh_scene(b, mesh, copies, tex)
    for row0 in (200, 540, 900):
        want, _, _ = b.render(w, h, 0, 1, 50, rng_mode=1, order=1, threads=16, row0=row0, row1=row0 + 2)
        assert np.array_equal(bits(got[row0:row0 + 2, :, :3]), bits(want[row0:row0 + 2, :, :3])), row0
    del b
    cnt = gpu_ctx.count(0, 1, 50)
    assert cnt["samples"] == w * h and cnt["hits"] + cnt["misses"] == cnt["rays"]
    assert cnt["fallbacks"] < 1e-3 * cnt["rays"] and cnt["tri_tests"] > 0 and cnt["sphere_tests"] == 0
    # the reference-order walk over the ten-million-entry array returns the same image
    gpu_ctx.set_reference_traversal(True)
    try:
        gpu_ctx.upload_scene(a)
        gpu_ctx.set_target(w, h)
        gpu_ctx.render(0, 1, 50)
        gpu_ctx.sync()
        assert np.array_equal(bits(gpu_ctx.read_accum()), bits(got))
    finally:
        gpu_ctx.set_reference_traversal(False)
    # and so does the structure built on the device, which is what a context left to itself does for a scene of
    # this size that has not been flattened on the host yet
    dev = rt.Context(0)
    try:
        a2 = rt.Scene()
        scenes.instanced_mesh_scene(a2, mesh, copies, tex)
        dev.upload_scene(a2)
        assert 0 < dev.last_build_ms() < 2000
        dev.set_target(w, h)
        dev.render(0, 1, 50)
        dev.sync()
        assert np.array_equal(bits(dev.read_accum()), bits(got))
    finally:
        dev.close()


def test_4k_frame_one_of_eight_ranks(rt, orc, scenes, tiling, gpu_ctx):
    """3840x2160 split over a world of 8: the payload a rank renders holds exactly its tiles of the
    single-rank frame, bit for bit, and the frame itself matches the oracle on sampled rows."""
    w, h, spp = 3840, 2160, 1
    a, b = rt.Scene(), orc.OracleScene()
    scenes.cover_scene(a, w, h)
    scenes.cover_scene(b, w, h)
    gpu_ctx.upload_scene(a)
    gpu_ctx.set_target(w, h)
    gpu_ctx.render(0, spp, 50)
    gpu_ctx.sync()
    whole = gpu_ctx.read_accum()
    for row0 in (0, 1077, 2158):
        want, _, _ = b.render(w, h, 0, spp, 50, rng_mode=1, order=1, threads=16, row0=row0, row1=row0 + 2)
        assert np.array_equal(bits(whole[row0:row0 + 2, :, :3]), bits(want[row0:row0 + 2, :, :3])), row0
    world = 8
    assert rt.accum_bytes(w, h, world) == tiling.payload_float4s(w, h, world) * 16
    for rank in (0, 5, 7):
        gpu_ctx.set_target(w, h, rank, world)
        gpu_ctx.render(0, spp, 50)
        gpu_ctx.sync()
        part = gpu_ctx.read_accum()
        own = tiling.payload_from_image(whole, rank, world)
        assert np.array_equal(bits(tiling.payload_from_image(part, rank, world)), bits(own)), rank
        # and nothing outside the rank's tiles was written
        other = tiling.payload_from_image(part, (rank + 1) % world, world)
        assert not other[..., :3].any()


def test_non_square_texture_and_out_of_range_uv(rt, orc, gpu_ctx):
    """Texture::At uses the HEIGHT as the row stride (R/src/Material/Texture.h:42), which only
    coincides with the width for square maps; and it does not clamp.  Wide and tall maps, (u, v) in
    and far outside [0, 1], exact 0 and 1: the kernels' index arithmetic equals the oracle's
    (which clamps the final index, where the reference would read out of bounds)."""
    rs = np.random.RandomState(5)
    for tw, th in ((48, 32), (32, 48), (7, 1), (1, 9), (64, 64)):
        tex_d = rs.randint(0, 256, (th, tw, 3)).astype(np.uint8)
        tex_r = rs.randint(0, 256, (th, tw, 3)).astype(np.uint8)

        def build(s):
            td, tr = s.add_texture(tex_d), s.add_texture(tex_r)
            s.add_pbr(td, tr, -1)
            s.add_pbr(td, -1, -1)
            s.add_sphere((0, 0, 0), 1.0, 0)

        a, b = rt.Scene(), orc.OracleScene()
        build(a)
        build(b)
        gpu_ctx.upload_scene(a)
        n = 3000
        uv = (rs.rand(n, 2) * 1.2 - 0.1).astype(np.float32)
        uv[:400] = (rs.rand(400, 2) * 8 - 4).astype(np.float32)           # far outside
        uv[400:420] = np.float32([[0, 0], [1, 1], [0, 1], [1, 0], [0.5, 0], [0, 0.5], [1, 0.5], [0.5, 1],
                                  [-0.0, 1.0], [1e-9, 1 - 1e-9]] * 2)
        uv[420:440, 0] = (np.arange(20) + 0.5) / tw                       # texel boundaries: (x + 0.5) / W
        uv[440:460, 1] = 1.0 - (np.arange(20) + 0.5) / th
        mat = rs.randint(0, 2, n).astype(np.int32)
        nrm = np.tile(np.float32([0, 1, 0]), (n, 1))
        din = rs.randn(n, 3).astype(np.float32)
        din[:, 1] = -np.abs(din[:, 1]) - np.float32(0.1)
        p = np.tile(np.float32([0, 1, 0]), (n, 1))
        ray = np.concatenate([p - din, din], 1).astype(np.float32)
        key = np.stack([rs.randint(0, 1920, n), rs.randint(0, 1080, n), rs.randint(0, 64, n), rs.randint(0, 40, n)], 1).astype(np.uint32)
        att, out, alive, cnt = gpu_ctx.scatter_test(mat, ray, p, nrm, np.ones(n, np.int32), uv, key)
        assert np.array_equal(bits(att), bits(b.texture_at(0, uv))), (tw, th)
        for i in range(0, n, 7):
            oa, oo, oal, oc = b.scatter(int(mat[i]), ray[i], p[i], nrm[i], 1, uv[i, 0], uv[i, 1], key[i, 0], key[i, 1],
                                        key[i, 2], key[i, 3])
            assert alive[i] == oal and cnt[i] == oc, (tw, th, i)
            assert np.array_equal(bits(att[i]), bits(oa)) and np.array_equal(bits(out[i]), bits(oo)), (tw, th, i)


def test_schlick_reflectance_sweep(rt, orc, gpu_ctx):
    """Dielectric::reflectance (R/src/Material/Material.h:90-95) calls pow(1 - cosine, 5) in double;
    the kernels form the fifth power with compensated products.  Four million cosines -- a dense
    sweep of [0, 1], random floats, values next to 0 and 1, the negative cosines fmin() lets
    through -- times both refraction ratios of glass: every float result equals the oracle's."""
    rs = np.random.RandomState(9)
    dense = np.linspace(0.0, 1.0, 1_000_001, dtype=np.float64).astype(np.float32)
    rnd = rs.rand(900_000).astype(np.float32)
    tiny = np.float32(1.0) - (np.float32(2.0) ** -np.arange(1, 25)).astype(np.float32)
    near0 = (np.float32(2.0) ** -np.arange(1, 60).astype(np.float32)).astype(np.float32)
    neg = -rs.rand(99_000).astype(np.float32)
    cos = np.concatenate([dense, rnd, tiny, near0, -near0, neg, np.float32([0, 1, -1, 0.5])])
    total = 0
    for ratio in (np.float32(1.5), np.float32(1.0) / np.float32(1.5), np.float32(1.33), np.float32(2.4)):
        r = np.full(cos.size, ratio, np.float32)
        got = gpu_ctx.reflectance_test(cos, r)
        want = orc.reflectance(cos, r)
        diff = np.flatnonzero(bits(got) != bits(want))
        assert diff.size == 0, (float(ratio), diff[:5], cos[diff[:5]], got[diff[:5]], want[diff[:5]])
        total += cos.size
    assert total >= 4_000_000


def test_far_camera_rays_take_the_reference_walk(rt, orc, gpu_ctx):
    """Ray origins far from small primitives: the rounding error of the search structure's fused
    box test grows with the origin's coordinates, so beyond the distance its padding covers a ray
    walks the reference-order array instead.  Unit spheres and thin axis-aligned triangles near the
    coordinate origin seen from 1e3 .. 1e6 units away: images equal the oracle's bit for bit."""
    rs = np.random.RandomState(21)

    def make(dist, seed):
        r = np.random.RandomState(seed)
        sph = [((r.rand(3) * 6 - 3).astype(np.float32), float(np.float32(0.3 + r.rand()))) for _ in range(30)]
        tris = []
        for _ in range(40):
            c = (r.rand(3) * 6 - 3).astype(np.float32)
            ax = r.randint(3)
            e1 = np.zeros(3, np.float32)
            e2 = np.zeros(3, np.float32)
            e1[(ax + 1) % 3] = 0.2 + r.rand()
            e2[(ax + 2) % 3] = 0.2 + r.rand()
            tris.append(np.concatenate([c, c + e1, c + e2]).astype(np.float32))   # flat along one axis
        d = r.randn(3)
        d /= np.linalg.norm(d)
        lookfrom = (d * dist).astype(np.float32)

        def build(s):
            m = [s.add_material(0, (0.6, 0.5, 0.4)), s.add_material(1, (0.9, 0.9, 0.8), 0.1), s.add_material(2, (0, 0, 0), 1.5)]
            for i, (c, rad) in enumerate(sph):
                s.add_sphere(c, rad, m[i % 3])
            for i, t in enumerate(tris):
                s.add_triangle(t, m[i % 2])
            s.wrap_bvh()
            # a narrow field of view so that the cluster fills the frame from any distance
            s.set_camera(lookfrom, (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), float(np.degrees(2 * np.arctan(5.0 / dist))), 1.0, 0.0,
                         float(dist))
            s.set_background((0.7, 0.8, 1.0))
        return build

    seen_far = 0
    for k, dist in enumerate((3e2, 1e3, 1e4, 1e5, 1e6, 3e3, 3e4)):
        build = make(dist, 100 + k)
        a, b = rt.Scene(), orc.OracleScene()
        build(a)
        build(b)
        gpu_ctx.upload_scene(a)
        gpu_ctx.set_target(96, 64)
        gpu_ctx.render(0, 2, 50)
        gpu_ctx.sync()
        got = gpu_ctx.read_accum()
        want, _, _ = b.render(96, 64, 0, 2, 50, rng_mode=1, order=1, threads=8)
        assert np.array_equal(bits(got[..., :3]), bits(want[..., :3])), dist
        rays = np.zeros((2000, 6), np.float32)
        d = rs.randn(3)
        rays[:, :3] = (d / np.linalg.norm(d) * dist).astype(np.float32)
        tgt = (rs.rand(2000, 3) * 6 - 3).astype(np.float32)
        rays[:, 3:] = tgt - rays[:, :3]
        k1, p1, t1, n1, _, _ = gpu_ctx.trace(rays)
        k2, p2, t2, n2, _, _ = b.trace(rays)
        assert np.array_equal(k1, k2) and np.array_equal(p1, p2) and np.array_equal(bits(t1), bits(t2)), dist
        seen_far += int((k1 >= 0).sum())
    assert seen_far > 1000


def test_cornell_vase_with_its_real_maps(rt, orc, scenes, gpu_ctx):
    """The Cornell box with the brass vase as the application sets it up (R/src/RaytracingApplication.cpp:
    296-346), textured with the vase's own three 4096x4096 JPEG maps decoded by the library (bytes equal
    stb_image's, tests/test_image_decode.py): image equal to the oracle's bit for bit, and the one-call
    loader (rt_scene_add_mesh_file) renders the same image."""
    w, h, spp = 320, 180, 2
    mesh = scenes.load_gltf_mesh(VASE)
    tex = scenes.load_gltf_textures(VASE)
    assert all(tex[k].shape == (4096, 4096, 3) for k in ("diffuse", "roughness", "normal"))
    a, b = rt.Scene(), orc.OracleScene()
    scenes.cornell_mesh_scene(a, w, h, mesh, tex)
    scenes.cornell_mesh_scene(b, w, h, mesh, tex)
    gpu_ctx.upload_scene(a)
    gpu_ctx.set_target(w, h)
    gpu_ctx.render(0, spp, 50)
    gpu_ctx.sync()
    got = gpu_ctx.read_accum()
    want, _, cnt = b.render(w, h, 0, spp, 50, rng_mode=1, order=1, threads=16, want_counters=True)
    assert np.array_equal(bits(got[..., :3]), bits(want[..., :3]))
    assert cnt["scatter_pbr"] > 1000 and got[..., :3].max() > 0
    c = rt.Scene()
    c.cornell_box()
    c.add_mesh_file(scenes.VASE_MODEL, VASE)
    c.set_camera((278.0, 278.0, -800.0), (278.0, 278.0, 0.0), (0.0, 1.0, 0.0), 40.0, scenes.aspect_ratio(w, h), 0.0, 10.0)
    c.set_background((0.0, 0.0, 0.0))
    gpu_ctx.upload_scene(c)
    gpu_ctx.set_target(w, h)
    gpu_ctx.render(0, spp, 50)
    gpu_ctx.sync()
    assert np.array_equal(bits(gpu_ctx.read_accum()), bits(got))
