"""The search structure built on the device (rt_set_device_build): its invariants, read back from
the GPU, and the images rendered with it -- which must not differ by a bit from the host-built
structure's, because the guard test against the reference's boxes decides every hit."""
import os

import numpy as np
import pytest

from test_search_structure import check_search_structure

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
VASE = os.path.join(HERE, "golden", "brass_vase", "brass_vase_04_4k.gltf")


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


@pytest.fixture()
def dev_ctx(rt):
    ctx = rt.Context(0)
    ctx.set_device_build(True)
    yield ctx
    ctx.close()


def render(ctx, scene, w, h, spp):
    ctx.upload_scene(scene)
    ctx.set_target(w, h)
    ctx.render(0, spp, 50)
    ctx.sync()
    return ctx.read_accum()


def test_device_built_structure_of_the_cover_scene(rt, orc, scenes, gpu_ctx, dev_ctx):
    w, h, spp = 200, 112, 2
    a, b = rt.Scene(), orc.OracleScene()
    scenes.cover_scene(a, w, h)
    scenes.cover_scene(b, w, h)
    got = render(dev_ctx, a, w, h, spp)
    entries, refs = dev_ctx.read_search_structure()
    assert len(refs) == 488 and 488 // 2 <= len(entries) < 2 * 488
    a2 = rt.Scene()
    scenes.cover_scene(a2, w, h)
    check_search_structure(a2, (entries, refs))
    want, _, _ = b.render(w, h, 0, spp, 50, rng_mode=1, order=1, threads=8)
    assert np.array_equal(bits(got[..., :3]), bits(want[..., :3]))
    assert np.array_equal(bits(got), bits(render(gpu_ctx, a2, w, h, spp)))
    cnt = dev_ctx.count(0, spp, 50)
    assert cnt["fallbacks"] < 1e-3 * cnt["rays"]
    assert dev_ctx.last_build_ms() > 0


@pytest.mark.parametrize("top", ["default", "1"])
def test_device_built_structure_edge_cases(rt, orc, dev_ctx, monkeypatch, top):
    """One primitive, two, coincident primitives (equal Morton keys are split by position), hundreds of coincident
    ones (the clusters the host splits the top of the tree over all have the same box), a mix of spheres and
    triangles with a huge sphere among small ones -- with the top of the tree split on the host (the default) and
    with clustering all the way up (RT_DEVICE_TOP=1)."""
    if top != "default":
        monkeypatch.setenv("RT_DEVICE_TOP", top)
    rs = np.random.RandomState(2)
    for case in ("one", "two", "coincident", "many coincident", "mixed"):
        def build(s):
            m = [s.add_material(0, (0.6, 0.5, 0.4)), s.add_material(1, (0.9, 0.9, 0.9), 0.1), s.add_material(2, (0, 0, 0), 1.5)]
            r = np.random.RandomState(7)
            if case == "one":
                s.add_sphere((0.0, 0.0, 0.0), 1.0, m[0])
            elif case == "two":
                s.add_sphere((0.0, 0.0, 0.0), 1.0, m[0])
                s.add_triangle(np.float32([-2, -1, 1, 2, -1, 1, 0, 2, 1]), m[1])
            elif case == "coincident":
                for i in range(9):
                    s.add_sphere((0.5, 0.0, 0.0), 1.0, m[i % 3])
                for i in range(5):
                    s.add_triangle(np.float32([-2, -1, 1.5, 2, -1, 1.5, 0, 2, 1.5]), m[i % 2])
            elif case == "many coincident":
                for i in range(300):
                    s.add_sphere((0.5, 0.0, 0.0), 1.0, m[i % 3])
                for i in range(260):
                    s.add_triangle(np.float32([-2, -1, 1.5, 2, -1, 1.5, 0, 2, 1.5]), m[i % 2])
                s.wrap_bvh()
            else:
                s.add_sphere((0.0, -1000.0, 0.0), 999.0, m[0])
                for i in range(300):
                    s.add_sphere((r.rand(3) * 6 - 3).astype(np.float32), float(np.float32(0.05 + 0.2 * r.rand())), m[i % 3])
                for i in range(200):
                    c = (r.rand(3) * 6 - 3).astype(np.float32)
                    s.add_triangle(np.concatenate([c, c + r.rand(3) * 0.5, c + r.rand(3) * 0.5]).astype(np.float32), m[i % 2])
                s.wrap_bvh()
            s.set_camera((0.0, 1.0, 9.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), 50.0, 1.0, 0.05, 9.0)
            s.set_background((0.7, 0.8, 1.0))
        a, b = rt.Scene(), orc.OracleScene()
        build(a)
        build(b)
        got = render(dev_ctx, a, 96, 64, 2)
        a2 = rt.Scene()
        build(a2)
        check_search_structure(a2, dev_ctx.read_search_structure())
        want, _, _ = b.render(96, 64, 0, 2, 50, rng_mode=1, order=1, threads=8)
        assert np.array_equal(bits(got[..., :3]), bits(want[..., :3])), case
        rays = np.concatenate([np.tile(np.float32([0, 1, 9]), (4000, 1)), (rs.rand(4000, 3) * 8 - 4 - np.float32([0, 1, 9])).astype(np.float32)], 1)
        k1, p1, t1, _, _, _ = dev_ctx.trace(rays)
        k2, p2, t2, _, _, _ = b.trace(rays)
        assert np.array_equal(k1, k2) and np.array_equal(p1, p2) and np.array_equal(bits(t1), bits(t2)), case


def test_device_build_of_a_million_triangles(rt, orc, scenes, gpu_ctx, dev_ctx):
    """The million-triangle instanced scene: the device build takes milliseconds where the host build
    takes seconds, the structure holds every triangle under enclosing boxes, and the image equals the
    host-structure image and the oracle's rows."""
    import time
    mesh = scenes.load_gltf_mesh(VASE)
    tex = {"diffuse": scenes.noise_texture(256, 1), "roughness": scenes.noise_texture(256, 2)}
    w, h = 1280, 720
    a = rt.Scene()
    n = scenes.instanced_mesh_scene(a, mesh, 363, tex)
    t0 = time.perf_counter()
    got = render(dev_ctx, a, w, h, 1)
    t_dev = time.perf_counter() - t0
    build_ms = dev_ctx.last_build_ms()
    assert 0 < build_ms < 200, build_ms
    entries, refs = dev_ctx.read_search_structure()
    assert len(refs) == n + 12 and len(entries) < 2 * len(refs)
    a2 = rt.Scene()
    scenes.instanced_mesh_scene(a2, mesh, 363, tex)
    t0 = time.perf_counter()
    want = render(gpu_ctx, a2, w, h, 1)
    t_host = time.perf_counter() - t0
    skip, first = entries[:, 6].astype(np.int64), entries[:, 7]
    assert (skip > np.arange(len(entries))).all() and skip.max() == len(entries)
    listed = np.sort(refs & np.uint32(0x7FFFFFFF))
    assert np.array_equal(listed, np.sort(a2.get_flat()["prim_refs"] & np.uint32(0x7FFFFFFF)))
    assert np.array_equal(bits(got), bits(want))
    b = orc.OracleScene()
    scenes.instanced_mesh_scene(b, mesh, 363, tex)
    ref, _, _ = b.render(w, h, 0, 1, 50, rng_mode=1, order=1, threads=16, row0=300, row1=302)
    assert np.array_equal(bits(got[300:302, :, :3]), bits(ref[300:302, :, :3]))
    cnt = dev_ctx.count(0, 1, 50)
    assert cnt["fallbacks"] < 1e-3 * cnt["rays"]
    print(f"device build {build_ms:.1f} ms; flatten + upload + render: device-built {t_dev:.2f} s, host-built {t_host:.2f} s; "
          f"node tests per ray {cnt['node_tests'] / cnt['rays']:.1f}")
