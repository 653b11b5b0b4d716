"""Normal mapping as an option (the reference forms the mapped normal and drops it,
R/src/Material/Material.h:123-140; tangent frame as in R/src/Core/Hittable.h:110-125).  Off, nothing
changes; on, the HIP path and the oracle agree bit for bit and the image differs from the unmapped one."""
import os

import numpy as np
import pytest

VASE = os.path.join(os.path.dirname(__file__), "golden", "brass_vase", "brass_vase_04_4k.gltf")


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def tilted_quad_scene(s, tex_normal, tex_diffuse, rough=None):
    """Two textured triangles under a light, seen from above at an angle."""
    d = s.add_texture(tex_diffuse)
    n = s.add_texture(tex_normal)
    r = s.add_texture(rough) if rough is not None else -1
    pbr = s.add_pbr(d, r, n)
    light = s.add_material(3, (6.0, 6.0, 6.0), 0.0)
    pos = np.float32([[-2, 0, -2], [2, 0, -2], [2, 0.5, 2], [-2, 0.5, 2]])
    uv = np.float32([[0.05, 0.05], [0.95, 0.05], [0.95, 0.95], [0.05, 0.95]])
    zero = np.zeros(9, np.float32)
    for tri in ((0, 1, 2), (0, 2, 3)):
        s.add_triangle(pos[list(tri)].reshape(-1), pbr, zero, uv[list(tri)].reshape(-1))
    s.add_sphere((0.0, 6.0, 0.0), 2.0, light)
    s.set_camera((0.0, 4.0, -6.0), (0.0, 0.2, 0.0), (0.0, 1.0, 0.0), 40.0, 1.0, 0.0, 10.0)
    s.set_background((0.2, 0.3, 0.5))


def maps(seed=3):
    rng = np.random.default_rng(seed)
    normal = np.empty((64, 48, 3), np.uint8)  # non-square on purpose
    normal[..., 0] = rng.integers(64, 192, (64, 48))
    normal[..., 1] = rng.integers(64, 192, (64, 48))
    normal[..., 2] = rng.integers(200, 256, (64, 48))
    diffuse = rng.integers(60, 250, (32, 32, 3)).astype(np.uint8)
    rough = rng.integers(0, 120, (16, 16, 3)).astype(np.uint8)
    return normal, diffuse, rough


def test_oracle_switch_changes_only_normal_mapped_materials(orc):
    normal, diffuse, rough = maps()
    imgs = []
    for on in (False, True):
        s = orc.OracleScene()
        tilted_quad_scene(s, normal, diffuse, rough)
        s.set_normal_mapping(on)
        imgs.append(s.render(48, 48, 0, 4, 8, rng_mode=1, order=1, threads=4)[0])
    assert not np.array_equal(bits(imgs[0]), bits(imgs[1]))
    # a flat normal map (0.5, 0.5, 1 -> the shading normal itself, up to rounding of the byte values) keeps the image close
    flat = np.zeros_like(normal)
    flat[..., 0] = flat[..., 1] = 128
    flat[..., 2] = 255
    s = orc.OracleScene()
    tilted_quad_scene(s, flat, diffuse, None)
    s.set_normal_mapping(True)
    a = s.render(48, 48, 0, 16, 8, rng_mode=1, order=1, threads=4)[0]
    s = orc.OracleScene()
    tilted_quad_scene(s, flat, diffuse, None)
    b = s.render(48, 48, 0, 16, 8, rng_mode=1, order=1, threads=4)[0]
    assert abs(float(a[..., :3].mean()) - float(b[..., :3].mean())) < 0.02 * float(b[..., :3].mean())


@pytest.mark.gpu
@pytest.mark.parametrize("with_roughness", [False, True])
def test_normal_mapped_quad_equals_the_oracle(rt, orc, gpu_ctx, with_roughness):
    normal, diffuse, rough = maps()
    w = h = 96
    out = {}
    for on in (False, True):
        a, b = rt.Scene(), orc.OracleScene()
        for s in (a, b):
            tilted_quad_scene(s, normal, diffuse, rough if with_roughness else None)
            s.set_normal_mapping(on)
        gpu_ctx.upload_scene(a)
        gpu_ctx.set_target(w, h)
        gpu_ctx.render(0, 8, 12)
        gpu_ctx.sync()
        got = gpu_ctx.read_accum()
        want = b.render(w, h, 0, 8, 12, rng_mode=1, order=1, threads=8)[0]
        assert np.array_equal(bits(got[..., :3]), bits(want[..., :3]))
        out[on] = got
    assert not np.array_equal(bits(out[False]), bits(out[True]))


@pytest.mark.gpu
def test_normal_mapped_vase_equals_the_oracle(rt, orc, scenes, gpu_ctx):
    """Cornell box + brass vase with its own normal map (4096x4096, decoded by the library) taking effect."""
    w, h, spp = 256, 144, 2
    mesh = scenes.load_gltf_mesh(VASE)
    tex = scenes.load_gltf_textures(VASE)
    a, b = rt.Scene(), orc.OracleScene()
    for s in (a, b):
        scenes.cornell_mesh_scene(s, w, h, mesh, tex)
        s.set_normal_mapping(True)
    gpu_ctx.upload_scene(a)
    gpu_ctx.set_target(w, h)
    gpu_ctx.render(0, spp, 50)
    gpu_ctx.sync()
    got = gpu_ctx.read_accum()
    want = b.render(w, h, 0, spp, 50, rng_mode=1, order=1, threads=16)[0]
    assert np.array_equal(bits(got[..., :3]), bits(want[..., :3]))
    a.set_normal_mapping(False)
    gpu_ctx.upload_scene(a)
    gpu_ctx.clear()
    gpu_ctx.render(0, spp, 50)
    gpu_ctx.sync()
    assert not np.array_equal(bits(gpu_ctx.read_accum()), bits(got))
