"""The texture path against the reference's own decoder.

tests/golden/stb_*.{npy,npz,json} were written by tests/golden/make_stb_fixtures.py from
oracle/_ref/libstb_ref.so, the reference's vendored stb_image v2.27 compiled unmodified: the
float bit patterns stbi_loadf produces (R/vendor/stb_image/stb_image.h:1840-1861), and sampled
texels of the three brass-vase maps as Texture's constructor receives them
(R/src/Material/Texture.h:7-29).  The oracle and the product must reproduce them bit for bit.
"""
import ctypes as C
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
G = os.path.join(HERE, "golden")
ROOT = os.path.dirname(HERE)


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def test_lut_of_oracle_and_product_equal_stbi_loadf(rt, orc):
    want = np.load(os.path.join(G, "stb_lut_bits.npy"))
    assert want.shape == (256,) and want[0] == 0 and want[255] == 0x3F800000
    assert np.array_equal(bits(orc.texture_lut()), want)
    assert np.array_equal(bits(rt.texture_lut()), want)


def test_float_texels_are_the_lut_of_the_byte_texels():
    """stbi_loadf(x) == lut[stbi_load(x)] on the vase maps: the RGB8 + LUT representation the
    library keeps in HBM loses nothing against the reference's float arrays."""
    lut = np.load(os.path.join(G, "stb_lut_bits.npy"))
    z = np.load(os.path.join(G, "stb_vase_texels.npz"))
    assert z["rgb8"].shape == (3, 4096, 3)
    assert np.array_equal(lut[z["rgb8"]], z["f32bits"])
    meta = json.load(open(os.path.join(G, "stb_vase_maps.json")))
    for m in meta.values():
        assert (m["width"], m["height"], m["channels"]) == (4096, 4096, 3)


def test_ref_library_reproduces_its_fixture_when_present():
    """Where oracle/_ref exists (it is built wherever the reference tree is), the committed LUT is
    what that binary returns today."""
    path = os.path.join(ROOT, "oracle", "_ref", "libstb_ref.so")
    if not os.path.exists(path):
        pytest.skip("oracle/_ref not built here")
    from golden.make_stb_fixtures import lut_bits, stb
    assert np.array_equal(lut_bits(stb()), np.load(os.path.join(G, "stb_lut_bits.npy")))


def texel_scene(scene, rgb8_tex):
    td = scene.add_texture(rgb8_tex)
    scene.add_pbr(td, -1, -1)
    scene.add_sphere((0, 0, 0), 1.0, 0)


@pytest.mark.gpu
def test_device_texture_fetch_equals_stbi_loadf(rt, orc, gpu_ctx):
    """The kernels' Texture::At (RGBX8 pool + LUT in HBM) returns, for texels taken from the real
    vase maps, exactly the floats stbi_loadf gave the reference -- through the scatter hook, whose
    attenuation for a PBR material is diffuse.At(u, v) (R/src/Material/Material.h:142)."""
    z = np.load(os.path.join(G, "stb_vase_texels.npz"))
    n = 64
    for k in range(3):
        tex = z["rgb8"][k].reshape(n, n, 3)
        a, b = rt.Scene(), orc.OracleScene()
        texel_scene(a, tex)
        texel_scene(b, tex)
        gpu_ctx.upload_scene(a)
        # texel (x, y) <- u = (x + 0.75) / n, v = 1 - (y + 0.75) / n  (int((u * n) - 0.5) == x)
        yy, xx = np.mgrid[0:n, 0:n]
        uv = np.stack([(xx + 0.75) / n, 1.0 - (yy + 0.75) / n], -1).reshape(-1, 2).astype(np.float32)
        m = n * n
        nrm = np.tile(np.float32([0, 1, 0]), (m, 1))
        ray = np.tile(np.float32([0, 2, 0, 0.1, -1, 0.2]), (m, 1))
        p = np.tile(np.float32([0, 1, 0]), (m, 1))
        key = np.stack([np.arange(m) % 640, np.arange(m) // 640, np.zeros(m), np.zeros(m)], 1).astype(np.uint32)
        att, _, _, _ = gpu_ctx.scatter_test(np.zeros(m, np.int32), ray, p, nrm, np.ones(m, np.int32), uv, key)
        assert np.array_equal(bits(att), z["f32bits"][k]), k
        assert np.array_equal(bits(b.texture_at(0, uv)), z["f32bits"][k]), k
