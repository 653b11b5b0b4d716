"""The library's own JPEG / PNG decoder against the reference's vendored stb_image v2.27.

tests/golden/images/ holds small synthetic files (written with Pillow by
tests/golden/make_stb_fixtures.py) and expected.npz, the bytes stbi_load() of oracle/_ref returned
for each.  Bar: identical bytes.  Where the reference tree is present (the build container) the
three 4096x4096 brass-vase maps are decoded too and compared with their committed SHA-256 and with
the sampled texels in stb_vase_texels.npz."""
import hashlib
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
IMG = os.path.join(HERE, "golden", "images")
REF_TEX = "/root/reference/Raytracing-In-A-Weekend/assets/models/brass_vase/textures"

EXPECTED = np.load(os.path.join(IMG, "expected.npz"))
NAMES = sorted(k for k in EXPECTED.files if not k.endswith(":channels"))


@pytest.mark.parametrize("name", NAMES)
def test_decoder_equals_stb_image(rt, name):
    blob = open(os.path.join(IMG, name), "rb").read()
    want = EXPECTED[name]
    if "progressive" in name:
        with pytest.raises(rt.RtError) as e:      # refused loudly, never mis-decoded
            rt.image_decode(blob)
        assert "progressive" in str(e.value)
        return
    got = rt.image_decode(blob)
    assert got.shape == want.shape, (got.shape, want.shape)
    assert got.shape[2] == int(EXPECTED[name + ":channels"])
    assert np.array_equal(got, want), f"{name}: {(got != want).sum()} bytes differ, max |d| {np.abs(got.astype(int) - want).max()}"


def test_every_decoder_path_has_a_fixture():
    kinds = " ".join(NAMES)
    for needle in ("rgb420", "rgb422", "rgb444", "grey_", "restart", "odd", "optimised", "rgba8", "palette", "bilevel",
                   "rgb16", "greyalpha", "uncompressed"):
        assert needle in kinds, needle


def test_malformed_files_are_refused(rt):
    for name in ("rgb420_64x48.jpg", "rgb8_31x17.png", "palette_40x30.png"):
        blob = open(os.path.join(IMG, name), "rb").read()
        for cut in (0, 1, 7, 20, len(blob) // 3, len(blob) // 2):
            try:
                rt.image_decode(blob[:cut])           # truncated: an error or a partial picture, never a crash
            except rt.RtError:
                pass
        rs = np.random.RandomState(len(blob))
        for _ in range(40):                           # random corruption
            b = bytearray(blob)
            for _ in range(4):
                b[rs.randint(len(b))] = rs.randint(256)
            try:
                rt.image_decode(bytes(b))
            except rt.RtError:
                pass
    with pytest.raises(rt.RtError):
        rt.image_decode(b"GIF89a not an image we read")


def test_texture_file_entry_point(rt, tmp_path):
    s = rt.Scene()
    t = s.add_texture_file(os.path.join(IMG, "rgb420_64x48.jpg"))
    t2 = s.add_texture_file(os.path.join(IMG, "grey8_19x23.png"))      # grey is replicated to three channels
    assert (t, t2) == (0, 1) and s.flatten().n_textures == 2
    with pytest.raises(rt.RtError):
        s.add_texture_file(str(tmp_path / "missing.jpg"))
    with pytest.raises(rt.RtError):
        s.add_texture_file(os.path.join(IMG, "rgb_progressive_32x32.jpg"))


@pytest.mark.skipif(not os.path.isdir(REF_TEX), reason="the reference's assets exist in the build container only")
def test_brass_vase_maps_decode_to_stb_images_bytes(rt):
    meta = json.load(open(os.path.join(HERE, "golden", "stb_vase_maps.json")))
    z = np.load(os.path.join(HERE, "golden", "stb_vase_texels.npz"))
    for k, key in enumerate(z["names"]):
        m = meta[str(key)]
        blob = open(os.path.join(REF_TEX, m["file"]), "rb").read()
        assert hashlib.sha256(blob).hexdigest() == m["sha256_file"]
        img = rt.image_decode(blob)
        assert img.shape == (m["height"], m["width"], m["channels"])
        xy = z["xy"][k]
        assert np.array_equal(img[xy[:, 1], xy[:, 0]], z["rgb8"][k])
        assert hashlib.sha256(img.tobytes()).hexdigest() == m["sha256_rgb8"], key   # all 50 MB, byte for byte
