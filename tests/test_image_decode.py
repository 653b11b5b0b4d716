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
    got = rt.image_decode(blob)
    assert got.shape == want.shape, (got.shape, want.shape)
    assert got.shape[2] == int(EXPECTED[name + ":channels"])
    assert np.array_equal(got, want), f"{name}: {(got != want).sum()} bytes differ, max |d| {np.abs(got.astype(int) - want).max()}"


def test_every_decoder_path_has_a_fixture():
    kinds = " ".join(NAMES)
    for needle in ("rgb420", "rgb422", "rgb444", "grey_", "restart", "odd", "optimised", "rgba8", "palette", "bilevel",
                   "rgb16", "greyalpha", "uncompressed", "progressive_420", "progressive_422", "progressive_444",
                   "progressive_grey", "progressive_420_restart", "interlaced_rgb8", "interlaced_grey1", "interlaced_palette4",
                   "interlaced_rgb16", "interlaced_rgba8", "cmyk_37x22_transform0", "transform1", "transform2", "cmyk_progressive"):
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
    t3 = s.add_texture_file(os.path.join(IMG, "rgb_progressive_32x32.jpg"))
    t4 = s.add_texture_file(os.path.join(IMG, "interlaced_rgb8_29x13.png"))
    assert (t3, t4) == (2, 3)
    # a frame header that announces arithmetic coding (SOF9 in place of SOF0): refused loudly, never mis-decoded
    blob = bytearray(open(os.path.join(IMG, "rgb420_64x48.jpg"), "rb").read())
    at = blob.index(b"\xff\xc0")
    blob[at + 1] = 0xC9
    bad = tmp_path / "arithmetic.jpg"
    bad.write_bytes(bytes(blob))
    with pytest.raises(rt.RtError) as e:
        s.add_texture_file(str(bad))
    assert "arithmetic" in str(e.value)


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


VASE_GLTF = os.path.join(HERE, "golden", "brass_vase", "brass_vase_04_4k.gltf")


def test_mesh_file_loads_the_vase_with_its_three_maps(rt, scenes):
    """rt_scene_add_mesh_file = Mesh(model, path): geometry, the three JPEG maps the glTF names and the
    one PBR material the reference builds from them (R/src/Core/Mesh.cpp:56-104) -- the same scene as
    putting it together by hand from the separate calls."""
    a = rt.Scene()
    first = a.add_mesh_file(scenes.VASE_MODEL, VASE_GLTF)
    fa = a.get_flat()
    assert first == 0 and fa["info"].n_triangles == 2760 and fa["info"].n_textures == 3 and fa["info"].n_materials == 1
    assert fa["mat_kind"][0] == rt.MAT_PBR and (fa["tri_mat"] == 0).all()
    b = rt.Scene()
    tex = os.path.join(os.path.dirname(VASE_GLTF), "textures")
    td = b.add_texture_file(os.path.join(tex, "brass_vase_04_diff_4k.jpg"))     # baseColorTexture
    tr = b.add_texture_file(os.path.join(tex, "brass_vase_04_arm_4k.jpg"))      # metallicRoughnessTexture
    tn = b.add_texture_file(os.path.join(tex, "brass_vase_04_nor_gl_4k.jpg"))   # normalTexture
    b.add_mesh_gltf(scenes.VASE_MODEL, VASE_GLTF, b.add_pbr(td, tr, tn))
    fb = b.get_flat()
    for k in ("nodes", "prim_refs", "tri_pos", "tri_normal", "tri_uv", "tri_mat", "mat_kind"):
        assert np.array_equal(fa[k], fb[k]), k
    # the textures the maps decode to are the reference decoder's (checked whole in
    # test_brass_vase_maps_decode_to_stb_images_bytes where the reference tree exists; here by sample)
    z = np.load(os.path.join(HERE, "golden", "stb_vase_texels.npz"))
    img = rt.image_decode(open(os.path.join(tex, "brass_vase_04_diff_4k.jpg"), "rb").read())
    xy = z["xy"][0]
    assert np.array_equal(img[xy[:, 1], xy[:, 0]], z["rgb8"][0])


def write_two_mesh_gltf(tmp_path, with_texture=True):
    """A hand-made file: root node 0 (mesh 1) with child node 1 (mesh 0); mesh 0 has two primitives
    (materials 0 and 1), mesh 1 one primitive (material 1); u8, u16 and no indices."""
    import struct
    quad = np.float32([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0]])
    nrm = np.float32([[0, 0, 1]] * 4)
    uv = np.float32([[0, 0], [1, 0], [1, 1], [0, 1]])
    blob = bytearray()

    def put(arr):
        off = len(blob)
        blob.extend(arr.tobytes())
        while len(blob) % 4:
            blob.append(0)
        return off, len(arr.tobytes())

    views, accs = [], []

    def acc(arr, ctype, typ):
        off, n = put(arr)
        views.append({"buffer": 0, "byteOffset": off, "byteLength": n})
        accs.append({"bufferView": len(views) - 1, "componentType": ctype, "count": len(arr), "type": typ})
        return len(accs) - 1

    prims = []
    for k, (shift, idx) in enumerate(((0.0, np.uint8([0, 1, 2, 0, 2, 3])), (2.0, np.uint16([0, 1, 2, 0, 2, 3])), (4.0, None))):
        pos = quad + np.float32([shift, 0, 0])
        if idx is None:
            pos, n2, uv2 = pos[[0, 1, 2, 0, 2, 3]], nrm[[0, 1, 2, 0, 2, 3]], uv[[0, 1, 2, 0, 2, 3]]
        else:
            n2, uv2 = nrm, uv
        p = {"attributes": {"POSITION": acc(pos, 5126, "VEC3"), "NORMAL": acc(n2, 5126, "VEC3"),
                            "TEXCOORD_0": acc(uv2, 5126, "VEC2")}, "material": [0, 1, 1][k]}
        if idx is not None:
            p["indices"] = acc(idx, 5121 if idx.dtype == np.uint8 else 5123, "SCALAR")
        prims.append(p)
    g = {"asset": {"version": "2.0"}, "scene": 0, "scenes": [{"nodes": [0]}],
         "nodes": [{"mesh": 1, "children": [1]}, {"mesh": 0, "translation": [100, 0, 0]}],
         "meshes": [{"primitives": prims[:2]}, {"primitives": prims[2:]}],
         "materials": [{"pbrMetallicRoughness": {"baseColorFactor": [0.8, 0.1, 0.2, 1.0]}},
                       {"pbrMetallicRoughness": ({"baseColorTexture": {"index": 0}} if with_texture else {})}],
         "textures": [{"source": 0}], "images": [{"uri": "tex.png"}],
         "buffers": [{"uri": "data.bin", "byteLength": len(blob)}], "bufferViews": views, "accessors": accs}
    d = tmp_path / "two"
    d.mkdir(parents=True)
    json.dump(g, open(d / "two.gltf", "w"))
    open(d / "data.bin", "wb").write(bytes(blob))
    import shutil
    shutil.copy(os.path.join(IMG, "rgb8_31x17.png"), d / "tex.png")
    return str(d / "two.gltf")


def test_multi_primitive_file_in_both_material_modes(rt, tmp_path):
    path = write_two_mesh_gltf(tmp_path)
    # the reference's rule: one shared PBR material from the first base-colour texture met
    a = rt.Scene()
    a.add_mesh_file(rt.IDENTITY, path, max_depth=-1)
    fa = a.get_flat()
    assert fa["info"].n_triangles == 6 and fa["info"].n_materials == 1 and fa["info"].n_textures == 1
    assert fa["mat_kind"][0] == rt.MAT_PBR and (fa["tri_mat"] == 0).all()
    # visiting order: the root node's mesh (mesh 1, x in [4, 5]) first, then the child's (mesh 0); the
    # node's translation is NOT applied (the reference never reads node transforms)
    x0 = fa["tri_pos"][:, 0::3].min(axis=1)
    assert list(x0) == [4.0, 4.0, 0.0, 0.0, 2.0, 2.0]
    # per-primitive materials (beyond the reference)
    b = rt.Scene()
    b.add_mesh_file(rt.IDENTITY, path, max_depth=-1, per_primitive_materials=True)
    fb = b.get_flat()
    assert fb["info"].n_materials == 3 and fb["info"].n_textures == 1          # the image is decoded once
    assert list(fb["mat_kind"]) == [rt.MAT_PBR, rt.MAT_LAMBERTIAN, rt.MAT_PBR]
    assert np.allclose(fb["mat_rgbp"][1][:3], [0.8, 0.1, 0.2])
    assert list(fb["tri_mat"]) == [0, 0, 1, 1, 2, 2]
    # a caller's material for everything: no image is read
    c = rt.Scene()
    c.add_mesh_file(rt.IDENTITY, path, material=c.add_material(0, (0.5, 0.5, 0.5)))
    assert c.flatten().n_textures == 0 and (c.get_flat()["tri_mat"] == 0).all()
    # no base-colour texture anywhere: the reference would dereference a null material; refused here
    with pytest.raises(rt.RtError):
        rt.Scene().add_mesh_file(rt.IDENTITY, write_two_mesh_gltf(tmp_path / "x", with_texture=False))
