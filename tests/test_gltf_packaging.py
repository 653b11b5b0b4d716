"""The glTF reader takes a file's buffers and images wherever the format allows them to be: in files beside the
.gltf, in "data:" URIs inside it, or in the binary chunk of a .glb container (glTF 2.0 specification, sections 3.6.1.1,
3.9.4.1 and 4.4).  One small textured mesh, packaged the three ways, must give the same scene -- and, on the GPU,
the same picture."""
import base64
import json
import os
import shutil
import struct

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
IMG = os.path.join(HERE, "golden", "images")
# base colour: a plain PNG; metallic-roughness: an interlaced PNG; normal: a progressive JPEG
IMAGES = ["rgb8_31x17.png", "interlaced_rgb8_29x13.png", "rgb_progressive_32x32.jpg"]


def packagings(scenes, tmp_path):
    m = scenes.lathe_mesh(9, 6, seed=3)
    pos, nrm, uv = m["positions"].astype(np.float32), m["normals"].astype(np.float32), m["uvs"].astype(np.float32)
    idx = m["indices"].astype(np.uint16)
    parts = [pos.tobytes(), nrm.tobytes(), uv.tobytes(), idx.tobytes()]
    offsets, blob = [], b""
    for p in parts:
        offsets.append(len(blob))
        blob += p + b"\0" * (-len(p) % 4)
    views = [{"buffer": 0, "byteOffset": o, "byteLength": len(p)} for o, p in zip(offsets, parts)]
    doc = {
        "asset": {"version": "2.0"},
        "scene": 0, "scenes": [{"nodes": [0]}], "nodes": [{"mesh": 0}],
        "meshes": [{"primitives": [{"attributes": {"POSITION": 0, "NORMAL": 1, "TEXCOORD_0": 2}, "indices": 3, "material": 0}]}],
        "accessors": [
            {"bufferView": 0, "componentType": 5126, "count": len(pos), "type": "VEC3"},
            {"bufferView": 1, "componentType": 5126, "count": len(nrm), "type": "VEC3"},
            {"bufferView": 2, "componentType": 5126, "count": len(uv), "type": "VEC2"},
            {"bufferView": 3, "componentType": 5123, "count": idx.size, "type": "SCALAR"}],
        "materials": [{"pbrMetallicRoughness": {"baseColorTexture": {"index": 0}, "metallicRoughnessTexture": {"index": 1}},
                       "normalTexture": {"index": 2}}],
        "textures": [{"source": 0}, {"source": 1}, {"source": 2}],
    }
    files = [open(os.path.join(IMG, n), "rb").read() for n in IMAGES]

    # (a) everything in files beside the .gltf
    a = tmp_path / "files"
    a.mkdir()
    for n in IMAGES:
        shutil.copy(os.path.join(IMG, n), a / n)
    (a / "mesh.bin").write_bytes(blob)
    da = dict(doc, bufferViews=views, buffers=[{"uri": "mesh.bin", "byteLength": len(blob)}], images=[{"uri": n} for n in IMAGES])
    (a / "mesh.gltf").write_text(json.dumps(da))

    # (b) everything in data: URIs
    def uri(mime, data):
        return f"data:{mime};base64," + base64.b64encode(data).decode()
    db = dict(doc, bufferViews=views, buffers=[{"uri": uri("application/octet-stream", blob), "byteLength": len(blob)}],
              images=[{"uri": uri("image/png" if n.endswith("png") else "image/jpeg", f)} for n, f in zip(IMAGES, files)])
    (tmp_path / "embedded.gltf").write_text(json.dumps(db))

    # (c) a .glb: the JSON chunk, and one binary chunk that holds the mesh data and the three image files
    glb_views, glb_blob = list(views), blob
    for f in files:
        glb_views.append({"buffer": 0, "byteOffset": len(glb_blob), "byteLength": len(f)})
        glb_blob += f + b"\0" * (-len(f) % 4)
    dc = dict(doc, bufferViews=glb_views, buffers=[{"byteLength": len(glb_blob)}],
              images=[{"bufferView": 4 + k, "mimeType": "image/png" if n.endswith("png") else "image/jpeg"} for k, n in enumerate(IMAGES)])
    js = json.dumps(dc).encode()
    js += b" " * (-len(js) % 4)
    glb = struct.pack("<4sII", b"glTF", 2, 12 + 8 + len(js) + 8 + len(glb_blob)) + struct.pack("<II", len(js), 0x4E4F534A) + js \
        + struct.pack("<II", len(glb_blob), 0x004E4942) + glb_blob
    (tmp_path / "container.glb").write_bytes(glb)
    return [str(a / "mesh.gltf"), str(tmp_path / "embedded.gltf"), str(tmp_path / "container.glb")]


def load(rt, path):
    s = rt.Scene()
    s.add_mesh_file(rt.IDENTITY, path, max_depth=6)
    s.set_camera((0.0, 0.1, 0.4), (0.0, 0.08, 0.0), (0.0, 1.0, 0.0), 40.0, 16.0 / 9.0, 0.0, 1.0)
    s.set_background((0.7, 0.8, 1.0))
    return s


def test_three_packagings_give_one_scene(rt, scenes, tmp_path):
    flats = [load(rt, p).get_flat() for p in packagings(scenes, tmp_path)]
    assert flats[0]["info"].n_textures == 3 and flats[0]["info"].n_triangles == 9 * 6 * 2
    for other in flats[1:]:
        assert other["info"].n_textures == 3
        for key in ("nodes", "prim_refs", "tri_pos", "tri_normal", "tri_uv", "tri_mat", "mat_kind", "mat_rgbp"):
            assert np.array_equal(flats[0][key].view(np.uint32), other[key].view(np.uint32)), key


def test_broken_packagings_are_refused(rt, scenes, tmp_path):
    files, embedded, glb = packagings(scenes, tmp_path)
    doc = json.loads(open(embedded).read())
    doc["buffers"][0]["uri"] = "data:application/octet-stream;base64,@@@not base64@@@"
    bad = tmp_path / "bad.gltf"
    bad.write_text(json.dumps(doc))
    with pytest.raises(rt.RtError):
        load(rt, str(bad))
    blob = open(glb, "rb").read()
    for cut in (10, 40, len(blob) - 5):
        short = tmp_path / f"cut{cut}.glb"
        short.write_bytes(blob[:cut])
        with pytest.raises(rt.RtError):
            load(rt, str(short))


@pytest.mark.gpu
def test_three_packagings_give_one_picture(rt, scenes, gpu_ctx, tmp_path):
    images = []
    for p in packagings(scenes, tmp_path):
        gpu_ctx.upload_scene(load(rt, p))
        gpu_ctx.set_target(96, 54)
        gpu_ctx.render(0, 3, 8)
        gpu_ctx.sync()
        images.append(gpu_ctx.read_accum())
    assert images[0][..., :3].max() > 0
    assert np.array_equal(images[0].view(np.uint32), images[1].view(np.uint32))
    assert np.array_equal(images[0].view(np.uint32), images[2].view(np.uint32))


def write_gltf(path, positions, uvs, indices, mode=None, uv_type=np.float32):
    """One untextured primitive: positions (float), uvs of the given component type, u16 indices (or none)."""
    parts = [positions.astype(np.float32).tobytes(), uvs.astype(uv_type).tobytes()]
    if indices is not None:
        parts.append(np.asarray(indices, np.uint16).tobytes())
    offsets, blob = [], b""
    for p in parts:
        offsets.append(len(blob))
        blob += p + b"\0" * (-len(p) % 4)
    ct = {np.float32: 5126, np.uint16: 5123, np.uint8: 5121}[uv_type]
    prim = {"attributes": {"POSITION": 0, "TEXCOORD_0": 1}}
    acc = [{"bufferView": 0, "componentType": 5126, "count": len(positions), "type": "VEC3"},
           {"bufferView": 1, "componentType": ct, "count": len(uvs), "type": "VEC2", "normalized": ct != 5126}]
    if indices is not None:
        prim["indices"] = 2
        acc.append({"bufferView": 2, "componentType": 5123, "count": len(indices), "type": "SCALAR"})
    if mode is not None:
        prim["mode"] = mode
    doc = {"asset": {"version": "2.0"}, "meshes": [{"primitives": [prim]}], "accessors": acc,
           "bufferViews": [{"buffer": 0, "byteOffset": o, "byteLength": len(p)} for o, p in zip(offsets, parts)],
           "buffers": [{"uri": "data:application/octet-stream;base64," + base64.b64encode(blob).decode(), "byteLength": len(blob)}]}
    path.write_text(json.dumps(doc))
    return str(path)


def triangles_of(rt, path):
    s = rt.Scene()
    s.add_mesh_file(rt.IDENTITY, path, max_depth=4, material=s.add_material(0, (0.5, 0.5, 0.5)))
    d = s.get_flat()
    return d["tri_pos"].view(np.uint32), d["tri_uv"].view(np.uint32)


def test_strips_fans_and_normalised_texture_coordinates(rt, tmp_path):
    rs = np.random.RandomState(11)
    pos = rs.rand(7, 3).astype(np.float32)
    uv16 = rs.randint(0, 65536, (7, 2)).astype(np.uint16)
    uv = uv16.astype(np.float32) / np.float32(65535.0)
    seq = [3, 0, 5, 1, 6, 2, 4]
    strip = [[seq[i + (i & 1)], seq[i + 1 - (i & 1)], seq[i + 2]] for i in range(5)]
    fan = [[seq[0], seq[i + 1], seq[i + 2]] for i in range(5)]
    for mode, want in ((5, strip), (6, fan)):
        a = triangles_of(rt, write_gltf(tmp_path / f"mode{mode}.gltf", pos, uv, seq, mode=mode))
        b = triangles_of(rt, write_gltf(tmp_path / f"list{mode}.gltf", pos, uv, np.array(want).reshape(-1)))
        assert a[0].shape[0] == 5 and np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    # without indices the vertices themselves are the sequence
    a = triangles_of(rt, write_gltf(tmp_path / "strip_plain.gltf", pos, uv, None, mode=5))
    b = triangles_of(rt, write_gltf(tmp_path / "strip_plain_list.gltf", pos, uv,
                                    np.array([[i + (i & 1), i + 1 - (i & 1), i + 2] for i in range(5)]).reshape(-1)))
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    # unsigned short / byte texture coordinates stand for value / 65535 and value / 255
    f = triangles_of(rt, write_gltf(tmp_path / "uv_float.gltf", pos, uv, seq[:6]))
    q = triangles_of(rt, write_gltf(tmp_path / "uv_u16.gltf", pos, uv16, seq[:6], uv_type=np.uint16))
    assert np.array_equal(f[0], q[0]) and np.array_equal(f[1], q[1])
    uv8 = rs.randint(0, 256, (7, 2)).astype(np.uint8)
    f = triangles_of(rt, write_gltf(tmp_path / "uv_float8.gltf", pos, uv8.astype(np.float32) / np.float32(255.0), seq[:6]))
    q = triangles_of(rt, write_gltf(tmp_path / "uv_u8.gltf", pos, uv8, seq[:6], uv_type=np.uint8))
    assert np.array_equal(f[1], q[1])
    with pytest.raises(rt.RtError):      # lines have no surface
        triangles_of(rt, write_gltf(tmp_path / "lines.gltf", pos, uv, seq, mode=1))
