"""Malformed meshes and glTF files are refused with RT_ERR_INVALID instead of being read out of
bounds (host code only; no GPU)."""
import json
import os
import shutil

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
VASE = os.path.join(HERE, "golden", "brass_vase")


def test_mesh_index_beyond_the_vertex_array_is_refused(rt):
    s = rt.Scene()
    m = s.add_material(0, (0.5, 0.5, 0.5))
    pos = np.float32([[0, 0, 0], [1, 0, 0], [0, 1, 0]])
    for bad in (3, 0x7FFFFFFF, 0x80000000, 0xFFFFFFFF):   # values >= 2^31 are negative as int
        idx = np.array([[0, 1, bad]], np.uint32)
        with pytest.raises(rt.RtError) as e:
            s.add_mesh(rt.IDENTITY, pos, idx, m)
        assert e.value.code == -1
    s.add_mesh(rt.IDENTITY, pos, np.array([[0, 1, 2]], np.uint32), m)   # and the valid one still loads


def edited_vase(tmp_path, edit):
    d = tmp_path / "vase"
    shutil.copytree(VASE, d)
    p = d / "brass_vase_04_4k.gltf"
    g = json.load(open(p))
    edit(g)
    json.dump(g, open(p, "w"))
    return str(p)


@pytest.mark.parametrize("name,edit", [
    ("bufferView index out of range", lambda g: g["accessors"][0].__setitem__("bufferView", 99)),
    ("negative bufferView", lambda g: g["accessors"][1].__setitem__("bufferView", -1)),
    ("buffer index out of range", lambda g: g["bufferViews"][0].__setitem__("buffer", 7)),
    ("indices accessor out of range", lambda g: g["meshes"][0]["primitives"][0].__setitem__("indices", 1000)),
    ("indices accessor not a number", lambda g: g["meshes"][0]["primitives"][0].__setitem__("indices", "3")),
    ("index accessor without componentType", lambda g: g["accessors"][3].pop("componentType")),
    ("POSITION accessor out of range", lambda g: g["meshes"][0]["primitives"][0]["attributes"].__setitem__("POSITION", 1e12)),
    ("count beyond the buffer", lambda g: g["accessors"][0].__setitem__("count", 10_000_000)),
    ("byteOffset beyond the buffer", lambda g: g["bufferViews"][3].__setitem__("byteOffset", 1 << 40)),
    ("negative byteOffset", lambda g: g["bufferViews"][0].__setitem__("byteOffset", -64)),
    ("stride smaller than the element", lambda g: g["bufferViews"][0].__setitem__("byteStride", 4)),
])
def test_malformed_gltf_is_refused(rt, tmp_path, name, edit):
    s = rt.Scene()
    m = s.add_material(0, (0.5, 0.5, 0.5))
    with pytest.raises(rt.RtError) as e:
        s.add_mesh_gltf(rt.IDENTITY, edited_vase(tmp_path, edit), m)
    assert e.value.code == -1, name


def test_gltf_with_an_out_of_range_vertex_index_is_refused(rt, tmp_path):
    d = tmp_path / "vase"
    shutil.copytree(VASE, d)
    g = json.load(open(d / "brass_vase_04_4k.gltf"))
    acc = g["accessors"][g["meshes"][0]["primitives"][0]["indices"]]
    view = g["bufferViews"][acc["bufferView"]]
    blob = bytearray(open(d / "brass_vase_04.bin", "rb").read())
    off = view.get("byteOffset", 0) + acc.get("byteOffset", 0)
    blob[off + 10: off + 12] = b"\xff\xff"    # u16 index 65535 of 1614 vertices
    open(d / "brass_vase_04.bin", "wb").write(blob)
    s = rt.Scene()
    with pytest.raises(rt.RtError):
        s.add_mesh_gltf(rt.IDENTITY, str(d / "brass_vase_04_4k.gltf"), s.add_material(0, (0.5, 0.5, 0.5)))


def test_the_unedited_vase_still_loads(rt):
    s = rt.Scene()
    first = s.add_mesh_gltf(rt.IDENTITY, os.path.join(VASE, "brass_vase_04_4k.gltf"), s.add_material(0, (0.5, 0.5, 0.5)))
    assert first == 0 and s.flatten().n_triangles == 2760


def test_jpeg_that_ends_inside_its_scan_header_is_refused(rt):
    """A file whose last bytes are FF DA 00 02 (a scan segment of length 2, i.e. no header bytes at all) after a
    valid frame header: the component count of the scan must not be read past the end of the buffer."""
    names = sorted(n for n in os.listdir(os.path.join(HERE, "golden", "images")) if n.endswith(".jpg") and "progressive" not in n)
    src = open(os.path.join(HERE, "golden", "images", names[0]), "rb").read()
    sos = src.index(b"\xff\xda")
    bad = src[:sos] + b"\xff\xda\x00\x02"
    with pytest.raises(rt.RtError) as e:
        rt.image_decode(bad)
    assert e.value.code == -1


def test_png_whose_data_inflates_far_beyond_its_header_is_refused(rt):
    """A 1x1 PNG with 8 MB of zeros in its IDAT (8 KB compressed): inflate stops at about twice the size the
    header announces instead of expanding whatever the stream holds."""
    import struct
    import zlib

    def chunk(kind, data):
        return struct.pack(">I", len(data)) + kind + data + struct.pack(">I", zlib.crc32(kind + data) & 0xFFFFFFFF)
    png = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", 1, 1, 8, 0, 0, 0, 0)) \
        + chunk(b"IDAT", zlib.compress(b"\x00" * (8 << 20), 9)) + chunk(b"IEND", b"")
    with pytest.raises(rt.RtError) as e:
        rt.image_decode(png)
    assert e.value.code == -1 and "more image data" in str(e.value)
    # the same image with exactly its two bytes of data decodes
    ok = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", 1, 1, 8, 0, 0, 0, 0)) \
        + chunk(b"IDAT", zlib.compress(b"\x00\x7f")) + chunk(b"IEND", b"")
    assert rt.image_decode(ok).tolist() == [[[0x7f]]]
