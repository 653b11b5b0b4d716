"""The C-ABI library loads without a GPU and exports every symbol the header declares."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "rtiow_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = set(re.findall(r"\b(rt_[a-z0-9_]+)\s*\(", text))
    names.discard("rt_pass_callback")
    return sorted(names)


def test_header_symbols_are_exported(rt):
    names = declared_symbols()
    assert len(names) >= 35
    for n in names:
        assert hasattr(rt.lib, n), f"{n} declared in include/rtiow_hip.h but not exported"
    # the binding table covers exactly the header
    assert sorted(rt.SYMBOLS) == names


def test_version_and_status_codes(rt):
    assert b"gfx950" in rt.lib.rt_version()
    assert rt.lib.rt_accum_bytes(1920, 1080, 1) == 240 * 135 * 64 * 16
    assert rt.lib.rt_accum_bytes(1920, 1080, 8) == 30 * 135 * 64 * 16
    assert rt.lib.rt_accum_bytes(10, 10, 1) == 2 * 2 * 64 * 16
    assert rt.lib.rt_accum_bytes(0, 10, 1) == 0


def test_device_entry_points_fail_loudly_without_a_gpu(rt):
    """No CPU fallback: without a device, context creation reports RT_ERR_NO_DEVICE."""
    if rt.device_count() > 0:
        pytest.skip("a HIP device is present")
    h = ctypes.c_void_p()
    rc = rt.lib.rt_create(0, ctypes.byref(h))
    assert rc == rt.RT_ERR_NO_DEVICE
    assert b"no CPU path" in rt.lib.rt_last_error()
    with pytest.raises(rt.RtError):
        rt.Context(0)
    # null-handle calls are rejected, not crashed on
    assert rt.lib.rt_render(None, 0, 1, 50) < 0
    assert rt.lib.rt_sync(None) < 0
    assert rt.lib.rt_upload_scene(None, None) < 0


def test_product_library_does_not_link_the_oracle(rt):
    import subprocess
    out = subprocess.run(["ldd", rt.LIB_PATH], capture_output=True, text=True).stdout
    assert "oracle" not in out
    syms = subprocess.run(["nm", "-D", "--defined-only", rt.LIB_PATH], capture_output=True, text=True).stdout
    assert "orc_" not in syms


def test_png_writer_round_trip(rt, tmp_path):
    """The PNG the library writes is read back by an independent reader (zlib + the chunk layout, here) and by
    the library's own decoder: RGB, 8 bits, top row first (the buffer's row 0 is the bottom row), alpha dropped;
    a frame larger than one stored deflate block (65,535 bytes) included."""
    import struct
    import zlib
    import numpy as np
    r = np.random.RandomState(1)
    for h, w in ((5, 7), (1, 1), (200, 333)):
        img = r.randint(0, 256, size=(h, w, 4)).astype(np.uint8)
        path = tmp_path / f"image_{w}x{h}.png"
        rt.write_png(str(path), img)
        data = path.read_bytes()
        assert data[:8] == b"\x89PNG\r\n\x1a\n"
        at, chunks = 8, []
        while at < len(data):
            n, typ = struct.unpack(">I4s", data[at:at + 8])
            body = data[at + 8:at + 8 + n]
            assert struct.unpack(">I", data[at + 8 + n:at + 12 + n])[0] == zlib.crc32(typ + body)
            chunks.append((typ, body))
            at += 12 + n
        assert [c[0] for c in chunks] == [b"IHDR", b"IDAT", b"IEND"]
        assert struct.unpack(">IIBBBBB", chunks[0][1]) == (w, h, 8, 2, 0, 0, 0)
        raw = np.frombuffer(zlib.decompress(chunks[1][1]), np.uint8).reshape(h, 1 + 3 * w)
        assert not raw[:, 0].any()
        assert np.array_equal(raw[:, 1:].reshape(h, w, 3), img[::-1, :, :3])
        assert np.array_equal(rt.image_decode(data), img[::-1, :, :3])
    with pytest.raises(rt.RtError):
        rt.write_png(str(tmp_path / "no_such_dir" / "x.png"), img)


def test_ppm_writer_round_trip(rt, tmp_path):
    """P3 header, 255 maximum, top row first (the buffer's row 0 is the bottom row), alpha dropped."""
    import numpy as np
    r = np.random.RandomState(0)
    img = r.randint(0, 256, size=(5, 7, 4)).astype(np.uint8)
    path = tmp_path / "image.ppm"
    rt.write_ppm(str(path), img)
    tok = path.read_text().split()
    assert tok[:4] == ["P3", "7", "5", "255"]
    vals = np.array(tok[4:], dtype=np.int64).reshape(5, 7, 3)
    assert np.array_equal(vals, img[::-1, :, :3])
    with pytest.raises(rt.RtError):
        rt.write_ppm(str(tmp_path / "no_such_dir" / "x.ppm"), img)
