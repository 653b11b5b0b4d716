"""Generates the fixtures that pin the texture path against the reference's own decoder.

Runs in the build container only: it loads oracle/_ref/libstb_ref.so -- the reference's vendored
stb_image v2.27 compiled unmodified by `make -C oracle ref` -- and reads the brass vase's maps from
the reference tree.  What it writes is data (inputs and expected outputs), small enough to commit:

  stb_lut_bits.npy        the 256 float bit patterns stbi_loadf maps the byte values 0..255 to
                          (R/vendor/stb_image/stb_image.h:1840-1861, the LDR -> float map behind
                          Texture, R/src/Material/Texture.h:11)
  stb_vase_texels.npz     for the three maps the glTF binds (diffuse, normal, ARM): 4096 seeded
                          texel positions with stbi_load's bytes and stbi_loadf's float bits there
  stb_vase_maps.json      size, channel count and SHA-256 of each whole decoded map
  images/*.jpg|*.png      small synthetic images written with Pillow (this script's own data, not the
                          reference's), and images/expected.npz with stbi_load's bytes for each:
                          the known answers for the library's own JPEG / PNG decoder

    python tests/golden/make_stb_fixtures.py
"""
import ctypes as C
import hashlib
import io
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF_TEX = "/root/reference/Raytracing-In-A-Weekend/assets/models/brass_vase/textures"
# the three images the vase's glTF names (R/assets/models/brass_vase/brass_vase_04_4k.gltf:68-84)
VASE_MAPS = {"diffuse": "brass_vase_04_diff_4k.jpg", "normal": "brass_vase_04_nor_gl_4k.jpg",
             "arm": "brass_vase_04_arm_4k.jpg"}


def stb():
    L = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libstb_ref.so"))
    ip = C.POINTER(C.c_int)
    L.stbi_load_from_memory.restype = C.POINTER(C.c_ubyte)
    L.stbi_load_from_memory.argtypes = [C.c_char_p, C.c_int, ip, ip, ip, C.c_int]
    L.stbi_loadf_from_memory.restype = C.POINTER(C.c_float)
    L.stbi_loadf_from_memory.argtypes = [C.c_char_p, C.c_int, ip, ip, ip, C.c_int]
    L.stbi_image_free.argtypes = [C.c_void_p]
    L.stbi_failure_reason.restype = C.c_char_p
    return L


def load_u8(L, blob, req=0):
    """stbi_load: (H, W, channels) uint8, or None with the failure reason."""
    w, h, n = C.c_int(), C.c_int(), C.c_int()
    p = L.stbi_load_from_memory(blob, len(blob), C.byref(w), C.byref(h), C.byref(n), req)
    if not p:
        return None, L.stbi_failure_reason().decode()
    ch = req or n.value
    a = np.ctypeslib.as_array(p, shape=(h.value, w.value, ch)).copy()
    L.stbi_image_free(p)
    return a, n.value


def load_f32(L, blob, req=0):
    """stbi_loadf, as Texture's constructor calls it (req_comp = 0)."""
    w, h, n = C.c_int(), C.c_int(), C.c_int()
    p = L.stbi_loadf_from_memory(blob, len(blob), C.byref(w), C.byref(h), C.byref(n), req)
    if not p:
        return None, L.stbi_failure_reason().decode()
    ch = req or n.value
    a = np.ctypeslib.as_array(p, shape=(h.value, w.value, ch)).copy()
    L.stbi_image_free(p)
    return a, n.value


def lut_bits(L):
    # a 256x1 binary PPM holding every byte value once, r = g = b
    ppm = b"P6\n256 1\n255\n" + bytes(v for i in range(256) for v in (i, i, i))
    f, n = load_f32(L, ppm)
    assert n == 3 and f.shape == (1, 256, 3)
    bits = f.view(np.uint32)[0]
    assert (bits[:, 0] == bits[:, 1]).all() and (bits[:, 0] == bits[:, 2]).all()
    return bits[:, 0].copy()


def synthetic_images():
    """Small images that exercise the decoder paths stb_image has for baseline JPEG and PNG."""
    from PIL import Image
    rs = np.random.RandomState(2024)

    def picture(w, h, ch):
        yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
        base = np.stack([128 + 100 * np.sin(xx / 7.0 + c) * np.cos(yy / 5.0 - c) for c in range(ch)], -1)
        base += rs.randn(h, w, ch) * 12
        if w > 20 and h > 12:
            base[h // 3: h // 3 + 6, w // 4: w // 4 + 9] = 255 - base[h // 3: h // 3 + 6, w // 4: w // 4 + 9]
        return np.clip(base, 0, 255).astype(np.uint8)

    out = {}

    def jpeg(name, arr, **kw):
        b = io.BytesIO()
        Image.fromarray(arr.squeeze() if arr.shape[-1] == 1 else arr).save(b, "JPEG", **kw)
        out[name + ".jpg"] = b.getvalue()

    def png(name, img, **kw):
        b = io.BytesIO()
        img.save(b, "PNG", **kw)
        out[name + ".png"] = b.getvalue()

    jpeg("rgb420_64x48", picture(64, 48, 3), quality=90, subsampling=2)
    jpeg("rgb420_odd_37x29", picture(37, 29, 3), quality=75, subsampling=2)
    jpeg("rgb422_50x33", picture(50, 33, 3), quality=85, subsampling=1)
    jpeg("rgb444_41x30", picture(41, 30, 3), quality=95, subsampling=0)
    jpeg("grey_33x21", picture(33, 21, 1), quality=80)
    jpeg("rgb420_restart_70x40", picture(70, 40, 3), quality=88, subsampling=2, restart_marker_blocks=3)
    jpeg("rgb420_lowq_48x48", picture(48, 48, 3), quality=20, subsampling=2)
    jpeg("rgb420_optimised_56x40", picture(56, 40, 3), quality=92, subsampling=2, optimize=True)
    jpeg("rgb_progressive_32x32", picture(32, 32, 3), quality=90, progressive=True)
    jpeg("progressive_420_odd_45x37", picture(45, 37, 3), quality=80, subsampling=2, progressive=True)
    jpeg("progressive_444_lowq_40x24", picture(40, 24, 3), quality=15, subsampling=0, progressive=True)
    jpeg("progressive_422_highq_33x50", picture(33, 50, 3), quality=98, subsampling=1, progressive=True)
    jpeg("progressive_grey_27x19", picture(27, 19, 1), quality=70, progressive=True)
    jpeg("progressive_420_restart_72x56", picture(72, 56, 3), quality=85, subsampling=2, progressive=True, restart_marker_blocks=2)
    jpeg("progressive_420_large_200x120", picture(200, 120, 3), quality=60, subsampling=2, progressive=True, optimize=True)
    jpeg("progressive_444_tiny_3x2", picture(3, 2, 3), quality=90, subsampling=0, progressive=True)
    # four components, with the Adobe marker's colour-transform byte as written and set to each of its other values
    # (0: CMYK, 2: YCCK, anything else: the fourth channel is ignored) -- stb_image.h:3924-3947
    def cmyk(name, arr, **kw):
        b = io.BytesIO()
        Image.fromarray(arr, "CMYK").save(b, "JPEG", **kw)
        blob = bytearray(b.getvalue())
        at = blob.index(b"Adobe") + 11
        for t in (0, 1, 2):
            blob[at] = t
            out[f"{name}_transform{t}.jpg"] = bytes(blob)
    cmyk("cmyk_37x22", picture(37, 22, 4), quality=85)
    cmyk("cmyk_progressive_420_40x31", picture(40, 31, 4), quality=70, progressive=True, subsampling=2)
    png("rgb8_31x17", Image.fromarray(picture(31, 17, 3)))
    png("rgba8_20x20", Image.fromarray(picture(20, 20, 4)))
    png("grey8_19x23", Image.fromarray(picture(19, 23, 1).squeeze()))
    png("greyalpha8_16x9", Image.fromarray(picture(16, 9, 2), "LA"))
    png("palette_40x30", Image.fromarray(picture(40, 30, 3)).quantize(37))
    png("palette4bit_23x11", Image.fromarray(picture(23, 11, 3)).quantize(13), bits=4)
    png("bilevel_29x13", Image.fromarray(picture(29, 13, 1).squeeze() > 128))
    png("rgb16_12x10", Image.fromarray((picture(12, 10, 1).squeeze().astype(np.uint16) * 257 + 31).astype(np.uint16)))
    png("rgb8_uncompressed_25x14", Image.fromarray(picture(25, 14, 3)), compress_level=0)
    png("rgb8_large_300x200", Image.fromarray(picture(300, 200, 3)), compress_level=9)
    # Adam7-interlaced files (Pillow does not write them: encoded here, with filter types none / sub / up in turn)
    out["interlaced_rgb8_29x13.png"] = png_interlaced(picture(29, 13, 3), 2, 8)
    out["interlaced_rgba8_9x10.png"] = png_interlaced(picture(9, 10, 4), 6, 8)
    out["interlaced_grey8_3x5.png"] = png_interlaced(picture(3, 5, 1), 0, 8)
    out["interlaced_grey1_21x9.png"] = png_interlaced((picture(21, 9, 1) > 128).astype(np.uint8), 0, 1)
    out["interlaced_grey2_1x1.png"] = png_interlaced(np.array([[[2]]], np.uint8), 0, 2)
    out["interlaced_palette4_23x11.png"] = png_interlaced((picture(23, 11, 1) >> 4).astype(np.uint8), 3, 4,
                                                          palette=bytes(rs.randint(0, 256, 48).astype(np.uint8)))
    out["interlaced_rgb16_12x10.png"] = png_interlaced(picture(12, 10, 3).astype(np.uint16) * 257 + 31, 2, 16)
    return out


def png_interlaced(arr, color_type, depth, palette=None):
    """An Adam7-interlaced PNG of the (H, W, channels) sample array (PNG specification, sections 8.2 and 9)."""
    import struct
    import zlib
    h, w, ch = arr.shape
    bpp = max(1, ch * depth // 8)

    def packed(row):   # (w, ch) samples -> bytes
        flat = row.reshape(-1).astype(np.uint32)
        if depth == 8:
            return bytes(flat.astype(np.uint8))
        if depth == 16:
            return b"".join(struct.pack(">H", int(v)) for v in flat)
        bits = "".join(format(int(v), f"0{depth}b") for v in flat)
        bits += "0" * (-len(bits) % 8)
        return bytes(int(bits[i:i + 8], 2) for i in range(0, len(bits), 8))

    raw = bytearray()
    for x0, y0, dx, dy in ((0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)):
        sub = arr[y0::dy, x0::dx]
        if sub.shape[0] == 0 or sub.shape[1] == 0:
            continue
        prev = None
        for j in range(sub.shape[0]):
            cur = packed(sub[j])
            f = j % 3
            if f == 1:
                out = bytes((cur[i] - (cur[i - bpp] if i >= bpp else 0)) & 255 for i in range(len(cur)))
            elif f == 2:
                out = bytes((cur[i] - (prev[i] if prev is not None else 0)) & 255 for i in range(len(cur)))
            else:
                out = cur
            raw += bytes([f]) + out
            prev = cur

    def chunk(kind, body):
        return struct.pack(">I", len(body)) + kind + body + struct.pack(">I", zlib.crc32(kind + body) & 0xFFFFFFFF)

    blob = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, color_type, 0, 0, 1))
    if palette is not None:
        blob += chunk(b"PLTE", palette)
    return blob + chunk(b"IDAT", zlib.compress(bytes(raw), 6)) + chunk(b"IEND", b"")


def main():
    L = stb()
    np.save(os.path.join(HERE, "stb_lut_bits.npy"), lut_bits(L))

    rs = np.random.RandomState(4096)
    pos, u8s, f32s, meta = [], [], [], {}
    for key, name in VASE_MAPS.items():
        blob = open(os.path.join(REF_TEX, name), "rb").read()
        a, n = load_u8(L, blob)
        f, _ = load_f32(L, blob)
        h, w = a.shape[:2]
        xy = np.stack([rs.randint(0, w, 4096), rs.randint(0, h, 4096)], 1).astype(np.int32)
        # corners and edges are where resampling differs between decoders
        xy[:8] = [[0, 0], [w - 1, 0], [0, h - 1], [w - 1, h - 1], [1, 1], [w - 2, h - 2], [w // 2, 0], [0, h // 2]]
        pos.append(xy)
        u8s.append(a[xy[:, 1], xy[:, 0]])
        f32s.append(f[xy[:, 1], xy[:, 0]].view(np.uint32))
        meta[key] = {"file": name, "width": w, "height": h, "channels": n,
                     "sha256_rgb8": hashlib.sha256(a.tobytes()).hexdigest(),
                     "jpeg_bytes": len(blob), "sha256_file": hashlib.sha256(blob).hexdigest()}
    np.savez_compressed(os.path.join(HERE, "stb_vase_texels.npz"), names=np.array(list(VASE_MAPS)),
                        xy=np.stack(pos), rgb8=np.stack(u8s), f32bits=np.stack(f32s))
    json.dump(meta, open(os.path.join(HERE, "stb_vase_maps.json"), "w"), indent=1)

    img_dir = os.path.join(HERE, "images")
    os.makedirs(img_dir, exist_ok=True)
    expected = {}
    for name, blob in synthetic_images().items():
        path = os.path.join(img_dir, name)
        if os.path.exists(path):   # a committed fixture stays as it is, whatever this Pillow would write today
            blob = open(path, "rb").read()
        else:
            open(path, "wb").write(blob)
        a, n = load_u8(L, blob)
        if a is None:
            raise SystemExit(f"stb_image refused {name}: {n}")
        expected[name] = a
        expected[name + ":channels"] = np.int32(n)
    np.savez_compressed(os.path.join(img_dir, "expected.npz"), **expected)
    print("wrote", len(expected) // 2, "image fixtures;", {k: v["sha256_rgb8"][:12] for k, v in meta.items()})


if __name__ == "__main__":
    main()
