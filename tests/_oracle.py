"""ctypes binding of the CPU oracle (oracle/liboracle.so).  Test infrastructure only."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.path.join(ROOT, "oracle", "liboracle.so")


class OrcCounters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in (
        "samples", "rays", "node_tests", "sphere_tests", "tri_tests", "hits",
        "scatter_lambertian", "scatter_metal", "scatter_dielectric", "scatter_light", "scatter_pbr",
        "misses", "depth_cutoffs")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} missing: run `make -C oracle`")
        L = C.CDLL(LIB_PATH)
        L.orc_scene_new.restype = C.c_void_p
        L.orc_hash_random.restype = C.c_float
        L.orc_hash_random.argtypes = [C.c_uint32, C.c_uint32, C.c_int32, C.c_uint32]
        for name in ("orc_scene_free", "orc_scene_random", "orc_scene_cornell", "orc_wrap_bvh"):
            getattr(L, name).argtypes = [C.c_void_p]
        L.orc_add_material.argtypes = [C.c_void_p, C.c_int] + [C.c_float] * 4 + [C.c_int] * 3
        L.orc_add_texture.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.orc_add_sphere.argtypes = [C.c_void_p] + [C.c_float] * 4 + [C.c_int]
        L.orc_add_triangle.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.orc_add_mesh.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                   C.c_void_p, C.c_int, C.c_int]
        L.orc_begin_prototype.argtypes = [C.c_void_p]
        L.orc_end_prototype.argtypes = [C.c_void_p]
        L.orc_add_instance.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.orc_set_camera.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_float] * 4
        L.orc_get_camera.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_set_background.argtypes = [C.c_void_p] + [C.c_float] * 3
        L.orc_set_normal_mapping.argtypes = [C.c_void_p, C.c_int]
        L.orc_rng_floats.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        for name in ("orc_num_spheres", "orc_num_triangles", "orc_num_materials"):
            getattr(L, name).argtypes = [C.c_void_p]
        L.orc_dump_spheres.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_dump_triangles.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_dump_materials.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_dump_tree.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_void_p, C.c_void_p,
                                    C.c_void_p]
        L.orc_trace.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 7
        L.orc_scatter.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_float,
                                  C.c_float, C.c_uint32, C.c_uint32, C.c_int32, C.c_uint32, C.c_void_p, C.c_void_p,
                                  C.POINTER(C.c_uint32)]
        L.orc_texture_at.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_texture_lut.argtypes = [C.c_void_p]
        L.orc_reflectance.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_write_color.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        L.orc_render.argtypes = [C.c_void_p] + [C.c_int] * 10 + [C.c_void_p, C.c_void_p, C.POINTER(OrcCounters)]
        _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


class OracleScene:
    """Same building vocabulary as the product's Scene, backed by the oracle."""

    def __init__(self):
        self.L = lib()
        self.h = C.c_void_p(self.L.orc_scene_new())

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orc_scene_free(self.h)
            self.h = None

    def rng_floats(self, n):
        out = np.zeros(n, np.float32)
        self.L.orc_rng_floats(self.h, n, _p(out))
        return out

    def add_material(self, kind, rgb=(0, 0, 0), param=0.0):
        return self.L.orc_add_material(self.h, kind, rgb[0], rgb[1], rgb[2], param, -1, -1, -1)

    def add_texture(self, rgb8):
        a = np.ascontiguousarray(rgb8, dtype=np.uint8)
        return self.L.orc_add_texture(self.h, a.shape[1], a.shape[0], _p(a))

    def add_pbr(self, td, tr=-1, tn=-1):
        return self.L.orc_add_material(self.h, 4, 0, 0, 0, 0, td, tr, tn)

    def add_sphere(self, c, r, m):
        return self.L.orc_add_sphere(self.h, c[0], c[1], c[2], r, m)

    def add_triangle(self, pos9, material, normal9=None, uv6=None):
        pos = _f32(pos9)
        nrm = _f32(np.zeros(9) if normal9 is None else normal9)
        uv = _f32(np.zeros(6) if uv6 is None else uv6)
        return self.L.orc_add_triangle(self.h, _p(pos), _p(nrm), _p(uv), material)

    def add_mesh(self, model, positions, indices, material, normals=None, uvs=None, max_depth=10):
        m, pos = _f32(model), _f32(positions)
        idx = np.ascontiguousarray(indices, dtype=np.uint32)
        nrm = None if normals is None else _f32(normals)
        uv = None if uvs is None else _f32(uvs)
        self.L.orc_add_mesh(self.h, _p(m), pos.size // 3, _p(pos), _p(nrm), _p(uv), idx.size // 3, _p(idx), material,
                            max_depth)

    def begin_prototype(self):
        return self.L.orc_begin_prototype(self.h)

    def end_prototype(self):
        return self.L.orc_end_prototype(self.h)

    def add_instance(self, prototype, model):
        m = _f32(model)
        rc = self.L.orc_add_instance(self.h, prototype, _p(m))
        assert rc >= 0
        return rc

    def random_scene(self):
        self.L.orc_scene_random(self.h)

    def cornell_box(self):
        self.L.orc_scene_cornell(self.h)

    def wrap_bvh(self):
        self.L.orc_wrap_bvh(self.h)

    def set_camera(self, lookfrom, lookat, vup, vfov, aspect, aperture, focus):
        a, b, c = _f32(lookfrom), _f32(lookat), _f32(vup)
        self.L.orc_set_camera(self.h, _p(a), _p(b), _p(c), vfov, aspect, aperture, focus)

    def get_camera(self):
        out = np.zeros(22, np.float32)
        self.L.orc_get_camera(self.h, _p(out))
        return out

    def set_background(self, rgb):
        self.L.orc_set_background(self.h, rgb[0], rgb[1], rgb[2])

    def set_normal_mapping(self, on):
        self.L.orc_set_normal_mapping(self.h, 1 if on else 0)

    def path_rays(self, W, H, px, py, sample, max_depth=50):
        """(rays Nx6, hits Nx2 {kind, id}, t N) of one hash-stream path."""
        rays = np.zeros((max_depth, 6), np.float32)
        hits = np.zeros((max_depth, 2), np.int32)
        t = np.zeros(max_depth, np.float32)
        self.L.orc_path_rays.argtypes = [C.c_void_p] + [C.c_int] * 7 + [C.c_void_p] * 3
        n = self.L.orc_path_rays(self.h, W, H, px, py, sample, max_depth, max_depth, _p(rays), _p(hits), _p(t))
        return rays[:n], hits[:n], t[:n]

    def dump(self):
        ns, nt, nm = (self.L.orc_num_spheres(self.h), self.L.orc_num_triangles(self.h),
                      self.L.orc_num_materials(self.h))
        d = {"spheres": np.zeros((ns, 4), np.float32), "sphere_mat": np.zeros(ns, np.int32),
             "tris": np.zeros((nt, 24), np.float32), "tri_mat": np.zeros(nt, np.int32),
             "mat_kind": np.zeros(nm, np.int32), "mat_rgbp": np.zeros((nm, 4), np.float32)}
        self.L.orc_dump_spheres(self.h, _p(d["spheres"]), _p(d["sphere_mat"]))
        self.L.orc_dump_triangles(self.h, _p(d["tris"]), _p(d["tri_mat"]))
        self.L.orc_dump_materials(self.h, _p(d["mat_kind"]), _p(d["mat_rgbp"]))
        return d

    def dump_tree(self):
        nb, npr = C.c_int(), C.c_int()
        self.L.orc_dump_tree(self.h, C.byref(nb), C.byref(npr), None, None, None)
        boxes = np.zeros((nb.value, 6), np.float32)
        desc = np.zeros(nb.value, np.int32)
        prims = np.zeros((npr.value, 2), np.int32)
        self.L.orc_dump_tree(self.h, C.byref(nb), C.byref(npr), _p(boxes), _p(desc), _p(prims))
        return boxes, desc, prims

    def trace(self, rays):
        rays = _f32(rays).reshape(-1, 6)
        n = rays.shape[0]
        kind, pid = np.zeros(n, np.int32), np.zeros(n, np.int32)
        t = np.zeros(n, np.float32)
        nrm = np.zeros((n, 3), np.float32)
        front = np.zeros(n, np.int32)
        uv = np.zeros((n, 2), np.float32)
        self.L.orc_trace(self.h, n, _p(rays), _p(kind), _p(pid), _p(t), _p(nrm), _p(front), _p(uv))
        return kind, pid, t, nrm, front, uv

    def scatter(self, mat, ray_in, p, normal, front, u, v, px, py, sample, counter0):
        ray_in, p, normal = _f32(ray_in), _f32(p), _f32(normal)
        att = np.zeros(3, np.float32)
        out = np.zeros(6, np.float32)
        cnt = C.c_uint32()
        alive = self.L.orc_scatter(self.h, mat, _p(ray_in), _p(p), _p(normal), int(front), float(u), float(v),
                                   int(px), int(py), int(sample), int(counter0), _p(att), _p(out), C.byref(cnt))
        return att, out, alive, cnt.value

    def texture_at(self, tex, uv):
        uv = _f32(uv).reshape(-1, 2)
        out = np.zeros((uv.shape[0], 3), np.float32)
        self.L.orc_texture_at(self.h, tex, uv.shape[0], _p(uv), _p(out))
        return out

    def render(self, W, H, first_sample, n_samples, max_depth=50, rng_mode=1, order=1, threads=1, row0=0, row1=None,
               accum=None, want_counters=False):
        """Returns (accum HxWx4 float32, rgba8 HxWx4 uint8, counters dict or None)."""
        if row1 is None:
            row1 = H
        if accum is None:
            accum = np.zeros((H, W, 4), np.float32)
        rgba = np.zeros((H, W, 4), np.uint8)
        cnt = OrcCounters() if want_counters else None
        self.L.orc_render(self.h, W, H, first_sample, n_samples, max_depth, rng_mode, order, threads, row0, row1,
                          _p(accum), _p(rgba), C.byref(cnt) if cnt is not None else None)
        return accum, rgba, (cnt.as_dict() if cnt is not None else None)


def hash_random(x, y, sample, counter):
    return lib().orc_hash_random(x, y, sample, counter)


def reflectance(cosine, ref_idx):
    cosine, ref_idx = _f32(cosine), _f32(ref_idx)
    out = np.zeros(cosine.size, np.float32)
    lib().orc_reflectance(cosine.size, _p(cosine), _p(ref_idx), _p(out))
    return out


def texture_lut():
    out = np.zeros(256, np.float32)
    lib().orc_texture_lut(_p(out))
    return out


def write_color(sums3, spp):
    s = _f32(sums3).reshape(-1, 3)
    out = np.zeros((s.shape[0], 4), np.uint8)
    lib().orc_write_color(s.shape[0], _p(s), spp, _p(out))
    return out


def checksum131(buf):
    """s = s*131 + byte over a byte string, modulo 2^64."""
    h = 0
    for b in bytes(buf):
        h = (h * 131 + b) & 0xFFFFFFFFFFFFFFFF
    return h
