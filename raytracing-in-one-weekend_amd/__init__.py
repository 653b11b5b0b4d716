"""Python host side of the MI355X path-tracer core.

A thin ctypes binding of ``librtiow_hip.so`` (C ABI: ``include/rtiow_hip.h``) plus a mirror of
the reference renderer's interface: ``Scene`` stands for ``struct Scene``
(R/src/Core/Raytracer.h:17-22) and ``RaytracerHIP`` for the ``Raytracer`` subclasses
(R/src/Core/Raytracer.h:24-90: ctor(image, scene, H, W, spp, maxDepth, buildUp), Run(), Cancel()).
R/ = /root/reference/Raytracing-In-A-Weekend/.

There is no CPU fallback here: if the shared library is missing, importing this package
raises; if no HIP device is present, creating a ``Context`` raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RTIOW_LIB") or os.path.join(_HERE, "librtiow_hip.so")  # RTIOW_LIB: experimental builds
if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} is missing: build it with `make -C {_HERE}/csrc` "
        "(or __graft_entry__.build()); this package has no fallback path"
    )
lib = C.CDLL(LIB_PATH)

MAT_LAMBERTIAN, MAT_METAL, MAT_DIELECTRIC, MAT_LIGHT, MAT_PBR = range(5)
RT_ERR_NO_DEVICE = -2
RT_ERR_CANCELLED = -5


class Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in (
        "samples", "rays", "node_tests", "sphere_tests", "tri_tests", "hits",
        "scatter_lambertian", "scatter_metal", "scatter_dielectric", "scatter_light", "scatter_pbr",
        "misses", "depth_cutoffs", "fallbacks")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class FlatInfo(C.Structure):
    _fields_ = [(n, C.c_int) for n in (
        "n_nodes", "n_prim_refs", "n_spheres", "n_triangles", "n_materials", "n_textures", "max_leaf", "fits_lds")]


PASS_CALLBACK = C.CFUNCTYPE(None, C.c_int, C.c_void_p)

# name -> (restype, argtypes); every symbol include/rtiow_hip.h declares
_VP, _I, _F = C.c_void_p, C.c_int, C.c_float
SYMBOLS = {
    "rt_last_error": (C.c_char_p, []),
    "rt_version": (C.c_char_p, []),
    "rt_scene_create": (_I, [C.POINTER(_VP)]),
    "rt_scene_destroy": (None, [_VP]),
    "rt_scene_rng_floats": (_I, [_VP, _I, _VP]),
    "rt_scene_add_material": (_I, [_VP, _I, _F, _F, _F, _F]),
    "rt_scene_add_texture_rgb8": (_I, [_VP, _I, _I, _VP]),
    "rt_texture_lut256": (_I, [_VP]),
    "rt_scene_add_texture_file": (_I, [_VP, C.c_char_p]),
    "rt_image_decode": (_I, [_VP, C.c_size_t, C.POINTER(_I), C.POINTER(_I), C.POINTER(_I), _VP, C.c_size_t]),
    "rt_scene_add_material_pbr": (_I, [_VP, _I, _I, _I]),
    "rt_scene_add_sphere": (_I, [_VP, _F, _F, _F, _F, _I]),
    "rt_scene_add_triangle": (_I, [_VP, _VP, _VP, _VP, _I]),
    "rt_scene_add_box": (_I, [_VP, _VP, _VP, _I, _VP]),
    "rt_scene_add_mesh": (_I, [_VP, _VP, _I, _VP, _VP, _VP, _I, _VP, _I, _I]),
    "rt_scene_add_mesh_gltf": (_I, [_VP, _VP, C.c_char_p, _I, _I]),
    "rt_scene_add_mesh_file": (_I, [_VP, _VP, C.c_char_p, _I, _I, _I]),
    "rt_scene_begin_bvh_node": (_I, [_VP, _VP]),
    "rt_scene_begin_mesh": (_I, [_VP, _VP]),
    "rt_scene_begin_list": (_I, [_VP]),
    "rt_scene_end_group": (_I, [_VP]),
    "rt_scene_begin_prototype": (_I, [_VP]),
    "rt_scene_end_prototype": (_I, [_VP]),
    "rt_scene_add_instance": (_I, [_VP, _I, _VP]),
    "rt_scene_random_scene": (_I, [_VP]),
    "rt_scene_cornell_box": (_I, [_VP]),
    "rt_scene_wrap_bvh": (_I, [_VP]),
    "rt_scene_set_camera": (_I, [_VP, _VP, _VP, _VP, _F, _F, _F, _F]),
    "rt_scene_get_camera": (_I, [_VP, _VP]),
    "rt_scene_set_camera_members": (_I, [_VP, _VP]),
    "rt_scene_set_background": (_I, [_VP, _F, _F, _F]),
    "rt_scene_set_normal_mapping": (_I, [_VP, _I]),
    "rt_scene_flatten": (_I, [_VP, C.POINTER(FlatInfo)]),
    "rt_scene_get_flat": (_I, [_VP] + [_VP] * 10),
    "rt_scene_get_search_structure": (_I, [_VP, C.POINTER(_I), _VP, C.POINTER(_I), _VP]),
    "rt_scene_get_search_structure_octants": (_I, [_VP, C.POINTER(_I), _VP]),
    "rt_device_count": (_I, []),
    "rt_create": (_I, [_I, C.POINTER(_VP)]),
    "rt_destroy": (None, [_VP]),
    "rt_set_stream": (_I, [_VP, _VP]),
    "rt_set_reference_traversal": (_I, [_VP, _I]),
    "rt_set_device_build": (_I, [_VP, _I]),
    "rt_last_build_ms": (_I, [_VP, C.POINTER(C.c_float)]),
    "rt_read_search_structure": (_I, [_VP, C.POINTER(_I), _VP, C.POINTER(_I), _VP]),
    "rt_upload_scene": (_I, [_VP, _VP]),
    "rt_set_target": (_I, [_VP, _I, _I, _I, _I]),
    "rt_set_target_external": (_I, [_VP, _I, _I, _I, _I, _VP]),
    "rt_accum_bytes": (C.c_size_t, [_I, _I, _I]),
    "rt_release_target": (_I, [_VP]),
    "rt_clear": (_I, [_VP]),
    "rt_render": (_I, [_VP, _I, _I, _I]),
    "rt_render_noalloc": (_I, [_VP, _I, _I, _I]),
    "rt_noalloc_max_pixels": (C.c_size_t, []),
    "rt_sync": (_I, [_VP]),
    "rt_cancel": (_I, [_VP]),
    "rt_run": (_I, [_VP, _I, _I, _I, PASS_CALLBACK, _VP]),
    "rt_read_accum_rgba32f": (_I, [_VP, _VP]),
    "rt_read_rgba8": (_I, [_VP, _I, _VP]),
    "rt_write_ppm": (_I, [C.c_char_p, _I, _I, _VP]),
    "rt_write_png": (_I, [C.c_char_p, _I, _I, _VP]),
    "rt_resolve_gathered": (_I, [_VP, _VP, _I, _VP]),
    "rt_resolve_gathered_rgba8": (_I, [_VP, _VP, _I, _I, _VP]),
    "rt_group_create": (_I, [_I, _VP, C.POINTER(_VP)]),
    "rt_group_destroy": (None, [_VP]),
    "rt_group_size": (_I, [_VP]),
    "rt_group_context": (_VP, [_VP, _I]),
    "rt_group_upload_scene": (_I, [_VP, _VP]),
    "rt_group_set_target": (_I, [_VP, _I, _I]),
    "rt_group_clear": (_I, [_VP]),
    "rt_group_render": (_I, [_VP, _I, _I, _I]),
    "rt_group_read_rgba8": (_I, [_VP, _I, _VP]),
    "rt_group_read_accum_rgba32f": (_I, [_VP, _VP]),
    "rt_group_last_error": (C.c_char_p, []),
    "rt_trace": (_I, [_VP, _I] + [_VP] * 7),
    "rt_scatter_test": (_I, [_VP, _I] + [_VP] * 11),
    "rt_reflectance_test": (_I, [_VP, _I, _VP, _VP, _VP]),
    "rt_count": (_I, [_VP, _I, _I, _I, C.POINTER(Counters)]),
    "rt_sched_stats": (_I, [_VP, _VP]),
    "rt_last_render_ms": (_I, [_VP, C.POINTER(C.c_float)]),
    "rt_stream": (_VP, [_VP]),
}
for _name, (_res, _args) in SYMBOLS.items():
    _fn = getattr(lib, _name)
    _fn.restype = _res
    _fn.argtypes = _args


class RtError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"rtiow_hip error {code}: {msg}")
        self.code = code


def _check(rc):
    if rc < 0:
        raise RtError(rc, lib.rt_last_error().decode())
    return rc


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f32(a, n=None):
    a = np.ascontiguousarray(a, dtype=np.float32)
    if n is not None and a.size != n:
        raise ValueError(f"expected {n} floats, got {a.size}")
    return a


IDENTITY = np.eye(4, dtype=np.float32)


def translate_scale(t, s):
    """glm::scale(glm::translate(mat4(1), t), s) as a column-major 4x4 (flattened: m[col*4+row])."""
    m = np.zeros((4, 4), np.float32)  # m[col][row]
    m[0][0], m[1][1], m[2][2] = np.float32(s[0]), np.float32(s[1]), np.float32(s[2])
    m[3][:3] = np.asarray(t, np.float32)
    m[3][3] = 1.0
    return m


class Scene:
    """Host-side scene: materials, textures, primitives, camera, background (struct Scene)."""

    def __init__(self):
        h = C.c_void_p()
        _check(lib.rt_scene_create(C.byref(h)))
        self.h = h

    def __del__(self):
        if getattr(self, "h", None):
            lib.rt_scene_destroy(self.h)
            self.h = None

    def rng_floats(self, n):
        out = np.zeros(n, np.float32)
        _check(lib.rt_scene_rng_floats(self.h, n, _p(out)))
        return out

    def add_material(self, kind, rgb=(0, 0, 0), param=0.0):
        return _check(lib.rt_scene_add_material(self.h, kind, rgb[0], rgb[1], rgb[2], param))

    def add_texture(self, rgb8):
        a = np.ascontiguousarray(rgb8, dtype=np.uint8)
        assert a.ndim == 3 and a.shape[2] == 3
        return _check(lib.rt_scene_add_texture_rgb8(self.h, a.shape[1], a.shape[0], _p(a)))

    def add_texture_file(self, path):
        return _check(lib.rt_scene_add_texture_file(self.h, os.fsencode(path)))

    def add_pbr(self, tex_diffuse, tex_roughness=-1, tex_normal=-1):
        return _check(lib.rt_scene_add_material_pbr(self.h, tex_diffuse, tex_roughness, tex_normal))

    def add_sphere(self, center, radius, material):
        return _check(lib.rt_scene_add_sphere(self.h, center[0], center[1], center[2], radius, material))

    def add_triangle(self, pos9, material, normal9=None, uv6=None):
        pos = _f32(pos9, 9)
        nrm = _f32(np.zeros(9) if normal9 is None else normal9, 9)
        uv = _f32(np.zeros(6) if uv6 is None else uv6, 6)
        return _check(lib.rt_scene_add_triangle(self.h, _p(pos), _p(nrm), _p(uv), material))

    def add_box(self, p0, p1, material, model=IDENTITY):
        a, b, m = _f32(p0, 3), _f32(p1, 3), _f32(model, 16)
        return _check(lib.rt_scene_add_box(self.h, _p(a), _p(b), material, _p(m)))

    def add_mesh(self, model, positions, indices, material, normals=None, uvs=None, max_depth=10):
        m = _f32(model, 16)
        pos = _f32(positions)
        idx = np.ascontiguousarray(indices, dtype=np.uint32)
        nrm = None if normals is None else _f32(normals)
        uv = None if uvs is None else _f32(uvs)
        return _check(lib.rt_scene_add_mesh(self.h, _p(m), pos.size // 3, _p(pos), _p(nrm), _p(uv), idx.size // 3,
                                            _p(idx), material, max_depth))

    def add_mesh_gltf(self, model, path, material, max_depth=10):
        m = _f32(model, 16)
        return _check(lib.rt_scene_add_mesh_gltf(self.h, _p(m), path.encode(), material, max_depth))

    def add_mesh_file(self, model, path, max_depth=10, material=-1, per_primitive_materials=False):
        """Mesh(model, path) for a glTF file, textures included (rt_scene_add_mesh_file)."""
        m = _f32(model, 16)
        return _check(lib.rt_scene_add_mesh_file(self.h, _p(m), os.fsencode(path), max_depth, material,
                                                 1 if per_primitive_materials else 0))

    def begin_bvh_node(self, box6):
        b = _f32(box6, 6)
        return _check(lib.rt_scene_begin_bvh_node(self.h, _p(b)))

    def begin_mesh(self, box6):
        b = _f32(box6, 6)
        return _check(lib.rt_scene_begin_mesh(self.h, _p(b)))

    def begin_list(self):
        return _check(lib.rt_scene_begin_list(self.h))

    def end_group(self):
        _check(lib.rt_scene_end_group(self.h))

    def begin_prototype(self):
        """Objects added until end_prototype() form an object in its own space, placed by add_instance()."""
        return _check(lib.rt_scene_begin_prototype(self.h))

    def end_prototype(self):
        return _check(lib.rt_scene_end_prototype(self.h))

    def add_instance(self, prototype, model):
        m = _f32(model, 16)
        return _check(lib.rt_scene_add_instance(self.h, prototype, _p(m)))

    def random_scene(self):
        _check(lib.rt_scene_random_scene(self.h))

    def cornell_box(self):
        _check(lib.rt_scene_cornell_box(self.h))

    def wrap_bvh(self):
        _check(lib.rt_scene_wrap_bvh(self.h))

    def set_camera(self, lookfrom, lookat, vup, vfov, aspect, aperture, focus_dist):
        a, b, c = _f32(lookfrom, 3), _f32(lookat, 3), _f32(vup, 3)
        _check(lib.rt_scene_set_camera(self.h, _p(a), _p(b), _p(c), vfov, aspect, aperture, focus_dist))

    def get_camera(self):
        out = np.zeros(22, np.float32)
        _check(lib.rt_scene_get_camera(self.h, _p(out)))
        return out

    def set_camera_members(self, members22):
        """The eight derived Camera members as they are (origin, lowerLeftCorner, horizontal, vertical, u, v, w,
        lensRadius: 22 floats), bit for bit."""
        m = _f32(members22, 22)
        _check(lib.rt_scene_set_camera_members(self.h, _p(m)))

    def set_background(self, rgb):
        _check(lib.rt_scene_set_background(self.h, rgb[0], rgb[1], rgb[2]))

    def set_normal_mapping(self, on):
        _check(lib.rt_scene_set_normal_mapping(self.h, 1 if on else 0))

    def flatten(self):
        info = FlatInfo()
        _check(lib.rt_scene_flatten(self.h, C.byref(info)))
        return info

    def get_flat(self):
        info = self.flatten()
        d = {
            "nodes": np.zeros((info.n_nodes, 8), np.uint32),
            "prim_refs": np.zeros(info.n_prim_refs, np.uint32),
            "spheres": np.zeros((info.n_spheres, 4), np.float32),
            "sphere_mat": np.zeros(info.n_spheres, np.int32),
            "tri_pos": np.zeros((info.n_triangles, 9), np.float32),
            "tri_normal": np.zeros((info.n_triangles, 9), np.float32),
            "tri_uv": np.zeros((info.n_triangles, 6), np.float32),
            "tri_mat": np.zeros(info.n_triangles, np.int32),
            "mat_kind": np.zeros(info.n_materials, np.int32),
            "mat_rgbp": np.zeros((info.n_materials, 4), np.float32),
        }
        _check(lib.rt_scene_get_flat(self.h, *[_p(d[k]) for k in (
            "nodes", "prim_refs", "spheres", "sphere_mat", "tri_pos", "tri_normal", "tri_uv", "tri_mat", "mat_kind",
            "mat_rgbp")]))
        d["info"] = info
        return d

    def search_structure(self):
        """The flattened scene's search structure: (entries [n, 8] uint32, refs [m] uint32)."""
        self.flatten()
        ne, nr = C.c_int(0), C.c_int(0)
        _check(lib.rt_scene_get_search_structure(self.h, C.byref(ne), None, C.byref(nr), None))
        entries = np.zeros((ne.value, 8), np.uint32)
        refs = np.zeros(nr.value, np.uint32)
        _check(lib.rt_scene_get_search_structure(self.h, None, _p(entries), None, _p(refs)))
        return entries, refs

    def search_structure_octants(self):
        """The eight direction-ordered copies: (stride, entries [8 * stride, 8] uint32), or (0, None)."""
        self.flatten()
        stride = C.c_int(0)
        _check(lib.rt_scene_get_search_structure_octants(self.h, C.byref(stride), None))
        if stride.value == 0:
            return 0, None
        entries = np.zeros((8 * stride.value, 8), np.uint32)
        _check(lib.rt_scene_get_search_structure_octants(self.h, None, _p(entries)))
        return stride.value, entries


def texture_lut():
    """The byte -> float map applied to texture channels (stbi_loadf's)."""
    out = np.zeros(256, np.float32)
    _check(lib.rt_texture_lut256(_p(out)))
    return out


def image_decode(file_bytes):
    """Decodes a JPEG (sequential or progressive) or PNG (interlaced or not): HxWxC uint8, C as stored in the file."""
    buf = np.frombuffer(bytes(file_bytes), np.uint8)
    w, h, c = C.c_int(), C.c_int(), C.c_int()
    _check(lib.rt_image_decode(_p(buf), buf.size, C.byref(w), C.byref(h), C.byref(c), None, 0))
    out = np.zeros((h.value, w.value, c.value), np.uint8)
    _check(lib.rt_image_decode(_p(buf), buf.size, C.byref(w), C.byref(h), C.byref(c), _p(out), out.size))
    return out


def device_count():
    return lib.rt_device_count()


class Context:
    """One device + one stream + the scene and accumulation buffers on it."""

    def __init__(self, device=0):
        h = C.c_void_p()
        _check(lib.rt_create(device, C.byref(h)))
        self.h = h
        self.width = self.height = 0
        self.world = 1

    def close(self):
        if getattr(self, "h", None):
            lib.rt_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def set_stream(self, hip_stream):
        _check(lib.rt_set_stream(self.h, C.c_void_p(hip_stream)))

    def set_reference_traversal(self, on):
        _check(lib.rt_set_reference_traversal(self.h, 1 if on else 0))

    def set_device_build(self, on):
        """True / False: always / never build the search structure on the device; None: for large scenes (the default)."""
        _check(lib.rt_set_device_build(self.h, -1 if on is None else (1 if on else 0)))

    def last_build_ms(self):
        ms = C.c_float()
        _check(lib.rt_last_build_ms(self.h, C.byref(ms)))
        return ms.value

    def read_search_structure(self):
        """(entries [n, 8] uint32, refs [m] uint32) as the context's kernels walk them, from the device."""
        ne, nr = C.c_int(0), C.c_int(0)
        _check(lib.rt_read_search_structure(self.h, C.byref(ne), None, C.byref(nr), None))
        entries = np.zeros((ne.value, 8), np.uint32)
        refs = np.zeros(nr.value, np.uint32)
        _check(lib.rt_read_search_structure(self.h, None, _p(entries), None, _p(refs)))
        return entries, refs

    def upload_scene(self, scene):
        _check(lib.rt_upload_scene(self.h, scene.h))

    def set_target(self, width, height, rank=0, world=1, external_ptr=None):
        if external_ptr is None:
            _check(lib.rt_set_target(self.h, width, height, rank, world))
        else:
            _check(lib.rt_set_target_external(self.h, width, height, rank, world, C.c_void_p(external_ptr)))
        self.width, self.height, self.world = width, height, world

    def release_target(self):
        _check(lib.rt_release_target(self.h))
        self.width = self.height = 0

    def clear(self):
        _check(lib.rt_clear(self.h))

    def render(self, first_sample, n_samples, max_depth=50):
        _check(lib.rt_render(self.h, first_sample, n_samples, max_depth))

    def render_noalloc(self, first_sample, n_samples, max_depth=50):
        _check(lib.rt_render_noalloc(self.h, first_sample, n_samples, max_depth))

    def sync(self):
        _check(lib.rt_sync(self.h))

    def cancel(self):
        _check(lib.rt_cancel(self.h))

    def run(self, samples_per_pixel, samples_per_pass, max_depth=50, on_pass=None):
        cb = PASS_CALLBACK(lambda done, user: on_pass(done)) if on_pass else PASS_CALLBACK()
        _check(lib.rt_run(self.h, samples_per_pixel, samples_per_pass, max_depth, cb, None))

    def read_accum(self):
        out = np.zeros((self.height, self.width, 4), np.float32)
        _check(lib.rt_read_accum_rgba32f(self.h, _p(out)))
        return out

    def read_rgba8(self, samples_per_pixel):
        out = np.zeros((self.height, self.width, 4), np.uint8)
        _check(lib.rt_read_rgba8(self.h, samples_per_pixel, _p(out)))
        return out

    def read_rgba8_into(self, samples_per_pixel, host_ptr):
        """rt_read_rgba8 into caller memory of W*H*4 bytes (e.g. a pinned staging buffer)."""
        _check(lib.rt_read_rgba8(self.h, samples_per_pixel, C.c_void_p(host_ptr)))

    def resolve_gathered(self, payload_ptr, world, out_ptr):
        _check(lib.rt_resolve_gathered(self.h, C.c_void_p(payload_ptr), world, C.c_void_p(out_ptr)))

    def resolve_gathered_rgba8(self, payload_ptr, world, samples_per_pixel, out_ptr):
        _check(lib.rt_resolve_gathered_rgba8(self.h, C.c_void_p(payload_ptr), world, samples_per_pixel,
                                             C.c_void_p(out_ptr)))

    def trace(self, rays):
        rays = _f32(rays).reshape(-1, 6)
        n = rays.shape[0]
        kind = np.zeros(n, np.int32)
        pid = np.zeros(n, np.int32)
        t = np.zeros(n, np.float32)
        nrm = np.zeros((n, 3), np.float32)
        front = np.zeros(n, np.int32)
        uv = np.zeros((n, 2), np.float32)
        _check(lib.rt_trace(self.h, n, _p(rays), _p(kind), _p(pid), _p(t), _p(nrm), _p(front), _p(uv)))
        return kind, pid, t, nrm, front, uv

    def scatter_test(self, material, ray_in, p, normal, front, uv, key4):
        material = np.ascontiguousarray(material, np.int32)
        n = material.size
        ray_in, p, normal, uv = _f32(ray_in, 6 * n), _f32(p, 3 * n), _f32(normal, 3 * n), _f32(uv, 2 * n)
        front = np.ascontiguousarray(front, np.int32)
        key4 = np.ascontiguousarray(key4, np.uint32)
        atten = np.zeros((n, 3), np.float32)
        out = np.zeros((n, 6), np.float32)
        alive = np.zeros(n, np.int32)
        cnt = np.zeros(n, np.uint32)
        _check(lib.rt_scatter_test(self.h, n, _p(material), _p(ray_in), _p(p), _p(normal), _p(front), _p(uv),
                                   _p(key4), _p(atten), _p(out), _p(alive), _p(cnt)))
        return atten, out, alive, cnt

    def reflectance_test(self, cosine, ref_idx):
        cosine, ref_idx = _f32(cosine), _f32(ref_idx)
        if cosine.size != ref_idx.size:
            raise ValueError("cosine and ref_idx must have the same length")
        out = np.zeros(cosine.size, np.float32)
        _check(lib.rt_reflectance_test(self.h, cosine.size, _p(cosine), _p(ref_idx), _p(out)))
        return out

    def count(self, first_sample, n_samples, max_depth=50):
        c = Counters()
        _check(lib.rt_count(self.h, first_sample, n_samples, max_depth, C.byref(c)))
        return c.as_dict()

    def sched_stats(self):
        out = np.zeros(14, np.uint64)
        _check(lib.rt_sched_stats(self.h, _p(out)))
        names = ("box_votes", "prim", "shade", "newpath", "first_pass_steps", "box_steps", "refbox_votes")
        return {n: (int(out[2 * i]), int(out[2 * i + 1])) for i, n in enumerate(names)}

    def last_render_ms(self):
        ms = C.c_float()
        _check(lib.rt_last_render_ms(self.h, C.byref(ms)))
        return ms.value

    @property
    def stream(self):
        return lib.rt_stream(self.h)


class Group:
    """Several GPUs of one node from one process (rt_group_*): tiles sharded over the devices, one RCCL
    all-gather per frame, resolve on device 0."""

    def __init__(self, devices):
        devs = np.ascontiguousarray(devices, dtype=np.int32)
        h = C.c_void_p()
        self._check(lib.rt_group_create(devs.size, _p(devs), C.byref(h)))
        self.h = h
        self.width = self.height = 0

    @staticmethod
    def _check(rc):
        if rc < 0:
            raise RtError(rc, lib.rt_group_last_error().decode())
        return rc

    def close(self):
        if getattr(self, "h", None):
            lib.rt_group_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def upload_scene(self, scene):
        self._check(lib.rt_group_upload_scene(self.h, scene.h))

    def set_target(self, width, height):
        self._check(lib.rt_group_set_target(self.h, width, height))
        self.width, self.height = width, height

    def clear(self):
        self._check(lib.rt_group_clear(self.h))

    def render(self, first_sample, n_samples, max_depth=50):
        self._check(lib.rt_group_render(self.h, first_sample, n_samples, max_depth))

    def read_rgba8(self, samples_per_pixel):
        out = np.zeros((self.height, self.width, 4), np.uint8)
        self._check(lib.rt_group_read_rgba8(self.h, samples_per_pixel, _p(out)))
        return out

    def read_accum(self):
        out = np.zeros((self.height, self.width, 4), np.float32)
        self._check(lib.rt_group_read_accum_rgba32f(self.h, _p(out)))
        return out


def write_ppm(path, rgba8):
    """Plain PPM (P3, top row first) of an HxWx4 uint8 image whose row 0 is the bottom row."""
    a = np.ascontiguousarray(rgba8, dtype=np.uint8)
    if a.ndim != 3 or a.shape[2] != 4:
        raise ValueError("expected an HxWx4 uint8 image")
    _check(lib.rt_write_ppm(os.fsencode(path), a.shape[1], a.shape[0], _p(a)))


def write_png(path, rgba8):
    """8-bit RGB PNG (top row first, uncompressed) of an HxWx4 uint8 image whose row 0 is the bottom row."""
    a = np.ascontiguousarray(rgba8, dtype=np.uint8)
    if a.ndim != 3 or a.shape[2] != 4:
        raise ValueError("expected an HxWx4 uint8 image")
    _check(lib.rt_write_png(os.fsencode(path), a.shape[1], a.shape[0], _p(a)))


def accum_bytes(width, height, world=1):
    return lib.rt_accum_bytes(width, height, world)


class RaytracerHIP:
    """Mirror of the reference's ``Raytracer`` interface (R/src/Core/Raytracer.h:24-49).

    ``image`` is a caller-owned ``uint8`` array of ``H*W*4`` bytes (RGBA8, row 0 = bottom),
    the buffer the application uploads with glTexImage2D (R/src/RaytracingApplication.cpp:194).
    ``Run()`` fills it progressively: after every pass when ``build_up`` is true (the build-up
    mode of Raytracer.cpp:48-73, with the correct 1/passes scale instead of the reference's
    off-by-one divisor), once at the end otherwise.  ``Cancel()`` may be called from another thread.
    """

    def __init__(self, image, scene, image_height, image_width, samples_per_pixel, max_depth, build_up,
                 device=0, samples_per_pass=1):
        if image.dtype != np.uint8 or image.size != image_height * image_width * 4:
            raise ValueError("image must be uint8 with H*W*4 elements")
        self.image = image.reshape(image_height, image_width, 4)
        self.spp, self.max_depth, self.build_up = samples_per_pixel, max_depth, build_up
        self.samples_per_pass = samples_per_pass if build_up else max(samples_per_pixel, 1)
        self.ctx = Context(device)
        self.ctx.upload_scene(scene)
        self.ctx.set_target(image_width, image_height)
        self.cancelled = False

    def _on_pass(self, done):
        if self.build_up or done >= self.spp:
            self.image[...] = self.ctx.read_rgba8(done)

    def Run(self):
        try:
            self.ctx.run(self.spp, self.samples_per_pass, self.max_depth, self._on_pass)
        except RtError as e:
            if e.code != RT_ERR_CANCELLED:
                raise
            self.cancelled = True

    def Cancel(self):
        self.ctx.cancel()
