"""Scene set-ups of the reference application, restated on the Scene vocabulary.

Each function takes any object with the Scene building methods (the product ``Scene`` or the
test oracle's mirror of it), so both sides of a parity test are built by the same calls.
R/ = /root/reference/Raytracing-In-A-Weekend/.
"""
import numpy as np

from . import translate_scale

COVER_BACKGROUND = (0.7, 0.8, 1.0)


def aspect_ratio(width, height, true_aspect=False):
    """`const float aspectRatio = imageWidth / imageHeight;` is an integer division in the
    reference (R/src/RaytracingApplication.cpp:251,284,336): 1.0 for every 16:9 size."""
    return float(width) / float(height) if true_aspect else float(width // height)


def cover_scene(scene, width, height, wrap_bvh=True, true_aspect=False):
    """Book-1 cover scene: randomScene() (R/src/RaytracingApplication.cpp:359-405) on a fresh
    generator, then BVHNode(world) on the continuing stream (R/src/AccelerationStructures/Bvh.h:17),
    the book's camera (lookfrom (13,2,3), lookat origin, vfov 20, aperture 0.1, focus 10) and
    the flat sky colour the renderer returns on a miss (R/src/Core/Raytracer.cpp:10-11)."""
    scene.random_scene()
    if wrap_bvh:
        scene.wrap_bvh()
    scene.set_camera((13.0, 2.0, 3.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), 20.0,
                     aspect_ratio(width, height, true_aspect), 0.1, 10.0)
    scene.set_background(COVER_BACKGROUND)
    return scene


VASE_MODEL = translate_scale((277.5, 100.0, 277.5), (2000.0, 2000.0, 2000.0))


def cornell_mesh_scene(scene, width, height, mesh, textures, true_aspect=False, max_depth=10):
    """setupWorld()'s CPU branch (R/src/RaytracingApplication.cpp:296-346): cornellBox(), then a
    Mesh with one PBR material placed by T(277.5,100,277.5)*S(2000), camera (278,278,-800) ->
    (278,278,0), vfov 40, aperture 0, focus 10, black background.

    mesh: dict(positions, normals, uvs, indices) in the importer's convention.
    textures: dict(diffuse=HxWx3 uint8, roughness=... or None, normal=... or None)."""
    scene.cornell_box()
    td = scene.add_texture(textures["diffuse"])
    tr = scene.add_texture(textures["roughness"]) if textures.get("roughness") is not None else -1
    tn = scene.add_texture(textures["normal"]) if textures.get("normal") is not None else -1
    mat = scene.add_pbr(td, tr, tn)
    scene.add_mesh(VASE_MODEL, mesh["positions"], mesh["indices"], mat, normals=mesh.get("normals"),
                   uvs=mesh.get("uvs"), max_depth=max_depth)
    scene.set_camera((278.0, 278.0, -800.0), (278.0, 278.0, 0.0), (0.0, 1.0, 0.0), 40.0,
                     aspect_ratio(width, height, true_aspect), 0.0, 10.0)
    scene.set_background((0.0, 0.0, 0.0))
    return scene


def instanced_mesh_scene(scene, mesh, copies, textures, spacing=0.09, scale=500.0):
    """The large-mesh configuration: `copies` instances of one mesh on a jittered square grid over the
    Cornell box's floor, every instance baked into ONE triangle list (the reference has no instancing:
    Mesh::processMesh pre-transforms every vertex, R/src/Core/Mesh.cpp:106-208) and added as one Mesh
    with an unlimited-depth BVH (the reference's depth-10 cap would leave list leaves of thousands of
    triangles).  The grid jitter comes from the scene's own generator, so it is part of the scene's
    random stream like everything else the reference draws.  Returns the triangle count."""
    scene.cornell_box()
    td = scene.add_texture(textures["diffuse"])
    tr = scene.add_texture(textures["roughness"]) if textures.get("roughness") is not None else -1
    mat = scene.add_pbr(td, tr, -1)
    jit = scene.rng_floats(2 * copies)
    side = int(np.ceil(np.sqrt(copies)))
    nv = mesh["positions"].shape[0]
    c = np.arange(copies)
    off = np.zeros((copies, 3), np.float32)
    off[:, 0] = ((c % side).astype(np.float32) + np.float32(0.3) * jit[0::2]) * np.float32(spacing) - np.float32(0.5 * spacing * side)
    off[:, 2] = ((c // side).astype(np.float32) + np.float32(0.3) * jit[1::2]) * np.float32(spacing)
    pos = (mesh["positions"][None, :, :] + off[:, None, :]).reshape(-1, 3)
    idx = (mesh["indices"][None, :, :] + (c * nv).astype(np.uint32)[:, None, None]).reshape(-1, 3)
    nrm = np.tile(mesh["normals"], (copies, 1)) if mesh.get("normals") is not None else None
    uv = np.tile(mesh["uvs"], (copies, 1)) if mesh.get("uvs") is not None else None
    scene.add_mesh(translate_scale((277.5, 100.0, 277.5), (scale, scale, scale)), pos, idx, mat, normals=nrm, uvs=uv,
                   max_depth=-1)
    scene.set_camera((278.0, 400.0, -800.0), (278.0, 150.0, 300.0), (0.0, 1.0, 0.0), 40.0, 1.0, 0.0, 10.0)
    scene.set_background((0.6, 0.7, 0.9))
    return idx.shape[0]


def placed_mesh_scene(scene, mesh, copies, textures, spacing=0.09, scale=500.0, max_depth=-1, wrap_bvh=True):
    """instanced_mesh_scene() with true instances (rt_scene_add_instance; the reference has none): the mesh is
    held ONCE, as a prototype in its own space with its own reference BVH, and `copies` placements on the same
    jittered grid refer to it -- 2,760 stored triangles instead of 2,760 x copies.  The placements go under one
    BVHNode (the reference's own builder over their boxes), the Cornell box beside it as in the baked scene (under
    a BVHNode its floor would get the reference's paper-thin box for a triangle at y = 0, which rays fall through).
    The image is close to the baked scene's but not the same: each baked copy has its own randomly split tree and
    world-space arithmetic."""
    td = scene.add_texture(textures["diffuse"])
    tr = scene.add_texture(textures["roughness"]) if textures.get("roughness") is not None else -1
    mat = scene.add_pbr(td, tr, -1)
    jit = scene.rng_floats(2 * copies)
    side = int(np.ceil(np.sqrt(copies)))
    c = np.arange(copies)
    off = np.zeros((copies, 3), np.float32)
    off[:, 0] = ((c % side).astype(np.float32) + np.float32(0.3) * jit[0::2]) * np.float32(spacing) - np.float32(0.5 * spacing * side)
    off[:, 2] = ((c // side).astype(np.float32) + np.float32(0.3) * jit[1::2]) * np.float32(spacing)
    proto = scene.begin_prototype()
    identity = translate_scale((0.0, 0.0, 0.0), (1.0, 1.0, 1.0))
    scene.add_mesh(identity, mesh["positions"], mesh["indices"], mat, normals=mesh.get("normals"), uvs=mesh.get("uvs"),
                   max_depth=max_depth)
    assert scene.end_prototype() == proto
    for k in range(copies):
        # T(277.5, 100, 277.5) S(scale) T(offset): the matrix the baked scene applies to (position + offset)
        m = translate_scale((277.5 + scale * float(off[k, 0]), 100.0 + scale * float(off[k, 1]), 277.5 + scale * float(off[k, 2])),
                            (scale, scale, scale))
        scene.add_instance(proto, m)
    if wrap_bvh:
        scene.wrap_bvh()
    scene.cornell_box()
    scene.set_camera((278.0, 400.0, -800.0), (278.0, 150.0, 300.0), (0.0, 1.0, 0.0), 40.0, 1.0, 0.0, 10.0)
    scene.set_background((0.6, 0.7, 0.9))
    return mesh["indices"].shape[0]


def lathe_mesh(n_around=24, n_up=20, seed=0):
    """A vase-like surface of revolution in the brass vase's model units (height about 0.17,
    radius up to about 0.05), with smooth normals and texture coordinates: synthetic stand-in
    for the imported mesh (positions Nx3, normals Nx3, uvs Nx2, indices Mx3)."""
    rng = np.random.RandomState(seed)
    ys = np.linspace(1e-4, 0.17, n_up + 1)
    prof = 0.018 + 0.022 * np.sin(np.linspace(0.2, 2.9, n_up + 1)) ** 2 + 0.002 * rng.rand(n_up + 1)
    ang = np.linspace(0, 2 * np.pi, n_around, endpoint=False)
    pos, nrm, uv = [], [], []
    dprof = np.gradient(prof, ys)
    for j in range(n_up + 1):
        for i in range(n_around):
            c, s = np.cos(ang[i]), np.sin(ang[i])
            pos.append((prof[j] * c, ys[j], prof[j] * s))
            n = np.array((c, -dprof[j], s))
            nrm.append(n / np.linalg.norm(n))
            uv.append((0.02 + 0.96 * i / n_around, 0.02 + 0.96 * j / n_up))
    idx = []
    for j in range(n_up):
        for i in range(n_around):
            a = j * n_around + i
            b = j * n_around + (i + 1) % n_around
            c2 = a + n_around
            d = b + n_around
            idx.append((a, c2, b))
            idx.append((b, c2, d))
    return {"positions": np.asarray(pos, np.float32), "normals": np.asarray(nrm, np.float32),
            "uvs": np.asarray(uv, np.float32), "indices": np.asarray(idx, np.uint32)}


def noise_texture(size, seed, octaves=4):
    """Deterministic value-noise RGB8 texture (size x size), a synthetic stand-in for the
    4096^2 JPEG maps of the brass vase."""
    rng = np.random.RandomState(seed)
    out = np.zeros((size, size, 3), np.float32)
    amp_sum = 0.0
    for o in range(octaves):
        cells = min(size, 4 << o)
        grid = rng.rand(cells + 1, cells + 1, 3).astype(np.float32)
        t = np.linspace(0, cells, size, endpoint=False)
        i0 = np.floor(t).astype(int)
        f = (t - i0).astype(np.float32)
        f = f * f * (3 - 2 * f)
        g00 = grid[i0][:, i0]
        g01 = grid[i0][:, i0 + 1]
        g10 = grid[i0 + 1][:, i0]
        g11 = grid[i0 + 1][:, i0 + 1]
        fx = f[None, :, None]
        fy = f[:, None, None]
        layer = (g00 * (1 - fx) + g01 * fx) * (1 - fy) + (g10 * (1 - fx) + g11 * fx) * fy
        amp = 0.5 ** o
        out += amp * layer
        amp_sum += amp
    out /= amp_sum
    return np.clip(out * 255.0 + 0.5, 0, 255).astype(np.uint8)


def load_gltf_textures(path):
    """The three maps a glTF file's first material binds, decoded with the library's own JPEG / PNG
    decoder: dict(diffuse, roughness, normal) of HxWx3 uint8 arrays (None where the file has none),
    keyed the way Mesh::processMesh uses them (R/src/Core/Mesh.cpp:56-104: base colour -> diffuse,
    metallic-roughness -> roughness, normal -> normal)."""
    import json
    import os
    from . import image_decode
    g = json.load(open(path))
    base = os.path.dirname(path)
    mat = g["materials"][0]
    pbr = mat.get("pbrMetallicRoughness", {})

    def tex(ref):
        if not ref:
            return None
        uri = g["images"][g["textures"][ref["index"]]["source"]]["uri"]
        img = image_decode(open(os.path.join(base, uri), "rb").read())
        return np.ascontiguousarray(np.repeat(img, 3, axis=2) if img.shape[2] < 3 else img[:, :, :3])

    return {"diffuse": tex(pbr.get("baseColorTexture")), "roughness": tex(pbr.get("metallicRoughnessTexture")),
            "normal": tex(mat.get("normalTexture"))}


def load_gltf_mesh(path):
    """Positions / normals / flipped texture coordinates / indices of the first primitive of a
    glTF 2.0 file with an external buffer: what the reference's importer hands to
    R/src/Core/Mesh.cpp:106-142 (numpy mirror of rt_scene_add_mesh_gltf's reader)."""
    import json
    import os
    g = json.load(open(path))
    base = os.path.dirname(path)
    blobs = [open(os.path.join(base, b["uri"]), "rb").read() for b in g["buffers"]]
    prim = g["meshes"][0]["primitives"][0]

    def acc(i, ncomp):
        a = g["accessors"][i]
        v = g["bufferViews"][a["bufferView"]]
        off = v.get("byteOffset", 0) + a.get("byteOffset", 0)
        dt = {5126: np.float32, 5123: np.uint16, 5125: np.uint32}[a["componentType"]]
        n = a["count"] * ncomp
        return np.frombuffer(blobs[v["buffer"]], dtype=dt, count=n, offset=off).reshape(a["count"], ncomp)

    at = prim["attributes"]
    out = {"positions": acc(at["POSITION"], 3).astype(np.float32),
           "indices": acc(prim["indices"], 1).astype(np.uint32).reshape(-1, 3)}
    if "NORMAL" in at:
        out["normals"] = acc(at["NORMAL"], 3).astype(np.float32)
    if "TEXCOORD_0" in at:
        uv = acc(at["TEXCOORD_0"], 2).astype(np.float32).copy()
        uv[:, 1] = np.float32(1.0) - uv[:, 1]
        out["uvs"] = uv
    return out
