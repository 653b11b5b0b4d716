"""The product's host code (scene build, reference-order BVH, flatten step) against the oracle.

No GPU needed: these compare the packed arrays the library would upload with the oracle's
pointer-tree walk, bit for bit.
"""
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
G = os.path.join(HERE, "golden")
NOBOX = 1 << 30
LAST, TRI = 1 << 31, 1 << 30


def flat_views(flat):
    nodes = flat["nodes"]
    has_box = (nodes[:, 3] & NOBOX) == 0
    skip = (nodes[:, 3] & (NOBOX - 1)).astype(np.int64)
    first = nodes[:, 7].view(np.int32)
    boxes = np.concatenate([nodes[:, 0:3], nodes[:, 4:7]], axis=1)
    return has_box, skip, first, boxes


def check_structure(flat, orc_boxes, orc_desc, orc_prims):
    """Entries with a box test, in order, are the oracle's box-tested nodes in visit order; a
    miss skips exactly the node's subtree; leaf references are the oracle's primitive visit order."""
    has_box, skip, first, boxes = flat_views(flat)
    n = len(skip)
    assert np.array_equal(boxes[has_box], orc_boxes.view(np.uint32))
    refs = flat["prim_refs"]
    kinds = ((refs & TRI) != 0).astype(np.int32)
    ids = (refs & (TRI - 1)).astype(np.int32)
    assert np.array_equal(np.stack([kinds, ids], axis=1), orc_prims)
    # skip links: entry e's subtree holds desc[k] box-tested entries
    box_idx = np.nonzero(has_box)[0]
    assert len(box_idx) == len(orc_desc)
    for k, e in enumerate(box_idx):
        inside = np.count_nonzero(has_box[e + 1:skip[e]])
        assert inside == orc_desc[k]
        assert e < skip[e] <= n
    # entries without a box never skip anything
    for e in np.nonzero(~has_box)[0]:
        assert skip[e] == e + 1 and first[e] >= 0
    # every leaf run is terminated, runs are visited in array order, and all references are used
    pos = 0
    for e in range(n):
        if first[e] >= 0:
            assert first[e] == pos
            while not (refs[pos] & LAST):
                pos += 1
            pos += 1
    assert pos == len(refs)


def build_both(rt, orc, builder):
    a, b = rt.Scene(), orc.OracleScene()
    builder(a)
    builder(b)
    return a, b


def compare_scene_arrays(flat, od):
    assert np.array_equal(flat["spheres"].view(np.uint32), od["spheres"].view(np.uint32))
    assert np.array_equal(flat["sphere_mat"], od["sphere_mat"])
    assert np.array_equal(flat["mat_kind"], od["mat_kind"])
    assert np.array_equal(flat["mat_rgbp"].view(np.uint32), od["mat_rgbp"].view(np.uint32))
    if od["tris"].shape[0]:
        assert np.array_equal(flat["tri_pos"].view(np.uint32), od["tris"][:, 0:9].view(np.uint32))
        assert np.array_equal(flat["tri_normal"].view(np.uint32), od["tris"][:, 9:18].view(np.uint32))
        assert np.array_equal(flat["tri_uv"].view(np.uint32), od["tris"][:, 18:24].view(np.uint32))
        assert np.array_equal(flat["tri_mat"], od["tri_mat"])


def test_rng_stream(rt, orc):
    a, b = rt.Scene(), orc.OracleScene()
    assert np.array_equal(a.rng_floats(1000).view(np.uint32), b.rng_floats(1000).view(np.uint32))


@pytest.mark.parametrize("wrap", [False, True])
def test_cover_scene_flatten(rt, orc, scenes, wrap):
    a, b = build_both(rt, orc, lambda s: scenes.cover_scene(s, 1920, 1080, wrap_bvh=wrap))
    flat = a.get_flat()
    compare_scene_arrays(flat, b.dump())
    check_structure(flat, *b.dump_tree())
    assert np.array_equal(a.get_camera().view(np.uint32), b.get_camera().view(np.uint32))
    info = flat["info"]
    assert info.n_spheres == 488 and info.n_prim_refs == 488 and info.fits_lds == 1
    if wrap:
        assert info.n_nodes == 511 and info.max_leaf == 2
        g = np.load(os.path.join(G, "cover_scene.npz"))
        has_box, _, _, boxes = flat_views(flat)
        assert np.array_equal(boxes[has_box], g["boxes"])
    else:
        assert info.n_nodes == 1 and info.max_leaf == 488
    # the generator continues identically after the build on both sides
    assert np.array_equal(a.rng_floats(16).view(np.uint32), b.rng_floats(16).view(np.uint32))


def test_cornell_and_gltf_mesh_flatten(rt, orc, scenes):
    path = os.path.join(G, "brass_vase", "brass_vase_04_4k.gltf")
    mesh = scenes.load_gltf_mesh(path)
    tex = {"diffuse": scenes.noise_texture(16, 1), "roughness": scenes.noise_texture(16, 2), "normal": None}
    a, b = build_both(rt, orc, lambda s: scenes.cornell_mesh_scene(s, 640, 360, mesh, tex))
    flat = a.get_flat()
    compare_scene_arrays(flat, b.dump())
    check_structure(flat, *b.dump_tree())
    assert flat["info"].n_triangles == 12 + 2760
    # the library's own glTF reader gives the same mesh as the numpy reader
    c = rt.Scene()
    c.cornell_box()
    td, tr = c.add_texture(tex["diffuse"]), c.add_texture(tex["roughness"])
    c.add_mesh_gltf(scenes.VASE_MODEL, path, c.add_pbr(td, tr, -1), max_depth=10)
    flat_c = c.get_flat()
    for k in ("nodes", "prim_refs", "tri_pos", "tri_normal", "tri_uv", "tri_mat"):
        assert np.array_equal(flat_c[k], flat[k]), k


@pytest.mark.parametrize("max_depth", [0, 1, 3, -1])
def test_mesh_bvh_depth_limits(rt, orc, scenes, max_depth):
    mesh = scenes.lathe_mesh(10, 7, seed=3)
    tex = {"diffuse": scenes.noise_texture(8, 5), "roughness": None, "normal": None}
    a, b = build_both(rt, orc, lambda s: scenes.cornell_mesh_scene(s, 64, 64, mesh, tex, max_depth=max_depth))
    flat = a.get_flat()
    compare_scene_arrays(flat, b.dump())
    check_structure(flat, *b.dump_tree())


def test_mixed_world_wrapped_in_bvh(rt, orc, scenes):
    """Spheres, loose triangles and a mesh under one top-level BVH: leaves that are composites."""
    mesh = scenes.lathe_mesh(6, 4, seed=9)
    r = np.random.RandomState(5)

    def build(s):
        m = [s.add_material(0, (0.5, 0.4, 0.3)), s.add_material(1, (0.9, 0.8, 0.7), 0.2), s.add_material(2, (0, 0, 0), 1.5)]
        rr = np.random.RandomState(5)
        for i in range(23):
            c = (rr.rand(3) * 400 + 50).astype(np.float32)
            s.add_sphere(c, float(np.float32(5 + 20 * rr.rand())), m[i % 3])
        for i in range(7):
            p = (rr.rand(9) * 500).astype(np.float32)
            s.add_triangle(p, m[i % 2])
        td = s.add_texture(scenes.noise_texture(8, 1))
        s.add_mesh(scenes.VASE_MODEL, mesh["positions"], mesh["indices"], s.add_pbr(td, -1, -1),
                   normals=mesh["normals"], uvs=mesh["uvs"], max_depth=2)
        s.add_sphere((10.0, 10.0, 10.0), 3.0, m[0])
        s.wrap_bvh()

    del r
    a, b = build_both(rt, orc, build)
    flat = a.get_flat()
    compare_scene_arrays(flat, b.dump())
    check_structure(flat, *b.dump_tree())


def test_duplicate_box_minima_sort_like_the_oracle(rt, orc):
    """Every small cover-scene sphere has the same minimum y, so an axis-1 split is decided by
    std::sort's handling of ties; exercise that with many equal keys."""
    def build(s):
        m = s.add_material(0, (0.5, 0.5, 0.5))
        rr = np.random.RandomState(11)
        for i in range(200):
            x, z = rr.rand(2) * 20
            s.add_sphere((float(np.float32(x)), 0.2, float(np.float32(z))), 0.2, m)
        s.wrap_bvh()

    a, b = build_both(rt, orc, build)
    check_structure(a.get_flat(), *b.dump_tree())


def test_box_helper_emits_twelve_triangles(rt):
    s = rt.Scene()
    m = s.add_material(0, (0.73, 0.73, 0.73))
    first = s.add_box((0, 0, 0), (165, 330, 165), m, rt.translate_scale((265, 0, 295), (1, 1, 1)))
    flat = s.get_flat()
    assert first == 0 and flat["info"].n_triangles == 12 and flat["info"].n_nodes == 1
    pos = flat["tri_pos"].reshape(12, 3, 3)
    assert pos[..., 0].min() == 265 and pos[..., 0].max() == 430
    assert pos[..., 1].min() == 0 and pos[..., 1].max() == 330
    assert (flat["tri_normal"] == 0).all()
    # first side of the reference's Box: (p0), (p0.x,p0.y,p1.z), (p0.x,p1.y,p0.z)
    assert pos[0].tolist() == [[265, 0, 295], [265, 0, 460], [265, 330, 295]]


def test_error_paths(rt):
    s = rt.Scene()
    with pytest.raises(rt.RtError):
        s.add_sphere((0, 0, 0), 1.0, 0)          # no such material
    with pytest.raises(rt.RtError):
        s.wrap_bvh()                              # empty world
    with pytest.raises(rt.RtError):
        s.add_pbr(0)                              # no such texture
    m = s.add_material(0, (1, 1, 1))
    with pytest.raises(rt.RtError):
        s.add_mesh(rt.IDENTITY, np.zeros((3, 3), np.float32), np.array([[0, 1, 7]], np.uint32), m)
    with pytest.raises(rt.RtError):
        s.add_mesh_gltf(rt.IDENTITY, "/nonexistent/file.gltf", m)
    assert rt.lib.rt_scene_flatten(None, None) < 0
    assert b"null" in rt.lib.rt_last_error()


def test_empty_world_flattens_to_one_boxless_entry(rt):
    """HittableList::Hit over no objects misses every ray; the flattened form is a single entry
    without a box test and without a leaf run."""
    s = rt.Scene()
    d = s.get_flat()
    assert d["info"].n_nodes == 1 and d["info"].n_prim_refs == 0
    assert d["nodes"][0, 3] == ((1 << 30) | 1) and d["nodes"][0, 7] == 0xFFFFFFFF


@pytest.mark.parametrize("ties", [False, True])
def test_threaded_build_equals_the_serial_one_and_the_oracle(rt, orc, monkeypatch, ties):
    """Ranges of 32,768 objects and more are built by several threads (both the reference-order BVH,
    whose axis draws are made up front in pre-order, and the search structure).  The result must not depend on the
    number of threads, and the reference-order arrays must still be the oracle's -- also when most keys are equal
    (`ties`: centres on a coarse grid, one radius), where the order std::sort leaves equal keys in decides the tree."""
    def build(s):
        m = s.add_material(0, (0.5, 0.5, 0.5))
        rr = np.random.RandomState(17)
        c = (rr.rand(70000, 3) * 1000).astype(np.float32)
        rad = (0.5 + 3 * rr.rand(70000)).astype(np.float32)
        if ties:
            c = np.floor(c / 25).astype(np.float32) * 25
            rad[:] = 2.0
        for i in range(70000):
            s.add_sphere(c[i], float(rad[i]), m)
        s.wrap_bvh()

    monkeypatch.setenv("RT_BUILD_THREADS", "8")
    a, b = build_both(rt, orc, build)
    flat = a.get_flat()
    ea, ra = a.search_structure()
    after_a = a.rng_floats(4)
    monkeypatch.setenv("RT_BUILD_THREADS", "1")
    c = rt.Scene()
    build(c)
    flat_c = c.get_flat()
    ec, rc = c.search_structure()
    assert np.array_equal(flat["nodes"], flat_c["nodes"]) and np.array_equal(flat["prim_refs"], flat_c["prim_refs"])
    assert np.array_equal(ea, ec) and np.array_equal(ra, rc)
    assert np.array_equal(after_a.view(np.uint32), c.rng_floats(4).view(np.uint32))   # same generator state
    check_structure(flat, *b.dump_tree())
    assert np.array_equal(after_a.view(np.uint32), b.rng_floats(4).view(np.uint32))


def rebuild_through_group_calls(rt, src_scene, dst_scene):
    """Walks the flattened tree of `src_scene` the way an adapter walks an existing Hittable tree
    (BVHNode: box, left, right; list: objects in order) and hands it to `dst_scene` through the
    ingestion calls.  Returns {kind: [source id of destination primitive k]}."""
    f = src_scene.get_flat()
    nodes, refs = f["nodes"], f["prim_refs"]
    order = {0: [], 1: []}

    def add_prim(ref):
        kind, pid = (ref >> 30) & 1, int(ref & 0x3FFFFFFF)
        order[kind].append(pid)
        if kind == 0:
            c = f["spheres"][pid]
            dst_scene.add_sphere(c[:3], float(c[3]), int(f["sphere_mat"][pid]))
        else:
            dst_scene.add_triangle(f["tri_pos"][pid], int(f["tri_mat"][pid]), f["tri_normal"][pid], f["tri_uv"][pid])

    def emit(i):
        w = nodes[i]
        skip, nobox, first = int(w[3] & 0x3FFFFFFF), bool(w[3] & (1 << 30)), int(np.int32(w[7]))
        if nobox:
            dst_scene.begin_list()
        else:
            dst_scene.begin_bvh_node(np.concatenate([w[0:3], w[4:7]]).view(np.float32))
        if first >= 0:
            p = first
            while True:
                add_prim(int(refs[p]))
                if refs[p] >> 31:
                    break
                p += 1
        j = i + 1
        while j < skip:
            j = emit(j)
        dst_scene.end_group()
        return skip

    j = 0
    while j < len(nodes):
        j = emit(j)
    return order


@pytest.mark.parametrize("which", ["cover", "cornell_vase", "mixed"])
def test_caller_built_tree_is_ingested_as_it_is(rt, scenes, which):
    """rt_scene_begin_bvh_node / begin_list / end_group: a tree handed in node by node flattens to the
    same entries (boxes, skip links, leaf runs) as the tree the library built itself, without a single
    random number being drawn."""
    a = rt.Scene()
    if which == "cover":
        scenes.cover_scene(a, 320, 180)
    elif which == "cornell_vase":
        mesh = scenes.load_gltf_mesh(os.path.join(G, "brass_vase", "brass_vase_04_4k.gltf"))
        tex = {"diffuse": scenes.noise_texture(16, 1), "roughness": scenes.noise_texture(16, 2), "normal": None}
        scenes.cornell_mesh_scene(a, 320, 180, mesh, tex)
    else:
        m = a.add_material(0, (0.5, 0.5, 0.5))
        rr = np.random.RandomState(3)
        for _ in range(5):
            a.add_sphere((rr.rand(3) * 10).astype(np.float32), 0.5, m)     # loose objects in the world list
        a.add_box((0, 0, 0), (1, 2, 3), m)                                  # a plain list of 12 triangles
        lm = scenes.lathe_mesh(6, 4, seed=1)
        a.add_mesh(scenes.VASE_MODEL, lm["positions"], lm["indices"], m, normals=lm["normals"], max_depth=1)
        a.add_sphere((3.0, 3.0, 3.0), 1.0, m)
    b = rt.Scene()
    for k in range(a.flatten().n_materials):       # same materials, same indices
        fa = a.get_flat()
        kind, rgbp = int(fa["mat_kind"][k]), fa["mat_rgbp"][k]
        if kind == 4:
            td = b.add_texture(scenes.noise_texture(16, 1))
            tr = b.add_texture(scenes.noise_texture(16, 2))
            b.add_pbr(td, tr, -1)
        else:
            b.add_material(kind, rgbp[:3], float(rgbp[3]))
    before = b.rng_floats(1)[0]
    order = rebuild_through_group_calls(rt, a, b)
    fa, fb = a.get_flat(), b.get_flat()
    assert np.array_equal(fa["nodes"], fb["nodes"])                       # boxes, skip links, flags, run starts
    ra, rb = fa["prim_refs"], fb["prim_refs"]
    assert np.array_equal(ra >> 30, rb >> 30)                             # kinds and run ends
    for kind in (0, 1):
        sel = ((rb >> 30) & 1) == kind
        assert np.array_equal(np.asarray(order[kind])[rb[sel] & 0x3FFFFFFF], ra[sel] & 0x3FFFFFFF)
    # the generator was not touched: the next draw is the second value of a fresh stream
    fresh = rt.Scene().rng_floats(2)
    assert before == fresh[0] and b.rng_floats(1)[0] == fresh[1]


def test_group_calls_must_balance(rt):
    s = rt.Scene()
    with pytest.raises(rt.RtError):
        s.end_group()
    s.begin_list()
    s.add_sphere((0, 0, 0), 1.0, s.add_material(0, (0.5, 0.5, 0.5)))
    with pytest.raises(rt.RtError):
        s.flatten()
    with pytest.raises(rt.RtError):
        s.wrap_bvh()
    s.end_group()
    assert s.flatten().n_spheres == 1


def test_large_mesh_is_taken_in_by_several_threads_as_by_one(rt, orc, scenes, monkeypatch):
    """From 65,536 faces on, the faces of a mesh (transform, normal matrix, mesh box) and the boxes of its triangles
    are prepared in pieces on the builders' threads, and so are the padded boxes of the search structure: the
    arrays must not depend on the number of threads, and must still be the oracle's."""
    mesh = scenes.lathe_mesh(300, 120, seed=4)          # 72,000 triangles with smooth normals and uvs
    assert mesh["indices"].shape[0] >= 65536

    def build(s):
        m = s.add_material(0, (0.5, 0.5, 0.5))
        s.add_mesh(scenes.VASE_MODEL, mesh["positions"], mesh["indices"], m, normals=mesh["normals"], uvs=mesh["uvs"], max_depth=-1)

    monkeypatch.setenv("RT_BUILD_THREADS", "8")
    a, b = build_both(rt, orc, build)
    flat = a.get_flat()
    ea, ra = a.search_structure()
    monkeypatch.setenv("RT_BUILD_THREADS", "1")
    c = rt.Scene()
    build(c)
    flat_c = c.get_flat()
    ec, rc = c.search_structure()
    for key in ("nodes", "prim_refs", "tri_pos", "tri_normal", "tri_uv", "tri_mat"):
        assert np.array_equal(flat[key].view(np.uint32), flat_c[key].view(np.uint32)), key
    assert np.array_equal(ea, ec) and np.array_equal(ra, rc)
    compare_scene_arrays(flat, b.dump())
    check_structure(flat, *b.dump_tree())
    # an index beyond the vertices refuses the mesh before any of it has been taken in
    bad = mesh["indices"].copy()
    bad[-1, 2] = mesh["positions"].shape[0]
    d = rt.Scene()
    m = d.add_material(0, (0.5, 0.5, 0.5))
    with pytest.raises(rt.RtError):
        d.add_mesh(rt.IDENTITY, mesh["positions"], bad, m)
    assert d.flatten().n_triangles == 0


def test_threaded_sort_leaves_every_element_where_std_sort_does(tmp_path):
    """The BVH builder sorts ranges of 32,768 objects and more with its own threaded introsort, which must produce
    std::sort's permutation exactly (equal keys included: their order decides the tree).  tests/sort_check.cpp
    compares the two directly on inputs full of ties, for 1, 3 and 8 threads."""
    import subprocess
    here = os.path.dirname(os.path.abspath(__file__))
    exe = str(tmp_path / "sort_check")
    subprocess.run(["g++", "-std=c++17", "-O2", "-pthread", os.path.join(here, "sort_check.cpp"), "-o", exe], check=True)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stdout[-2000:]


def test_large_tree_with_a_composite_in_a_leaf_is_emitted_the_long_way(rt, orc, scenes):
    """Large built trees are emitted in pieces only if every leaf holds primitives; a top-level BVH over 70,000
    spheres and a mesh has one leaf that holds the mesh, so its entries come from the recursive emission -- while
    nothing of the abandoned attempt may be left behind -- and must be the oracle's all the same."""
    mesh = scenes.lathe_mesh(6, 4, seed=9)

    def build(s):
        m = s.add_material(0, (0.5, 0.5, 0.5))
        rr = np.random.RandomState(23)
        c = (rr.rand(70000, 3) * 1000).astype(np.float32)
        rad = (0.5 + 3 * rr.rand(70000)).astype(np.float32)
        for i in range(35000):
            s.add_sphere(c[i], float(rad[i]), m)
        s.add_mesh(scenes.VASE_MODEL, mesh["positions"], mesh["indices"], m, normals=mesh["normals"], uvs=mesh["uvs"], max_depth=2)
        for i in range(35000, 70000):
            s.add_sphere(c[i], float(rad[i]), m)
        s.wrap_bvh()

    a, b = build_both(rt, orc, build)
    flat = a.get_flat()
    compare_scene_arrays(flat, b.dump())
    check_structure(flat, *b.dump_tree())
