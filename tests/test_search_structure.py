"""Invariants of the structure the kernels search first (host side, no GPU).

The kernels' answer is only kept when the reference's own boxes confirm it, so a wrong search
structure cannot change an image -- but it could silently cull a closest hit often enough to
send every ray through the slow reference-order walk, or loop.  These tests pin what the walk
relies on: links only point forward, subtrees nest, every primitive of the world is listed exactly
once, and every entry's box encloses the true bounds of everything listed below it.
"""
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
G = os.path.join(HERE, "golden")
LAST, TRI = 1 << 31, 1 << 30
NONE = 0xFFFFFFFF


def prim_bounds(flat, ref):
    i = int(ref & (TRI - 1))
    if ref & TRI:
        p = flat["tri_pos"][i].reshape(3, 3)
        return p.min(axis=0), p.max(axis=0)
    c, r = flat["spheres"][i, :3], abs(flat["spheres"][i, 3])
    return c - r, c + r


def check_search_structure(scene, structure=None):
    """`structure`: (entries, refs) to check instead of the scene's host-built one (e.g. the
    device-built structure read back from a context)."""
    flat = scene.get_flat()
    entries, refs = structure if structure is not None else scene.search_structure()
    n = len(entries)
    assert n >= 1
    box = entries[:, :6].view(np.float32)
    skip = entries[:, 6].astype(np.int64)
    first = entries[:, 7]

    # forward links, nested subtrees
    for e in range(n):
        assert e < skip[e] <= n
    stack = []
    for e in range(n):
        while stack and stack[-1] <= e:
            stack.pop()
        if stack:
            assert skip[e] <= stack[-1], "a subtree reaches past the subtree that holds it"
        stack.append(skip[e])

    # leaf runs: terminated, in array order, covering refs exactly; leaves have no entries below
    pos = 0
    runs = {}
    for e in range(n):
        if first[e] != NONE:
            assert first[e] == pos
            assert skip[e] == e + 1, "an entry with a leaf run has no entries below it"
            start = pos
            while not (refs[pos] & LAST):
                pos += 1
            pos += 1
            runs[e] = (start, pos)
        else:
            assert skip[e] > e + 1, "an inner entry without anything below it"
    assert pos == len(refs)

    # every primitive of the world exactly once
    world = np.unique(flat["prim_refs"] & ~np.uint32(LAST))
    listed = refs & ~np.uint32(LAST)
    assert len(listed) == len(world) and np.array_equal(np.sort(listed), world)

    # boxes enclose the true bounds of everything below them
    lo = np.full((n, 3), np.inf, np.float32)
    hi = np.full((n, 3), -np.inf, np.float32)
    for e, (a, b) in runs.items():
        for r in refs[a:b]:
            mn, mx = prim_bounds(flat, r)
            lo[e] = np.minimum(lo[e], mn)
            hi[e] = np.maximum(hi[e], mx)
    for e in range(n - 1, -1, -1):   # fold every entry into the entries that hold it
        j = e + 1
        while j < skip[e]:
            lo[e] = np.minimum(lo[e], lo[j])
            hi[e] = np.maximum(hi[e], hi[j])
            j = skip[j]
    assert (box[:, 0::2] <= lo).all() and (box[:, 1::2] >= hi).all()
    return entries, refs


@pytest.mark.parametrize("wrap", [False, True])
def test_cover_scene(rt, scenes, wrap):
    s = rt.Scene()
    scenes.cover_scene(s, 640, 360, wrap_bvh=wrap)
    entries, refs = check_search_structure(s)
    assert len(refs) == 488
    # the structure does not depend on how the reference's own BVH was built
    t = rt.Scene()
    scenes.cover_scene(t, 640, 360, wrap_bvh=not wrap)
    e2, r2 = t.search_structure()
    assert np.array_equal(entries, e2) and np.array_equal(refs & ~np.uint32(LAST), r2 & ~np.uint32(LAST))


def test_cornell_and_vase(rt, scenes):
    mesh = scenes.load_gltf_mesh(os.path.join(G, "brass_vase", "brass_vase_04_4k.gltf"))
    tex = {"diffuse": scenes.noise_texture(16, 1), "roughness": scenes.noise_texture(16, 2), "normal": None}
    s = rt.Scene()
    scenes.cornell_mesh_scene(s, 640, 360, mesh, tex)
    _, refs = check_search_structure(s)
    assert len(refs) == 12 + 2760


def test_mixed_world(rt, scenes):
    mesh = scenes.lathe_mesh(6, 4, seed=9)
    s = rt.Scene()
    m = [s.add_material(0, (0.5, 0.4, 0.3)), s.add_material(1, (0.9, 0.8, 0.7), 0.2)]
    rr = np.random.RandomState(5)
    for i in range(40):
        c = (rr.rand(3) * 400 + 50).astype(np.float32)
        s.add_sphere(c, float(np.float32(5 + 20 * rr.rand())), m[i % 2])
    for i in range(9):
        s.add_triangle((rr.rand(9) * 500).astype(np.float32), m[i % 2])
    td = s.add_texture(scenes.noise_texture(8, 1))
    s.add_mesh(scenes.VASE_MODEL, mesh["positions"], mesh["indices"], s.add_pbr(td, -1, -1),
               normals=mesh["normals"], uvs=mesh["uvs"], max_depth=2)
    s.wrap_bvh()
    check_search_structure(s)


def test_coincident_and_single_primitives(rt):
    """All centroids equal (the split falls back to halving the list), and a world of one."""
    s = rt.Scene()
    m = s.add_material(0, (0.5, 0.5, 0.5))
    for _ in range(9):
        s.add_sphere((1.0, 2.0, 3.0), 0.5, m)
    check_search_structure(s)
    one = rt.Scene()
    one.add_sphere((0.0, 0.0, 0.0), 1.0, one.add_material(0, (1, 1, 1)))
    entries, refs = check_search_structure(one)
    assert len(entries) == 1 and len(refs) == 1


def test_empty_world(rt):
    s = rt.Scene()
    entries, refs = s.search_structure()
    assert len(entries) == 1 and len(refs) == 0
    b = entries[0, :6].view(np.float32)
    assert (b[0::2] > b[1::2]).all() and entries[0, 7] == NONE   # inverted box, no leaf run


def test_moved_eye_reorders_and_keeps_the_invariants(rt, scenes):
    """Children are visited nearer-to-the-eye first, so the structure follows the camera: a new
    eye position makes the next flatten rebuild it, the same position does not change it."""
    s = rt.Scene()
    scenes.cover_scene(s, 640, 360)
    e0, r0 = check_search_structure(s)
    cam = dict(lookat=(0.0, 0.0, 0.0), vup=(0.0, 1.0, 0.0), vfov=20.0, aspect=1.0, aperture=0.1, focus=10.0)
    s.set_camera((13.0, 2.0, 3.0), cam["lookat"], cam["vup"], cam["vfov"], cam["aspect"], cam["aperture"], cam["focus"])
    e1, r1 = s.search_structure()
    assert np.array_equal(e0, e1) and np.array_equal(r0, r1)
    s.set_camera((-13.0, 2.0, -3.0), cam["lookat"], cam["vup"], cam["vfov"], cam["aspect"], cam["aperture"], cam["focus"])
    e2, r2 = check_search_structure(s)
    assert e2.shape == e0.shape and not np.array_equal(e0, e2)
    assert np.array_equal(np.sort(r0 & ~np.uint32(LAST)), np.sort(r2 & ~np.uint32(LAST)))


def test_direction_ordered_copies(rt, scenes):
    """A scene too large for LDS carries eight copies of the entries, one per sign pattern of the ray
    direction: same boxes and leaf runs as the plain structure, links inside the copy or "walk over",
    and of two sibling subtrees the one that starts earlier along the copy's direction comes first."""
    small = rt.Scene()
    scenes.cover_scene(small, 640, 360)
    assert small.search_structure_octants() == (0, None)

    mesh = scenes.load_gltf_mesh(os.path.join(G, "brass_vase", "brass_vase_04_4k.gltf"))
    tex = {"diffuse": scenes.noise_texture(16, 1), "roughness": scenes.noise_texture(16, 2), "normal": None}
    s = rt.Scene()
    scenes.cornell_mesh_scene(s, 640, 360, mesh, tex)
    entries, refs = check_search_structure(s)
    stride, oct_entries = s.search_structure_octants()
    n = len(entries)
    assert stride == n + 1 and oct_entries.shape == (8 * stride, 8)
    total = 8 * stride
    plain = {tuple(e[:6]) + (e[7],) for e in entries}
    for k in range(8):
        base = k * stride
        c = oct_entries[base:base + stride].copy()
        # a copy stores {entry plane, exit plane} per axis for its sign pattern: max before min where the
        # direction is negative (bit a of k = axis a); put back in (min, max) order it holds ...
        f = c[:n, :6].view(np.float32)
        for a in range(3):
            lo, hi = f[:, 2 * a].copy(), f[:, 2 * a + 1].copy()
            assert ((lo >= hi) if (k >> a) & 1 else (lo <= hi)).all()
            if (k >> a) & 1:
                f[:, 2 * a], f[:, 2 * a + 1] = hi, lo
        # ... the same set of (box, leaf run) pairs, plus the closing entry
        assert {tuple(e[:6]) + (e[7],) for e in c[:n]} == plain
        last = c[n]
        assert np.isinf(last[:2].view(np.float32)).all() and last[6] == total and last[7] == NONE
        skip = c[:n, 6].astype(np.int64)
        inside = (skip > base + np.arange(n)) & (skip <= base + n)
        assert (inside | (skip == total)).all()
        local = np.where(skip == total, n, skip - base)
        stack = []
        for e in range(n):      # forward links, nested subtrees, as in the plain structure
            assert e < local[e] <= n
            while stack and stack[-1] <= e:
                stack.pop()
            if stack:
                assert local[e] <= stack[-1]
            stack.append(local[e])
        leaf = c[:n, 7] != NONE
        assert (local[leaf] == np.nonzero(leaf)[0] + 1).all()
