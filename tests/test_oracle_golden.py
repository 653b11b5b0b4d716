"""The CPU oracle against the committed golden vectors and the survey-recorded anchors.

The reference has no tests or fixtures of its own (its CMakeLists ends in a "TODO tests"
comment), so the strongest pins available are: (a) facts fixed by the C++ standard or
readable off the reference sources (mt19937's first outputs, the scene census), and (b) the
whole-image checksums recorded in anchors.json from the survey's probe of the reference.
"""
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
G = os.path.join(HERE, "golden")
ANCHORS = json.load(open(os.path.join(G, "anchors.json")))


def _cover(orc, scenes, w, h, wrap):
    s = orc.OracleScene()
    scenes.cover_scene(s, w, h, wrap_bvh=wrap)
    return s


def test_rng_stream_is_mt19937_scaled(orc):
    # mt19937 default seed: first outputs are fixed by the C++ standard
    first_u32 = [3499211612, 581869302, 3890346734, 3586334585, 545404204]
    f = orc.OracleScene().rng_floats(64)
    for v, u in zip(f, first_u32):
        assert v == np.float32(np.float32(u) / np.float32(4294967296.0))
    assert abs(float(f[0]) - ANCHORS["survey_recorded"]["first_draw"]) < 1e-7
    assert np.array_equal(f.view(np.uint32), np.load(os.path.join(G, "rng64_bits.npy")))
    assert f.min() >= 0.0 and f.max() < 1.0


def test_hash_random_is_the_shader_hash(orc):
    def hsh(x):
        x = (x + (x << 10)) & 0xFFFFFFFF
        x ^= x >> 6
        x = (x + (x << 3)) & 0xFFFFFFFF
        x ^= x >> 11
        x = (x + (x << 15)) & 0xFFFFFFFF
        return x

    def bits(v):
        return int(np.float32(v).view(np.uint32))

    for (x, y, s, c) in [(0, 0, 0, 1), (7, 3, 5, 2), (1919, 1079, 63, 200), (12, 900, 1023, 17)]:
        m = hsh(bits(x) ^ hsh(bits(y)) ^ hsh(bits(s)) ^ hsh(bits(c)))
        m = (m & 0x007FFFFF) | 0x3F800000
        want = np.uint32(m).view(np.float32) - np.float32(1.0)
        got = orc.hash_random(x, y, s, c)
        assert np.float32(got) == want
        assert 0.0 <= got < 1.0


def test_cover_scene_census_and_first_sphere(orc, scenes):
    s = _cover(orc, scenes, 200, 112, False)
    d = s.dump()
    a = ANCHORS["survey_recorded"]
    assert d["spheres"].shape[0] == a["n_spheres"]
    kinds = d["mat_kind"][d["sphere_mat"]]
    assert (kinds == 0).sum() == a["n_lambertian"]
    assert (kinds == 1).sum() == a["n_metal"]
    assert (kinds == 2).sum() == a["n_dielectric"]
    f = a["first_grid_sphere"]
    assert kinds[1] == f["kind"]
    assert np.allclose(d["spheres"][1, :3], f["center"], rtol=0, atol=1e-6)
    m = d["mat_rgbp"][d["sphere_mat"][1]]
    assert np.allclose(m[:3], f["albedo"], rtol=0, atol=1e-7)
    assert abs(m[3] - f["fuzz"]) < 1e-7
    # ground and the three big spheres
    assert tuple(d["spheres"][0]) == (0.0, -1000.0, 0.0, 1000.0)
    assert [tuple(x) for x in d["spheres"][-3:]] == [(0, 1, 0, 1), (-4, 1, 0, 1), (4, 1, 0, 1)]


def test_cover_scene_matches_golden(orc, scenes):
    s = _cover(orc, scenes, 200, 112, True)
    d = s.dump()
    g = np.load(os.path.join(G, "cover_scene.npz"))
    assert np.array_equal(d["spheres"].view(np.uint32), g["spheres"])
    assert np.array_equal(d["sphere_mat"], g["sphere_mat"])
    assert np.array_equal(d["mat_kind"], g["mat_kind"])
    assert np.array_equal(d["mat_rgbp"].view(np.uint32), g["mat_rgbp"])
    boxes, desc, prims = s.dump_tree()
    assert np.array_equal(boxes.view(np.uint32), g["boxes"])
    assert np.array_equal(desc, g["desc"])
    assert np.array_equal(prims, g["prims"])
    assert np.array_equal(s.get_camera().view(np.uint32), g["camera"])
    a = ANCHORS["survey_recorded"]
    assert boxes.shape[0] == a["bvh_internal_nodes"]
    assert list(boxes[0]) == a["bvh_root_box"]
    # every sphere appears exactly once in the visit order
    assert sorted(prims[:, 1].tolist()) == list(range(488))


def test_whole_image_checksums_match_the_survey_probe(orc, scenes):
    """200x112 at 1 spp, RaytracerNormal loop order, sequential mt19937 stream: the byte
    checksums equal the ones the survey recorded from the reference sources."""
    a = ANCHORS["survey_recorded"]
    for wrap, key in ((False, "checksum_200x112x1_flat"), (True, "checksum_200x112x1_bvh")):
        s = _cover(orc, scenes, 200, 112, wrap)
        _, rgba, _ = s.render(200, 112, 0, 1, 50, rng_mode=0, order=0)
        assert str(orc.checksum131(rgba.tobytes())) == a[key]
        assert (rgba[..., 3] == 255).all()


def test_small_mt_images_match_golden(orc, scenes):
    for name, wrap in (("flat", False), ("bvh", True)):
        s = _cover(orc, scenes, 64, 36, wrap)
        _, rgba, _ = s.render(64, 36, 0, 2, 50, rng_mode=0, order=0)
        assert np.array_equal(rgba, np.load(os.path.join(G, f"cover_mt_{name}_64x36x2.npy")))


def test_hit_table_matches_golden(orc, scenes):
    from golden.make_golden import golden_rays
    s = _cover(orc, scenes, 200, 112, True)
    kind, pid, t, nrm, front, _ = s.trace(golden_rays(4096, 1234))
    g = np.load(os.path.join(G, "cover_hits.npz"))
    assert np.array_equal(kind, g["kind"])
    assert np.array_equal(pid, g["pid"])
    assert np.array_equal(t.view(np.uint32), g["t"])
    assert np.array_equal(nrm.view(np.uint32), g["normal"])
    assert np.array_equal(front, g["front"])
    assert 0.2 < (kind >= 0).mean() < 1.0
    # flat list and BVH agree on what is hit (ties aside, which this ray set does not contain)
    s2 = _cover(orc, scenes, 200, 112, False)
    k2, p2, t2, *_ = s2.trace(golden_rays(4096, 1234))
    assert np.array_equal(k2, kind) and np.array_equal(p2, pid) and np.array_equal(t2, t)


def test_hash_images_match_golden_and_orders_agree(orc, scenes):
    g = np.load(os.path.join(G, "cover_hash_64x36x4.npz"))
    s = _cover(orc, scenes, 64, 36, True)
    acc1, _, _ = s.render(64, 36, 0, 4, 50, rng_mode=1, order=1, threads=4)
    assert np.array_equal(acc1.view(np.uint32), g["throughput"])
    s = _cover(orc, scenes, 64, 36, True)
    acc0, _, _ = s.render(64, 36, 0, 4, 50, rng_mode=1, order=0)
    assert np.array_equal(acc0.view(np.uint32), g["recursive"])
    # front-to-back throughput vs the reference's recursion: same paths, products re-associated
    rel = np.abs(acc1[..., :3] - acc0[..., :3]) / np.maximum(np.abs(acc0[..., :3]), 1e-6)
    assert rel.max() < 2e-6


def test_progressive_split_is_exact_in_hash_mode(orc, scenes):
    s = _cover(orc, scenes, 48, 27, True)
    whole, _, _ = s.render(48, 27, 0, 4, 50, rng_mode=1, order=1)
    acc, _, _ = s.render(48, 27, 0, 1, 50, rng_mode=1, order=1)
    acc, _, _ = s.render(48, 27, 1, 3, 50, rng_mode=1, order=1, accum=acc)
    assert np.array_equal(whole.view(np.uint32), acc.view(np.uint32))


def test_write_color_table(orc):
    sums = np.array([[0, 0, 0], [1, 1, 1], [4, 4, 4], [0.25, 0.5, 100.0], [1e-9, 3.996, 3.9999]], np.float32)
    out = orc.write_color(sums, 4)
    assert out[0].tolist() == [0, 0, 0, 255]
    assert out[1].tolist() == [128, 128, 128, 255]          # sqrt(1/4) = 0.5 -> 128
    assert out[2].tolist() == [255, 255, 255, 255]          # clamps at 0.999 -> 255
    assert out[3].tolist() == [64, int(256 * np.sqrt(np.float32(0.125))), 255, 255]
    assert out[4, 0] == 0 and out[4, 1] == 255


def test_vase_scene_matches_golden_and_survey_bounds(orc, scenes):
    from golden.make_golden import golden_rays
    mesh = scenes.load_gltf_mesh(os.path.join(G, "brass_vase", "brass_vase_04_4k.gltf"))
    a = ANCHORS["survey_recorded"]
    assert mesh["indices"].shape[0] == a["vase_triangles"]
    tex = {"diffuse": scenes.noise_texture(64, 1), "roughness": scenes.noise_texture(64, 2), "normal": None}
    s = orc.OracleScene()
    scenes.cornell_mesh_scene(s, 64, 36, mesh, tex)
    boxes, desc, prims = s.dump_tree()
    g = np.load(os.path.join(G, "vase_scene.npz"))
    assert np.array_equal(boxes.view(np.uint32), g["boxes"])
    assert np.array_equal(desc, g["desc"]) and np.array_equal(prims, g["prims"])
    # box 0 is the mesh AABB (with its 1e-8 max initialisation), box 1 the BVH root
    assert np.allclose(boxes[0], a["vase_world_bounds"], atol=0.06)
    d = s.dump()
    assert np.array_equal(d["tris"][12:12 + 64].view(np.uint32), g["tri0"])
    kind, pid, t, nrm, front, uv = s.trace(
        golden_rays(4096, 99, target=(278, 250, 278), spread=400.0, origin_box=((100, 100, -800), (450, 450, -100))))
    assert np.array_equal(kind, g["kind"]) and np.array_equal(pid, g["pid"])
    assert np.array_equal(t.view(np.uint32), g["t"])
    assert np.array_equal(nrm.view(np.uint32), g["normal"]) and np.array_equal(uv.view(np.uint32), g["uv"])
    assert (kind == 1).mean() > 0.9 and (pid[kind == 1] >= 12).mean() > 0.02
    acc, _, _ = s.render(64, 36, 0, 2, 50, rng_mode=1, order=1, threads=4)
    assert np.array_equal(acc.view(np.uint32), g["accum"])


def test_texture_lookup_index_math(orc, scenes):
    tex = np.arange(8 * 8 * 3, dtype=np.uint8).reshape(8, 8, 3)
    s = orc.OracleScene()
    t = s.add_texture(tex)
    lut = np.array([np.float32(np.power(np.float32(i / np.float32(255.0)), np.float32(2.2))) for i in range(256)])
    uv = np.array([[0.5, 0.5], [0.07, 0.93], [0.99, 0.01], [0.3, 0.6]], np.float32)
    got = s.texture_at(t, uv)
    for (u, v), rgb in zip(uv, got):
        x = int(np.float32(u * np.float32(8)) - np.float32(0.5))
        y = int(np.float32((np.float32(1) - v) * np.float32(8)) - np.float32(0.5))
        texel = tex.reshape(-1, 3)[y * 8 + x]
        assert np.allclose(rgb, lut[texel], rtol=1e-6)
