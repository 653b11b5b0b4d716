"""Regenerates the golden vectors under tests/golden/ from the CPU oracle.

The reference holds no tests, fixtures or known-answer vectors for this path and cannot be
built here (its math library, glm, is an empty submodule), so these files are outputs of the
oracle restatement, committed to pin it against regressions.  `anchors.json` additionally
records values taken from a structural survey of the reference's sources (scene census,
first grid sphere, and two whole-image checksums of the reference's own sources rendering
200x112 at 1 spp), which the oracle reproduces.

Run:  make -C oracle && python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import _oracle  # noqa: E402
from conftest import load_package  # noqa: E402

load_package()
from rtiow_amd import scenes  # noqa: E402


def golden_rays(n, seed, target=(0.0, 0.5, 0.0), spread=6.0, origin_box=((-14, 0.2, -14), (14, 6, 14))):
    """Seeded rays aimed at a jittered point near `target` from origins in a box."""
    r = np.random.RandomState(seed)
    lo, hi = np.asarray(origin_box[0], np.float32), np.asarray(origin_box[1], np.float32)
    o = (lo + (hi - lo) * r.rand(n, 3)).astype(np.float32)
    tgt = (np.asarray(target, np.float32) + spread * (r.rand(n, 3) - 0.5)).astype(np.float32)
    d = (tgt - o).astype(np.float32)
    return np.concatenate([o, d], axis=1).astype(np.float32)


def main():
    out = {}
    # (1) RNG stream
    s = _oracle.OracleScene()
    np.save(os.path.join(HERE, "rng64_bits.npy"), s.rng_floats(64).view(np.uint32))

    # (2) cover scene + (3) BVH order
    s = _oracle.OracleScene()
    scenes.cover_scene(s, 200, 112, wrap_bvh=True)
    d = s.dump()
    boxes, desc, prims = s.dump_tree()
    np.savez_compressed(os.path.join(HERE, "cover_scene.npz"), spheres=d["spheres"].view(np.uint32),
                        sphere_mat=d["sphere_mat"], mat_kind=d["mat_kind"], mat_rgbp=d["mat_rgbp"].view(np.uint32),
                        boxes=boxes.view(np.uint32), desc=desc, prims=prims, camera=s.get_camera().view(np.uint32))
    # (4) ray -> hit table
    rays = golden_rays(4096, 1234)
    kind, pid, t, nrm, front, uv = s.trace(rays)
    np.savez_compressed(os.path.join(HERE, "cover_hits.npz"), kind=kind.astype(np.int8), pid=pid.astype(np.int16),
                        t=t.view(np.uint32), normal=nrm.view(np.uint32), front=front.astype(np.int8))
    # (8) whole images, sequential mt19937 stream, as RaytracerNormal with buildUp = false
    img_sums = {}
    for name, wrap in (("flat", False), ("bvh", True)):
        for (w, h, spp) in ((64, 36, 2), (200, 112, 1)):
            sc = _oracle.OracleScene()
            scenes.cover_scene(sc, w, h, wrap_bvh=wrap)
            _, rgba, _ = sc.render(w, h, 0, spp, 50, rng_mode=0, order=0)
            img_sums[f"{name}_{w}x{h}x{spp}"] = str(_oracle.checksum131(rgba.tobytes()))
            if w == 64:
                np.save(os.path.join(HERE, f"cover_mt_{name}_64x36x2.npy"), rgba)
    # hash-stream image (the device RNG contract), both evaluation orders
    sc = _oracle.OracleScene()
    scenes.cover_scene(sc, 64, 36, wrap_bvh=True)
    acc1, _, _ = sc.render(64, 36, 0, 4, 50, rng_mode=1, order=1)
    sc = _oracle.OracleScene()
    scenes.cover_scene(sc, 64, 36, wrap_bvh=True)
    acc0, _, _ = sc.render(64, 36, 0, 4, 50, rng_mode=1, order=0)
    np.savez_compressed(os.path.join(HERE, "cover_hash_64x36x4.npz"), throughput=acc1.view(np.uint32),
                        recursive=acc0.view(np.uint32))

    # Cornell box + brass vase (glTF fixture), synthetic textures
    mesh = scenes.load_gltf_mesh(os.path.join(HERE, "brass_vase", "brass_vase_04_4k.gltf"))
    tex = {"diffuse": scenes.noise_texture(64, 1), "roughness": scenes.noise_texture(64, 2), "normal": None}
    sc = _oracle.OracleScene()
    scenes.cornell_mesh_scene(sc, 64, 36, mesh, tex)
    boxes, desc, prims = sc.dump_tree()
    dd = sc.dump()
    rays = golden_rays(4096, 99, target=(278, 250, 278), spread=400.0, origin_box=((100, 100, -800), (450, 450, -100)))
    kind, pid, t, nrm, front, uv = sc.trace(rays)
    acc, _, _ = sc.render(64, 36, 0, 2, 50, rng_mode=1, order=1)
    np.savez_compressed(os.path.join(HERE, "vase_scene.npz"), boxes=boxes.view(np.uint32), desc=desc, prims=prims,
                        tri0=dd["tris"][12:12 + 64].view(np.uint32), kind=kind.astype(np.int8),
                        pid=pid.astype(np.int16), t=t.view(np.uint32), normal=nrm.view(np.uint32),
                        uv=uv.view(np.uint32), accum=acc.view(np.uint32))

    out["oracle_checksums"] = img_sums
    out["survey_recorded"] = {
        "comment": "values recorded by a probe of the reference sources (g++ build, sequential mt19937 stream)",
        "n_spheres": 488, "n_lambertian": 392, "n_metal": 71, "n_dielectric": 25,
        "first_draw": 0.81472367,
        "first_grid_sphere": {"kind": 1, "center": [-10.1847878, 0.2, -10.8780708],
                              "albedo": [0.984433889, 0.563493431, 0.917504311], "fuzz": 0.456687927},
        "bvh_internal_nodes": 511, "bvh_root_box": [-1000, -2000, -1000, 1000, 2, 1000],
        "checksum_200x112x1_flat": "12033002001534070305",
        "checksum_200x112x1_bvh": "6761946699157987483",
        "vase_triangles": 2760,
        "vase_world_bounds": [174.6, 100.2, 216.5, 338.5, 440.7, 338.5],
    }
    json.dump(out, open(os.path.join(HERE, "anchors.json"), "w"), indent=1)
    print(json.dumps(out["oracle_checksums"], indent=1))


if __name__ == "__main__":
    main()
