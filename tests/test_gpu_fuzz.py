"""Randomised worlds through the GPU path and the oracle (a slice of profiles/fuzz_parity.py, which
ran 60,000 worlds = 531 M rays without a differing bit): spheres of every material, loose triangles,
meshes with and without smooth normals / PBR maps / normal maps / depth limits -- baked, or in every fourth world
held once and placed by transforms --, coincident primitives, cameras inside and outside the world, ragged frame sizes.  Seed 61890 is kept by name: paths that bounce
inside a glass ground sphere reach the small spheres from 66 units away, where Sphere::Hit's discriminant
(R/src/Core/Hittable.h:236-239) is so inaccurate that it reports hits for rays that pass a sphere at 1.5 radii
-- which the search structure has to let through (a fixed padding did not)."""
import importlib.util
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_random_worlds_render_bit_exact(rt, orc, gpu_ctx):
    spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(ROOT, "profiles", "fuzz_parity.py"))
    fuzz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fuzz)
    rays = 0
    for seed in list(range(5000, 5120)) + [61890]:
        build = fuzz.make_builder(seed)
        a, b = rt.Scene(), orc.OracleScene()
        build(a)
        build(b)
        rr = np.random.RandomState(seed + 7)
        w, h = int(rr.randint(2, 97)), int(rr.randint(2, 71))
        spp, depth = int(rr.randint(1, 5)), int(rr.choice([1, 3, 50]))
        gpu_ctx.upload_scene(a)
        gpu_ctx.set_target(w, h)
        gpu_ctx.render(0, spp, depth)
        gpu_ctx.sync()
        got = gpu_ctx.read_accum()
        want, _, _ = b.render(w, h, 0, spp, depth, rng_mode=1, order=1, threads=8)
        assert np.array_equal(got[..., :3].view(np.uint32), want[..., :3].view(np.uint32)), seed
        rays += gpu_ctx.count(0, spp, depth)["rays"]
    assert rays > 500_000
