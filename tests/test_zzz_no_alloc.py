"""rt_render_noalloc(): the render entry point that allocates nothing (its colour buffer and queue
counter live in a fixed device array inside the library) against rt_render(), through the C ABI.
Bar: identical bits.  Two calls in a row, the second on a smaller frame, so that whatever the
first call left in the shared array would show."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def test_noalloc_render_equals_render(rt, scenes, gpu_ctx):
    cases = [(640, 360, 3, 50), (200, 112, 5, 50)]      # the second input is smaller than the first
    scene = {}
    want = {}
    for w, h, spp, depth in cases:
        a = rt.Scene()
        scenes.cover_scene(a, w, h)
        scene[(w, h)] = a
        gpu_ctx.upload_scene(a)
        gpu_ctx.set_target(w, h)
        gpu_ctx.render(0, spp, depth)
        gpu_ctx.sync()
        want[(w, h)] = gpu_ctx.read_accum()
    for w, h, spp, depth in cases:                      # back to back through the entry point under test
        gpu_ctx.upload_scene(scene[(w, h)])
        gpu_ctx.set_target(w, h)
        gpu_ctx.render_noalloc(0, spp, depth)
        gpu_ctx.sync()
        got = gpu_ctx.read_accum()
        assert np.array_equal(bits(got), bits(want[(w, h)])), (w, h)
        assert got[..., :3].max() > 0
    # progressive use: consecutive sample ranges add up to the same sums
    w, h, spp, depth = cases[1]
    gpu_ctx.clear()
    gpu_ctx.render_noalloc(0, 2, depth)
    gpu_ctx.render_noalloc(2, spp - 2, depth)
    gpu_ctx.sync()
    assert np.array_equal(bits(gpu_ctx.read_accum()), bits(want[(w, h)]))


def test_noalloc_cuts_long_sample_ranges_and_refuses_oversized_frames(rt, scenes, gpu_ctx):
    # 1920x1080 has 2,073,600 pixel slots: 24.9 MB of colours per sample (12 bytes each), so 64 MiB holds two samples
    # per launch and five samples run as three launches
    w, h, spp = 1920, 1080, 5
    a = rt.Scene()
    scenes.cover_scene(a, w, h)
    gpu_ctx.upload_scene(a)
    gpu_ctx.set_target(w, h)
    gpu_ctx.render(0, spp, 50)
    gpu_ctx.sync()
    want = gpu_ctx.read_accum()
    gpu_ctx.clear()
    gpu_ctx.render_noalloc(0, spp, 50)
    gpu_ctx.sync()
    assert np.array_equal(bits(gpu_ctx.read_accum()), bits(want))
    # 3840x2160 = 8.3 M slots > rt_noalloc_max_pixels(): refused, nothing rendered
    assert rt.lib.rt_noalloc_max_pixels() == (64 << 20) // 12
    gpu_ctx.set_target(3840, 2160)
    with pytest.raises(rt.RtError) as e:
        gpu_ctx.render_noalloc(0, 1, 50)
    assert e.value.code == -6 and "scratch" in str(e.value)
    gpu_ctx.sync()
    assert not gpu_ctx.read_accum()[..., :3].any()
    # an eighth of the same frame (one of eight ranks) fits
    gpu_ctx.set_target(3840, 2160, 3, 8)
    gpu_ctx.render_noalloc(0, 1, 50)
    gpu_ctx.sync()
    part = gpu_ctx.read_accum()
    gpu_ctx.set_target(3840, 2160, 3, 8)
    gpu_ctx.render(0, 1, 50)
    gpu_ctx.sync()
    assert np.array_equal(bits(gpu_ctx.read_accum()), bits(part)) and part[..., :3].max() > 0
