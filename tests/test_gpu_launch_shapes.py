"""The launch shapes of the render call -- single-pass loop, first pass + loop over path records, and the
library's own choice between them by launch size -- give the same bits."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def render(ctx, w, h, spp, depth=50):
    ctx.set_target(w, h)
    ctx.render(0, spp, depth)
    ctx.sync()
    return ctx.read_accum()


@pytest.mark.parametrize("wrap", [True, False])
def test_single_pass_two_passes_and_the_automatic_choice_agree(rt, orc, scenes, gpu_ctx, monkeypatch, wrap):
    """Small frame, all three settings, and the oracle."""
    w, h, spp = 192, 108, 5
    a, b = rt.Scene(), orc.OracleScene()
    for s in (a, b):
        scenes.cover_scene(s, w, h, wrap_bvh=wrap)
    gpu_ctx.upload_scene(a)
    want, _, _ = b.render(w, h, 0, spp, 50, rng_mode=1, order=1, threads=8)
    for setting in ("0", "1", None):
        if setting is None:
            monkeypatch.delenv("RT_TWO_PASS", raising=False)
        else:
            monkeypatch.setenv("RT_TWO_PASS", setting)
        got = render(gpu_ctx, w, h, spp)
        assert np.array_equal(bits(got[..., :3]), bits(want[..., :3])), f"RT_TWO_PASS={setting}"


def test_automatic_choice_at_the_size_where_two_passes_start(rt, scenes, gpu_ctx, monkeypatch):
    """1080p at 24 spp is 49.8 M samples: the library takes two passes there by itself, at 8 spp it does not (seen in
    the scheduler statistics of a counting launch of the same size: the loop over path records has no NEWPATH phase of
    its own); both equal the forced single pass bit for bit."""
    w, h = 1920, 1080
    a = rt.Scene()
    scenes.cover_scene(a, w, h)
    ctx = rt.Context(0)
    try:
        ctx.upload_scene(a)
        for spp, two_passes in ((8, False), (24, True)):
            monkeypatch.delenv("RT_TWO_PASS", raising=False)
            auto = render(ctx, w, h, spp)
            ctx.count(0, spp, 50)
            newpath_executions = ctx.sched_stats()["newpath"][0]
            assert (newpath_executions == 0) == two_passes
            monkeypatch.setenv("RT_TWO_PASS", "0")
            single = render(ctx, w, h, spp)
            assert np.array_equal(bits(auto), bits(single))
    finally:
        ctx.close()
