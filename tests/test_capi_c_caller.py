"""The C ABI driven from C: tests/capi_driver.c is compiled by the system C compiler (strict C99) against
include/rtiow_hip.h and linked with the product library, like the binding a maintainer of the reference
would write.  (Built by tests/Makefile from __graft_entry__.build(); not built inside the tests.)"""
import ctypes
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
DRIVER = os.path.join(HERE, "capi_driver")


def run_driver(*args, timeout=120):
    if not os.path.exists(DRIVER):
        pytest.fail("tests/capi_driver is missing: run `make -C tests` (or __graft_entry__.build())")
    p = subprocess.run([DRIVER, *args], capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, p.stderr
    return p.stdout


def test_struct_layout_seen_from_c_equals_the_ctypes_binding(rt):
    lines = dict(l.split(" ", 1) for l in run_driver("layout").strip().splitlines())
    c = [int(x) for x in lines["rt_counters"].split()]
    assert c[0] == ctypes.sizeof(rt.Counters)
    assert c[1:] == [getattr(rt.Counters, n).offset for n, _ in rt.Counters._fields_]
    f = [int(x) for x in lines["rt_flat_info"].split()]
    assert f[0] == ctypes.sizeof(rt.FlatInfo)
    assert f[1:] == [getattr(rt.FlatInfo, n).offset for n, _ in rt.FlatInfo._fields_]
    assert [int(x) for x in lines["status"].split()] == [0, -1, -2, -3, -4, -5, -6]
    assert rt.RT_ERR_NO_DEVICE == -2 and rt.RT_ERR_CANCELLED == -5


def test_camera_members_round_trip_bitwise(rt, scenes):
    """rt_scene_set_camera_members takes the eight derived members (R/src/Core/Camera.h:35-38) as they are."""
    s = rt.Scene()
    scenes.cover_scene(s, 160, 90)
    cam = s.get_camera()
    rng = np.random.default_rng(5)
    x = rng.standard_normal(22).astype(np.float32) * np.float32(1e3)
    x[5] = np.float32(-0.0)
    x[7] = np.float32(1e-42)       # a denormal survives
    x[9] = np.float32(np.inf)      # nothing is checked or normalised
    s.set_camera_members(x)
    assert np.array_equal(s.get_camera().view(np.uint32), x.view(np.uint32))
    s.set_camera_members(cam)
    assert np.array_equal(s.get_camera().view(np.uint32), cam.view(np.uint32))
    with pytest.raises(ValueError):
        s.set_camera_members(np.zeros(21, np.float32))
    assert rt.lib.rt_scene_set_camera_members(s.h, None) == -1
    assert rt.lib.rt_scene_set_camera_members(None, cam.ctypes.data_as(ctypes.c_void_p)) == -1


def test_texture_map_is_strictly_increasing(rt):
    """What lets an adapter recover a Texture's bytes from the floats stbi_loadf left it with."""
    lut = rt.texture_lut()
    assert (np.diff(lut) > 0).all()
    assert np.array_equal(np.searchsorted(lut, lut), np.arange(256))


@pytest.mark.gpu
def test_c_caller_renders_the_oracles_bytes(rt, scenes, orc, tmp_path):
    w, h, spp = 160, 90, 2
    out = tmp_path / "frame.rgba8"
    text = run_driver("render", str(w), str(h), str(spp), str(out))
    got = np.fromfile(out, np.uint8).reshape(h, w, 4)
    o = orc.OracleScene()
    scenes.cover_scene(o, w, h)
    _, want8, _ = o.render(w, h, 0, spp, 50, rng_mode=1, order=1, threads=4)
    assert np.array_equal(got, want8)
    fields = text.split()
    stats = dict(zip(fields[0::2], (int(x) for x in fields[1::2])))
    assert stats["spheres"] == 488 and stats["passes"] == spp and stats["samples"] == w * h
    assert stats["hits"] + stats["misses"] <= stats["rays"]


@pytest.mark.gpu
def test_camera_handed_over_as_members_renders_the_same_bits(rt, scenes, gpu_ctx):
    """A scene whose camera came in as the 22 member floats equals the constructor-argument path bit for bit."""
    w, h, spp = 200, 112, 3
    a = rt.Scene()
    scenes.cover_scene(a, w, h)
    gpu_ctx.upload_scene(a)
    gpu_ctx.set_target(w, h)
    gpu_ctx.render(0, spp, 50)
    gpu_ctx.sync()
    want = gpu_ctx.read_accum()
    b = rt.Scene()
    b.random_scene()
    b.wrap_bvh()
    b.set_background(scenes.COVER_BACKGROUND)
    b.set_camera_members(a.get_camera())       # no constructor arguments on this side
    gpu_ctx.upload_scene(b)
    gpu_ctx.set_target(w, h)
    gpu_ctx.render(0, spp, 50)
    gpu_ctx.sync()
    got = gpu_ctx.read_accum()
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
