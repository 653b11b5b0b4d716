"""Shared fixtures.  GPU tests are marked `gpu`; everything else runs without a device."""
import importlib.util
import os
import sys

import pytest

# PyTorch-ROCm bundles its own copy of the HIP runtime (same soname as /opt/rocm's).  A process that
# loads the product library first and torch afterwards ends up with two HIP/HSA runtimes, and the
# second one may find no GPU.  Loaded in this order the dynamic loader resolves the library's
# dependency to the runtime torch already brought in, so the process has exactly one.
try:
    import torch  # noqa: F401
except ImportError:
    pass

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG_DIR = os.path.join(ROOT, "raytracing-in-one-weekend_amd")
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def load_package():
    """Imports the product package (its directory name is not a Python identifier)."""
    if "rtiow_amd" in sys.modules:
        return sys.modules["rtiow_amd"]
    spec = importlib.util.spec_from_file_location(
        "rtiow_amd", os.path.join(PKG_DIR, "__init__.py"), submodule_search_locations=[PKG_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["rtiow_amd"] = mod
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(autouse=True)
def two_pass_launches(monkeypatch):
    """Left to itself the library renders launches of fewer than 48 M samples -- every frame of this suite -- with
    the single-pass loop.  The suite asks for the two-pass launch wherever a scene allows one, so that the first
    pass and the loop over its records are what the parity tests check; scenes in global memory, the entry point
    that allocates nothing and tests/test_gpu_launch_shapes.py run the single pass."""
    if "RT_TWO_PASS" not in os.environ:
        monkeypatch.setenv("RT_TWO_PASS", "1")


@pytest.fixture(scope="session")
def rt():
    return load_package()


@pytest.fixture(scope="session")
def scenes(rt):
    import importlib
    return importlib.import_module("rtiow_amd.scenes")


@pytest.fixture(scope="session")
def tiling(rt):
    import importlib
    return importlib.import_module("rtiow_amd.tiling")


@pytest.fixture(scope="session")
def orc():
    import _oracle
    _oracle.lib()
    return _oracle


@pytest.fixture(scope="session")
def gpu_ctx(rt):
    """One context on device 0 shared by the GPU tests (a box allows few GPU processes)."""
    if rt.device_count() < 1:
        pytest.fail("GPU test selected but no HIP device is visible; the product path has no CPU fallback")
    ctx = rt.Context(0)
    yield ctx
    ctx.close()
