import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_engine():
    import torch  # noqa: F401  first: libwgahip.so then shares torch's HIP runtime (same SONAME)
    from wgatools_amd import build, engine, _lib
    # the product library; on the GPU box the prebuilt in-tree .so travels with the snapshot
    if not os.path.exists(build.HIP_LIB):
        build.build_hip()
    return engine.Engine(0, _lib.load(build.HIP_LIB))


def _emu_engine():
    from wgatools_amd import build, engine, _lib
    return engine.Engine(0, _lib.load(build.build_emu()))


@pytest.fixture(scope="session")
def gpu():
    """Engine on libwgahip.so / cuda:0 — the product path.  No fallback: fails without a GPU."""
    eng = _gpu_engine()
    yield eng
    eng.close()


@pytest.fixture(scope="session")
def emu():
    """Engine on the SIMT-emulator build of the same kernel source (CPU logic tests)."""
    eng = _emu_engine()
    yield eng
    eng.close()
