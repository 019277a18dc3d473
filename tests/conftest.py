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


@pytest.hookimpl(tryfirst=True)
def pytest_cmdline_main(config):
    """The CPU suite (`-m "not gpu"`: the kernels on the SIMT emulator, the oracle under sanitizers, gloo ranks) is fifteen minutes
    on one core and four on six: when pytest-xdist is installed and the caller did not choose (`-n ...`), it runs over up to six
    worker processes.  Nothing is shared between tests but the built libraries (the build functions take a file lock) and fixed,
    distinct rendezvous ports.  The GPU suite (`-m gpu`) is never split: its tests at stated size each want the device to
    themselves.  WGA_TEST_PROCS=<n> overrides (1: one process)."""
    if os.environ.get("PYTEST_XDIST_WORKER") or not config.pluginmanager.hasplugin("xdist"):
        return None
    if (config.getoption("markexpr", "") or "").strip() != "not gpu" or config.getoption("numprocesses", None) is not None:
        return None
    want = os.environ.get("WGA_TEST_PROCS")
    procs = int(want) if want and want.isdigit() else min(6, os.cpu_count() or 1)
    if procs > 1:
        config.option.numprocesses = procs
    return None


def pytest_collection_modifyitems(config, items):
    """a plain `pytest tests` on a box without an MI355X skips the GPU tests instead of failing them one by one
    (`-m gpu` on such a box still fails loudly: the product has no CPU fallback)"""
    if "gpu" in (config.getoption("-m") or ""):
        return
    from wgatools_amd import build
    have = False
    if os.path.exists(build.HIP_LIB):
        try:
            import torch  # noqa: F401  first, as in _gpu_engine()
            import ctypes
            have = ctypes.CDLL(build.HIP_LIB).wga_device_count() > 0
        except OSError:
            have = False
    if have:
        return
    skip = pytest.mark.skip(reason="no MI355X visible (GPU tests: run with -m gpu on the GPU box)")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


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
