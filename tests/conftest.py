import os
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


def _gpu_count():
    try:
        # torch first: it ships its own copy of the HIP runtime, and the process must end up with ONE runtime — the copy that is
        # loaded first wins (libidkpt.so's dependency then resolves to it by soname); the other order leaves torch without devices
        try:
            import torch  # noqa: F401
        except Exception:
            pass
        import ctypes
        from idkengine_amd import _lib
        n = ctypes.c_int32(0)
        _lib.load().idkptGetDeviceCount(ctypes.byref(n))
        return n.value
    except Exception:
        return 0


def pytest_collection_modifyitems(config, items):
    if _gpu_count() > 0:
        return
    skip = pytest.mark.skip(reason="no HIP device visible (GPU tests run on the MI355X box)")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Everything native is built in-tree before any test runs (hipcc cross-compiles without a GPU)."""
    from idkengine_amd import build as B
    B.build_all()
    from oracle import oracle as O
    O.build()


@pytest.fixture(scope="session")
def oracle_mod():
    from oracle import oracle as O
    return O


@pytest.fixture(scope="session")
def native_builder():
    from idkengine_amd.bvh import NativeBuilder
    return NativeBuilder()


@pytest.fixture(scope="session")
def oracle_builder(oracle_mod):
    return oracle_mod.OracleBuilder()
