import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def hostsim():
    """ctypes handle on the CPU harness that runs the GPU lane programs on host loops (tests only)."""
    import __graft_entry__ as g
    from metaworld_amd import native
    return native.load("mwh_", g.build_host_harness())


@pytest.fixture(scope="session")
def gpulib():
    from metaworld_amd import native
    return native.load()


@pytest.fixture(scope="session")
def gpulib_split():
    """the -DMW_SPLIT_COLLISION variant of the library (the default library does not carry the split-collision kernels)"""
    import __graft_entry__ as g
    from metaworld_amd import native
    return native.Lib(g.build_gpu_split(), "mw_")
