import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # a `-m gpu` run without a device must fail loudly, not skip: the product path is HIP-only
    pass


@pytest.fixture(scope="session", autouse=True)
def _native_libraries_built():
    """A fresh checkout has no built .so (they are git-ignored): build once per session, exactly as
    the driver's __graft_entry__.build() does.  The package itself never builds or falls back."""
    lib = os.path.join(ROOT, "orthogonalsphericalshellgrids.jl_amd", "libtripolar_hip.so")
    if not os.path.exists(lib) or not os.path.exists(os.path.join(ROOT, "tools", "libtripolar_hip_test.so")):
        import __graft_entry__
        __graft_entry__.build()


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as o
    o.lib()
    return o


@pytest.fixture(scope="session")
def kats():
    import json
    with open(os.path.join(GOLDEN, "reference_kats.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def osg():
    import orthogonalsphericalshellgrids.jl_amd as m
    return m


@pytest.fixture(scope="session")
def tlib():
    """tools/libtripolar_hip_test.so: the product's objects + the test-only hooks (include/tripolar_hip_test.h)"""
    from tools import testlib
    return testlib.lib()


@pytest.fixture
def via_testlib():
    """the package's own calls go through the test library for this test (the TPG_* cross-check knobs live there)"""
    from tools import testlib
    with testlib.active() as handle:
        yield handle


@pytest.fixture(scope="session")
def gpu():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device (there is no CPU fallback)"
    torch.cuda.set_device(0)
    return torch.device("cuda", 0)
