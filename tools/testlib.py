"""ctypes binding of tools/libtripolar_hip_test.so (include/tripolar_hip_test.h) -- TEST / BENCH INFRASTRUCTURE.

The test library is every object of the product library plus the test-only hooks (synthetic field fill, the same-shape copy
probe of the fold, the elementary-function probe) and the TPG_* cross-check knobs.  tests/, tools/ and bench.py load it;
the package never does.  `lib()` is the handle (all product symbols + the hooks); `active()` makes the package's own calls
(osg.TripolarGrid, osg.fill_halo_regions, ...) go through it for the duration of a `with` block, which is how the
kernel-variant tests reach the knobs."""
import contextlib
import ctypes as C
import os

from orthogonalsphericalshellgrids.jl_amd import _lib as product

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libtripolar_hip_test.so")

_vp, _i = C.c_void_p, C.c_int
_geom = [_i] * 6
TEST_SIGNATURES = {
    "tpg_reload_config": (_i, []),
    "tpg_zipper_copy_probe": (_i, [C.POINTER(_vp), _i, C.POINTER(C.c_int8)] + _geom + [_i, _vp, _vp, _vp]),
    "tpg_fill_synthetic": (_i, [_vp, C.c_uint64, C.c_double] + _geom + [_i, _vp]),
    "tpg_math_probe": (_i, [_i, _vp, _vp, _vp, C.c_longlong, _vp]),
}

_handle = None


def lib():
    global _handle
    if _handle is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} not found: build it with `make -C orthogonalsphericalshellgrids.jl_amd/csrc`")
        _handle = product.bind(LIB_PATH, {**product.SIGNATURES, **TEST_SIGNATURES})
    return _handle


def check(status):
    if status != 0:
        raise product.TripolarHipError(status, lib().tpg_last_error().decode("utf-8", "replace"))


@contextlib.contextmanager
def active():
    """route the package's calls through the test library (same kernels, knobs honoured)"""
    saved = product._lib
    product._lib = lib()
    try:
        yield lib()
    finally:
        product._lib = saved
