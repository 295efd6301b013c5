"""ctypes front-end of oracle/libtpg_oracle.so -- TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module
(see oracle/tpg_oracle.c header).  It is the checker, never the thing measured or shipped.
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libtpg_oracle.so")

# order of src/tripolar_grid.jl:308-328 (z omitted); shared with the product ABI by design
ARRAY_NAMES = (
    "lambda_cc", "lambda_fc", "lambda_cf", "lambda_ff",
    "phi_cc", "phi_fc", "phi_cf", "phi_ff",
    "dx_cc", "dx_fc", "dx_cf", "dx_ff",
    "dy_cc", "dy_cf", "dy_fc", "dy_ff",
    "az_cc", "az_fc", "az_cf", "az_ff",
)
# x/y location of each of the 20 arrays (0 = Center, 1 = Face)
ARRAY_LOCS = {n: (1 if n[-2] == "f" else 0, 1 if n[-1] == "f" else 0) for n in ARRAY_NAMES}


class Params(C.Structure):
    _fields_ = [
        ("Nx", C.c_int32), ("Ny", C.c_int32), ("Nz", C.c_int32),
        ("Hx", C.c_int32), ("Hy", C.c_int32), ("Hz", C.c_int32),
        ("southernmost_latitude", C.c_double),
        ("north_poles_latitude", C.c_double),
        ("first_pole_longitude", C.c_double),
        ("radius", C.c_double),
        ("ft", C.c_int32), ("jstart", C.c_int32), ("jend", C.c_int32), ("reserved", C.c_int32),
    ]


def build_library(force=False):
    if force or not os.path.exists(_LIB_PATH) or any(
            os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(_LIB_PATH)
            for f in ("tpg_oracle.c", "detmath.h")):
        subprocess.check_call(["make", "-C", _HERE, "-s"], stdout=sys.stderr)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build_library()
        _lib = C.CDLL(_LIB_PATH)
        _lib.tpo_build_grid.argtypes = [C.POINTER(Params), C.POINTER(C.c_void_p)]
        _lib.tpo_build_grid.restype = C.c_int
        for f in (_lib.tpo_zipper_fill,):
            f.argtypes = [C.c_void_p] + [C.c_int] * 12
            f.restype = C.c_int
        _lib.tpo_periodic_x_fill.argtypes = [C.c_void_p] + [C.c_int] * 7
        _lib.tpo_fill_halo_regions.argtypes = [C.c_void_p] + [C.c_int] * 10
        _lib.tpo_nonorthogonality_angle.argtypes = [C.c_void_p] * 4 + [C.c_int] * 5
        _lib.tpo_convert_frame.argtypes = [C.c_void_p] * 8 + [C.c_int] * 8
        _lib.tpo_math_probe.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_long]
        _lib.tpo_math_probe.restype = None
        _lib.tpo_tables.argtypes = [C.POINTER(Params)] + [C.c_void_p] * 4
        _lib.tpo_tables.restype = None
        _lib.tpo_stretch_tables.argtypes = [C.POINTER(Params)] + [C.c_void_p] * 4
        _lib.tpo_stretch_tables.restype = None
    return _lib


R_EARTH = 6371.0e3  # Oceananigans.Grids.R_Earth [recalled]


def make_params(size, halo=(4, 4, 4), southernmost_latitude=-80, north_poles_latitude=55,
                first_pole_longitude=70, radius=R_EARTH, dtype=np.float64, jstart=None, jend=None):
    Nx, Ny, Nz = size
    Hx, Hy, Hz = halo
    return Params(Nx, Ny, Nz, Hx, Hy, Hz, float(southernmost_latitude), float(north_poles_latitude),
                  float(first_pole_longitude), float(radius), 1 if np.dtype(dtype) == np.float64 else 0,
                  1 if jstart is None else jstart, Ny if jend is None else jend, 0)


def set_threads(n):
    lib().tpo_set_threads(int(n))


def max_threads():
    return int(lib().tpo_max_threads())


def build_grid(size, dtype=np.float64, **kw):
    """Reference TripolarGrid(...) metric precompute -> dict name -> padded array [j, i] (numpy,
    C-order, so that arr[j + Hy - 1, i + Hx - 1] is the reference's A[i, j])."""
    p = make_params(size, dtype=dtype, **kw)
    rows = p.jend - p.jstart + 1 + 2 * p.Hy
    cols = p.Nx + 2 * p.Hx
    arrs = [np.empty((rows, cols), dtype=dtype) for _ in ARRAY_NAMES]
    ptrs = (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
    rc = lib().tpo_build_grid(C.byref(p), ptrs)
    if rc == -2:
        raise ValueError("The number of cells in the longitude dimension should be even!")
    if rc != 0:
        raise RuntimeError(f"tpo_build_grid failed: {rc}")
    return dict(zip(ARRAY_NAMES, arrs))


def _ft(a):
    if a.dtype == np.float64:
        return 1
    if a.dtype == np.float32:
        return 0
    raise TypeError(a.dtype)


def _check_field(field, Nx, Ny, Nz, Hx, Hy, Hz):
    assert field.flags.c_contiguous and field.shape == (Nz + 2 * Hz, Ny + 2 * Hy, Nx + 2 * Hx), field.shape


def zipper_fill(field, xloc, yloc, sign, size, halo, kstart=1, kcount=None):
    """In-place fold_north_*! over (i, k) in 1..Nx x kstart..kstart+kcount-1.
    field: numpy [k, j, i] C-order padded parent array."""
    (Nx, Ny, Nz), (Hx, Hy, Hz) = size, halo
    _check_field(field, Nx, Ny, Nz, Hx, Hy, Hz)
    kcount = Nz if kcount is None else kcount
    lib().tpo_zipper_fill(field.ctypes.data, xloc, yloc, sign, Nx, Ny, Nz, Hx, Hy, Hz, kstart, kcount, _ft(field))
    return field


def periodic_x_fill(field, size, halo):
    (Nx, Ny, Nz), (Hx, Hy, Hz) = size, halo
    _check_field(field, Nx, Ny, Nz, Hx, Hy, Hz)
    lib().tpo_periodic_x_fill(field.ctypes.data, Nx, Ny, Nz, Hx, Hy, Hz, _ft(field))
    return field


def fill_halo_regions(field, xloc, yloc, sign, size, halo):
    (Nx, Ny, Nz), (Hx, Hy, Hz) = size, halo
    _check_field(field, Nx, Ny, Nz, Hx, Hy, Hz)
    lib().tpo_fill_halo_regions(field.ctypes.data, xloc, yloc, sign, Nx, Ny, Nz, Hx, Hy, Hz, _ft(field))
    return field


def nonorthogonality_angle(lam_ff, phi_ff, size, halo, immersed=None):
    """compute_nonorthogonality_angle! over (Nx-1, Ny-1) (test/test_tripolar_grid.jl:8-34,70) -> (Ny, Nx) Float64"""
    (Nx, Ny, _), (Hx, Hy, _) = size, halo
    assert lam_ff.shape == (Ny + 2 * Hy, Nx + 2 * Hx) and lam_ff.dtype == phi_ff.dtype and lam_ff.flags.c_contiguous
    angle = np.empty((Ny, Nx), dtype=np.float64)
    mask = None if immersed is None else np.ascontiguousarray(immersed, dtype=np.uint8)
    lib().tpo_nonorthogonality_angle(lam_ff.ctypes.data, phi_ff.ctypes.data, None if mask is None else mask.ctypes.data,
                                     angle.ctypes.data, Nx, Ny, Hx, Hy, _ft(lam_ff))
    return angle


def convert_frame(grid, u, v, size, halo, to_native=False):
    """convert_to_latlong_frame / convert_to_native_frame (examples/convert_to_latlong_frame.jl:12-55) on the interior of
    two padded (Center, Center, Center) parents; grid: dict of the padded 2-D arrays.  Returns (u_out, v_out), zero outside
    the interior."""
    (Nx, Ny, Nz), (Hx, Hy, Hz) = size, halo
    _check_field(u, Nx, Ny, Nz, Hx, Hy, Hz); _check_field(v, Nx, Ny, Nz, Hx, Hy, Hz)
    uo, vo = np.zeros_like(u), np.zeros_like(v)
    g = [np.ascontiguousarray(grid[n], dtype=u.dtype) for n in ("phi_cf", "phi_fc", "dy_cc", "dx_cc")]
    lib().tpo_convert_frame(*[a.ctypes.data for a in g], u.ctypes.data, v.ctypes.data, uo.ctypes.data, vo.ctypes.data,
                            1 if to_native else 0, Nx, Ny, Nz, Hx, Hy, Hz, _ft(u))
    return uo, vo


MATH_FUNCS = {"sin": 0, "cos": 1, "sind": 2, "cosd": 3, "tand": 4, "atan": 5, "asin": 6,
              "asinh": 7, "sinh": 8, "cosh": 9, "acos": 10}


def math_probe(name, x):
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.empty_like(x)
    lib().tpo_math_probe(MATH_FUNCS[name], x.ctypes.data, y.ctypes.data, x.size)
    return y


def tables(size, dtype=np.float64, **kw):
    p = make_params(size, dtype=dtype, **kw)
    lf, lc = np.empty(p.Nx), np.empty(p.Nx)
    pf, pc = np.empty(p.Ny), np.empty(p.Ny)
    lib().tpo_tables(C.byref(p), lf.ctypes.data, lc.ctypes.data, pf.ctypes.data, pc.ctypes.data)
    return lf, lc, pf, pc


def stretch_tables(size, **kw):
    p = make_params(size, **kw)
    out = [np.empty(p.Ny) for _ in range(4)]
    lib().tpo_stretch_tables(C.byref(p), *[o.ctypes.data for o in out])
    return out
