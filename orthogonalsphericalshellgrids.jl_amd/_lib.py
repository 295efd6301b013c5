"""ctypes binding of libtripolar_hip.so (include/tripolar_hip.h).

This is the ONLY compute backend of the package: if the shared library is missing or a call
fails, the caller gets an exception -- there is no CPU or PyTorch fallback by design.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libtripolar_hip.so")

TPG_F32, TPG_F64 = 0, 1
TPG_CENTER, TPG_FACE = 0, 1
TPG_MAX_FIELDS = 16
TPG_BUILD_TABLES_VALID = 1        # tpg_params.reserved flag

# enum tpg_array (order of src/tripolar_grid.jl:308-328 in the reference)
ARRAY_NAMES = (
    "lambda_cc", "lambda_fc", "lambda_cf", "lambda_ff",
    "phi_cc", "phi_fc", "phi_cf", "phi_ff",
    "dx_cc", "dx_fc", "dx_cf", "dx_ff",
    "dy_cc", "dy_cf", "dy_fc", "dy_ff",
    "az_cc", "az_fc", "az_cf", "az_ff",
)

STATUS = {
    0: "TPG_OK", -1: "TPG_ERR_INVALID_ARGUMENT", -2: "TPG_ERR_ODD_NLAMBDA", -3: "TPG_ERR_BAD_PARTITION",
    -4: "TPG_ERR_WORKSPACE", -5: "TPG_ERR_UNSUPPORTED", -6: "TPG_ERR_NOT_NORTH", -7: "TPG_ERR_RCCL",
}
TPG_COMM_ID_BYTES = 128


class TpgParams(C.Structure):
    """struct tpg_params"""
    _fields_ = [
        ("Nx", C.c_int32), ("Ny", C.c_int32), ("Nz", C.c_int32),
        ("Hx", C.c_int32), ("Hy", C.c_int32), ("Hz", C.c_int32),
        ("southernmost_latitude", C.c_double),
        ("north_poles_latitude", C.c_double),
        ("first_pole_longitude", C.c_double),
        ("radius", C.c_double),
        ("ft", C.c_int32), ("jstart", C.c_int32), ("jend", C.c_int32), ("reserved", C.c_int32),
    ]


class TripolarHipError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(f"libtripolar_hip: {STATUS.get(status, status)}: {message}")
        self.status = status


# every symbol include/tripolar_hip.h declares: (restype, argtypes)
_vp, _i, _sz = C.c_void_p, C.c_int, C.c_size_t
_geom = [_i] * 6
SIGNATURES = {
    "tpg_version": (_i, []),
    "tpg_last_error": (C.c_char_p, []),
    "tpg_status_string": (C.c_char_p, [_i]),
    "tpg_build_grid_workspace_bytes": (_sz, [C.POINTER(TpgParams)]),
    "tpg_build_grid": (_i, [C.POINTER(TpgParams), C.POINTER(_vp), _vp, _sz, _vp]),
    "tpg_zipper_fill": (_i, [C.POINTER(_vp), _i, C.POINTER(C.c_int8), C.POINTER(C.c_int8), C.POINTER(C.c_int32)]
                        + _geom + [_i, _i, _i, _vp]),
    "tpg_zipper_fill_timed": (_i, [C.POINTER(_vp), _i, C.POINTER(C.c_int8), C.POINTER(C.c_int8), C.POINTER(C.c_int32)]
                              + _geom + [_i, _i, _i, _vp, _vp, _vp]),
    "tpg_event_create": (_i, [C.POINTER(_vp)]),
    "tpg_event_destroy": (_i, [_vp]),
    "tpg_event_elapsed_ms": (_i, [_vp, _vp, C.POINTER(C.c_float)]),
    "tpg_periodic_x_fill": (_i, [C.POINTER(_vp), _i] + _geom + [_i, _vp]),
    "tpg_fill_halo_regions": (_i, [C.POINTER(_vp), _i, C.POINTER(C.c_int8), C.POINTER(C.c_int8), C.POINTER(C.c_int32)]
                              + _geom + [_i, _i, _vp]),
    "tpg_fill_halo_regions_timed": (_i, [C.POINTER(_vp), _i, C.POINTER(C.c_int8), C.POINTER(C.c_int8), C.POINTER(C.c_int32)]
                                    + _geom + [_i, _i, _vp, _vp, _vp]),
    "tpg_y_halo_buffer_elems": (_sz, [_i] * 6),
    "tpg_pack_y_halo": (_i, [C.POINTER(_vp), _i, _vp, _i] + _geom + [_i, _vp]),
    "tpg_unpack_y_halo": (_i, [C.POINTER(_vp), _i, _vp, _i] + _geom + [_i, _vp]),
    "tpg_comm_available": (_i, []),
    "tpg_comm_unique_id": (_i, [_vp]),
    "tpg_comm_init_rank": (_i, [C.POINTER(_vp), _i, _vp, _i]),
    "tpg_comm_destroy": (_i, [_vp]),
    "tpg_halo_exchange_y": (_i, [_vp, _i, _i, C.POINTER(_vp), _i, _vp, _vp, _vp, _vp] + _geom + [_i, _vp]),
    "tpg_halo_exchange_y_peers": (_i, [_vp, _i, _i, C.POINTER(_vp), _i, _vp, _vp, _vp, _vp] + _geom + [_i, _vp]),
    "tpg_halo_exchange_y_pipelined": (_i, [_vp, _i, _i, C.POINTER(_vp), _i, _vp, _vp, _vp, _vp] + _geom + [_i, _vp, _vp, _i]),
    "tpg_halo_exchange_y_pipelined_peers": (_i, [_vp, _i, _i, C.POINTER(_vp), _i, _vp, _vp, _vp, _vp] + _geom + [_i, _vp, _vp, _i]),
    "tpg_fill_halo_regions_distributed_pipelined": (_i, [_vp, _i, _i, C.POINTER(_vp), _i, C.POINTER(C.c_int8), C.POINTER(C.c_int8),
                                                         C.POINTER(C.c_int32), _vp, _vp, _vp, _vp] + _geom + [_i, _vp, _vp, _i]),
    "tpg_fill_halo_regions_distributed_pipelined_peers": (_i, [_vp, _i, _i, _i, C.POINTER(_vp), _i, C.POINTER(C.c_int8), C.POINTER(C.c_int8),
                                                               C.POINTER(C.c_int32), _vp, _vp, _vp, _vp] + _geom + [_i, _vp, _vp, _i]),
    "tpg_fill_halo_regions_distributed": (_i, [_vp, _i, _i, C.POINTER(_vp), _i, C.POINTER(C.c_int8), C.POINTER(C.c_int8), C.POINTER(C.c_int32),
                                               _vp, _vp, _vp, _vp] + _geom + [_i, _vp]),
    "tpg_fill_halo_regions_distributed_peers": (_i, [_vp, _i, _i, _i, C.POINTER(_vp), _i, C.POINTER(C.c_int8), C.POINTER(C.c_int8),
                                                     C.POINTER(C.c_int32), _vp, _vp, _vp, _vp] + _geom + [_i, _vp]),
    "tpg_nonorthogonality_angle": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "tpg_convert_frame": (_i, [_vp] * 8 + [_i] + _geom + [_i, _vp]),
}

_lib = None


def bind(path, signatures):
    """dlopen `path` and declare `signatures` on it (AttributeError if the ABI is incomplete)"""
    handle = C.CDLL(path)
    for name, (restype, argtypes) in signatures.items():
        fn = getattr(handle, name)
        fn.restype, fn.argtypes = restype, argtypes
    return handle


def lib():
    """Load libtripolar_hip.so; raise loudly if it has not been built (python __graft_entry__.py)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} not found: the HIP extension is the only backend of this package. "
                "Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C orthogonalsphericalshellgrids.jl_amd/csrc`).")
        _lib = bind(LIB_PATH, SIGNATURES)
    return _lib


def check(status):
    if status != 0:
        raise TripolarHipError(status, lib().tpg_last_error().decode("utf-8", "replace"))


def ft_of(dtype):
    import torch
    if dtype == torch.float64:
        return TPG_F64
    if dtype == torch.float32:
        return TPG_F32
    raise TypeError(f"unsupported element type {dtype}: Float32 or Float64 only")


def current_stream_ptr(device):
    """hipStream_t of torch's current stream on `device` (so torch.cuda.Event sees our kernels)."""
    import torch
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def ptr_table(tensors):
    return (C.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])
