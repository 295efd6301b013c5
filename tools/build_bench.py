"""GPU box: tpg_build_grid throughput at the BASELINE grid sizes (events around 20 back-to-back builds)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import orthogonalsphericalshellgrids.jl_amd as osg
from orthogonalsphericalshellgrids.jl_amd import _lib
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
lib = _lib.lib()
for name, (nx, ny), dt in (("1/4 deg f64", (1440, 720), torch.float64), ("1/10 deg f64", (3600, 1800), torch.float64),
                           ("1/10 deg f32", (3600, 1800), torch.float32), ("1/24 deg f64", (8640, 4320), torch.float64)):
    p = _lib.TpgParams(nx, ny, 1, 4, 4, 4, -80.0, 55.0, 70.0, osg.R_Earth, _lib.ft_of(dt), 1, ny, 0)
    out = [torch.empty((ny + 8, nx + 8), dtype=dt, device=dev) for _ in _lib.ARRAY_NAMES]
    ptrs = _lib.ptr_table(out)
    ws = torch.empty(int(lib.tpg_build_grid_workspace_bytes(C.byref(p))), dtype=torch.uint8, device=dev)
    st = _lib.current_stream_ptr(dev)
    for _ in range(3): _lib.check(lib.tpg_build_grid(C.byref(p), ptrs, ws.data_ptr(), ws.numel(), st))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    R = 20
    e0.record()
    for _ in range(R): _lib.check(lib.tpg_build_grid(C.byref(p), ptrs, ws.data_ptr(), ws.numel(), st))
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / R
    print(f"{name:14s} {nx}x{ny}: {ms * 1e3:8.1f} us per build -> {nx * ny / ms * 1e3:.3e} cells/s; stores {20 * (nx + 8) * (ny + 8) * out[0].element_size() / ms / 1e6:.0f} GB/s")
    del out
