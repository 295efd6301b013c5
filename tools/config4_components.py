#!/usr/bin/env python3
"""GPU box: the device-local components of ONE rank's step in BASELINE config 4 (the 3600 x 1800 x 75 globe in 8 latitude bands of
225 rows), measured on one GPU: band build, band halo fill (north band with the zipper; a middle band: periodic x only), and the
pack / unpack kernels of the seam messages (4 fields x 9.58 MB per side).  What is NOT here is the link: the RCCL send/recv of the
packed messages needs a second GPU.  Prints one JSON line.  usage: python tools/config4_components.py [N]"""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import orthogonalsphericalshellgrids.jl_amd as osg
from orthogonalsphericalshellgrids.jl_amd import _lib
from tools import testlib

NX, NY, NZ, H = 3600, 1800, 75, 4
R = int(sys.argv[1]) if len(sys.argv) > 1 else 8
ny = NY // R
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
lib, tlib = _lib.lib(), testlib.lib()
st = _lib.current_stream_ptr(dev)

def timed(fn, reps=100, warm=60):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3            # us

out = {"config": f"3600x1800x75 in {R} bands of {ny} rows, Float64, halo 4, fields c/u/v/zeta", "unit": "us"}
# Whatever FP64-heavy kernel runs FIRST sits in the power-management transient that follows its onset (~40 launches of the 1/10
# degree build, ~25 ms): round 3 and the first runs of round 4 measured the north band first and read the transient as "the north band is 8 us
# slower".  So: a common warm-up well past the transient before anything is timed, and the three band builds measured again at the end
# (`*_build_late`) to show the order no longer matters.
_pw = _lib.TpgParams(NX, NY, NZ, H, H, H, -80.0, 55.0, 70.0, osg.R_Earth, _lib.TPG_F64, 1 + ny * (R // 2), ny * (R // 2 + 1), 0)
_aw = [torch.empty((ny + 2 * H, NX + 2 * H), dtype=torch.float64, device=dev) for _ in _lib.ARRAY_NAMES]
_ww = torch.empty(int(lib.tpg_build_grid_workspace_bytes(C.byref(_pw))), dtype=torch.uint8, device=dev)
for _ in range(1500):
    _lib.check(lib.tpg_build_grid(C.byref(_pw), _lib.ptr_table(_aw), _ww.data_ptr(), _ww.numel(), st))
torch.cuda.synchronize()
late = []
for label, rank in (("north_band", R - 1), ("middle_band", R // 2), ("south_band", 0)):
    jstart, jend = 1 + ny * rank, ny * (rank + 1)
    p = _lib.TpgParams(NX, NY, NZ, H, H, H, -80.0, 55.0, 70.0, osg.R_Earth, _lib.TPG_F64, jstart, jend, 0)
    arrs = [torch.empty((ny + 2 * H, NX + 2 * H), dtype=torch.float64, device=dev) for _ in _lib.ARRAY_NAMES]
    ptrs = _lib.ptr_table(arrs)
    ws = torch.empty(int(lib.tpg_build_grid_workspace_bytes(C.byref(p))), dtype=torch.uint8, device=dev)
    out[label + "_build"] = timed(lambda: _lib.check(lib.tpg_build_grid(C.byref(p), ptrs, ws.data_ptr(), ws.numel(), st)))
    late.append((label, p, ptrs, ws, arrs))
    pv = _lib.TpgParams(NX, NY, NZ, H, H, H, -80.0, 55.0, 70.0, osg.R_Earth, _lib.TPG_F64, jstart, jend, 1)     # TPG_BUILD_TABLES_VALID
    out[label + "_build_tables_cached"] = timed(lambda: _lib.check(lib.tpg_build_grid(C.byref(pv), ptrs, ws.data_ptr(), ws.numel(), st)))
    fields = [torch.empty((NZ + 2 * H, ny + 2 * H, NX + 2 * H), dtype=torch.float64, device=dev) for _ in range(4)]
    for k, f in enumerate(fields):
        testlib.check(tlib.tpg_fill_synthetic(f.data_ptr(), 0xC4 + k, 12345.0, NX, ny, NZ, H, H, H, _lib.TPG_F64, None))
    fp = _lib.ptr_table(fields)
    xl = (C.c_int8 * 4)(0, 1, 0, 1); yl = (C.c_int8 * 4)(0, 0, 1, 1); sg = (C.c_int32 * 4)(1, -1, -1, 1)
    zip_ = 1 if rank == R - 1 else 0
    out[label + "_local_fill"] = timed(lambda: _lib.check(lib.tpg_fill_halo_regions(fp, 4, xl, yl, sg, NX, ny, NZ, H, H, H, zip_, _lib.TPG_F64, st)))
    n = int(lib.tpg_y_halo_buffer_elems(4, NX, NZ, H, H, H))
    buf = torch.empty(n, dtype=torch.float64, device=dev)
    out[label + "_pack_one_side"] = timed(lambda: _lib.check(lib.tpg_pack_y_halo(fp, 4, buf.data_ptr(), 0, NX, ny, NZ, H, H, H, _lib.TPG_F64, st)))
    out[label + "_unpack_one_side"] = timed(lambda: _lib.check(lib.tpg_unpack_y_halo(fp, 4, buf.data_ptr(), 0, NX, ny, NZ, H, H, H, _lib.TPG_F64, st)))
    out["seam_message_MB"] = n * 8 / 1e6
    del fields, buf
for label, p, ptrs, ws, arrs in reversed(late):
    out[label + "_build_late"] = timed(lambda: _lib.check(lib.tpg_build_grid(C.byref(p), ptrs, ws.data_ptr(), ws.numel(), st)))
link_us = out["seam_message_MB"] * 1e6 / 153e9 * 1e6
out["link_floor_one_direction"] = link_us
out["note"] = ("a middle rank's step = max(build, local fill + 2 packs + [>= link floor: 38.3 MB on one ~153 GB/s xGMI link per direction, the two "
               "seams use different links] + 2 unpacks); the link term is a spec figure, not a measurement")
print(json.dumps(out))
