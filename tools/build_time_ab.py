#!/usr/bin/env python3
"""GPU box: in-process alternating A/B of tpg_build_grid (3600 x 1800 Float64) between two builds of the library, e.g.
tools/ab/libtripolar_hip_r02.so (round-2 kernels, `git archive 51dc38d` + make) and the current one.  Both are loaded side by side
with ctypes; after a common warm-up (the power-management transient of the cell kernel) blocks of 40 builds alternate.
usage: python tools/build_time_ab.py <libA.so> <libB.so> [alternations]"""
import ctypes as C, os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from orthogonalsphericalshellgrids.jl_amd import _lib

def load(path):
    h = C.CDLL(path)
    h.tpg_build_grid_workspace_bytes.restype = C.c_size_t
    h.tpg_build_grid_workspace_bytes.argtypes = [C.POINTER(_lib.TpgParams)]
    h.tpg_build_grid.restype = C.c_int
    h.tpg_build_grid.argtypes = [C.POINTER(_lib.TpgParams), C.POINTER(C.c_void_p), C.c_void_p, C.c_size_t, C.c_void_p]
    return h

libs = [load(sys.argv[1]), load(sys.argv[2])]
alts = int(sys.argv[3]) if len(sys.argv) > 3 else 8
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
nx, ny = 3600, 1800
p = _lib.TpgParams(nx, ny, 1, 4, 4, 4, -80.0, 55.0, 70.0, 6371e3, 1, 1, ny, 0)
outs = [[torch.empty((ny + 8, nx + 8), dtype=torch.float64, device=dev) for _ in _lib.ARRAY_NAMES] for _ in libs]
ptrs = [_lib.ptr_table(o) for o in outs]
ws = torch.empty(int(max(l.tpg_build_grid_workspace_bytes(C.byref(p)) for l in libs)), dtype=torch.uint8, device=dev)
st = _lib.current_stream_ptr(dev)
def run(k, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        assert libs[k].tpg_build_grid(C.byref(p), ptrs[k], ws.data_ptr(), ws.numel(), st) == 0
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
run(0, 60); run(1, 60)                                  # past the transient
t = [[], []]
for a in range(alts):
    for k in (0, 1):
        t[k].append(run(k, 40))
same = all(torch.equal(x, y) for x, y in zip(outs[0], outs[1]))
for k in (0, 1):
    print(f"{os.path.basename(sys.argv[1 + k]):34s} median {statistics.median(t[k]):7.1f} us per build   (blocks: {' '.join(f'{x:.0f}' for x in t[k])})")
print(f"B / A = {statistics.median(t[1]) / statistics.median(t[0]):.4f}; the 20 arrays of the two builds are bit-identical: {same}")
