"""GPU box: zipper launch duration vs number of fields per launch (config-3 geometry), device
timestamps via tpg_zipper_fill_timed; cold = after the caches were flushed with a 1 GiB read."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from orthogonalsphericalshellgrids.jl_amd import _lib
from tools import testlib           # knobs, synthetic fill, copy probe: the test library (same kernels)
NX, NY, NZ, H = 3600, 1800, 75, 4
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
lib = testlib.lib()
shape = (NZ + 2 * H, NY + 2 * H, NX + 2 * H)
NF = 16
fields = []
for fid in range(NF):
    f = torch.empty(shape, dtype=torch.float64, device=dev)
    lib.tpg_fill_synthetic(f.data_ptr(), 0x5EED + fid, 12345.0, NX, NY, NZ, H, H, H, 1, None); fields.append(f)
specs = [(0, 0, 1), (1, 0, -1), (0, 1, -1), (1, 1, 1)] * 4
flush = torch.zeros(1 << 27, dtype=torch.float64, device=dev)
stream = _lib.current_stream_ptr(dev)
def ev():
    e = C.c_void_p(); lib.tpg_event_create(C.byref(e)); return e
for n in (1, 2, 4, 8, 16):
    fp = _lib.ptr_table(fields[:n])
    xl = (C.c_int8 * n)(*[s[0] for s in specs[:n]]); yl = (C.c_int8 * n)(*[s[1] for s in specs[:n]]); sg = (C.c_int32 * n)(*[s[2] for s in specs[:n]])
    nbytes = sum(17.28e6 if s[1] else 19.44e6 for s in specs[:n])
    for mode in ("cold", "warm"):
        ts = []
        for rep in range(12):
            if mode == "cold": flush.sum()
            e0, e1 = ev(), ev()
            assert lib.tpg_zipper_fill_timed(fp, n, xl, yl, sg, NX, NY, NZ, H, H, H, 1, NZ, 1, stream, e0, e1) == 0
            ms = C.c_float(); lib.tpg_event_elapsed_ms(e0, e1, C.byref(ms)); ts.append(ms.value * 1e3)
        ts = sorted(ts[2:]); med = ts[len(ts) // 2]
        print(f"fields={n:2d} {mode}: median {med:7.2f} us -> {nbytes / med / 1e3:6.0f} GB/s = {nbytes / med / 1e3 / 80:5.1f}% of 8 TB/s")
