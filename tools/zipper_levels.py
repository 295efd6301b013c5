"""GPU box: zipper launch duration vs number of levels folded (4 fields, config-3 geometry): does the
launch lose time when its waves exceed one resident round (8 waves/SIMD x 1024 SIMDs = 8192)?
75 levels = 8448 waves; 72 levels = 8112 waves."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from orthogonalsphericalshellgrids.jl_amd import _lib
NX, NY, NZ, H = 3600, 1800, 75, 4
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
lib = _lib.lib()
shape = (NZ + 2 * H, NY + 2 * H, NX + 2 * H)
fields = []
for fid in range(4):
    f = torch.empty(shape, dtype=torch.float64, device=dev)
    lib.tpg_fill_synthetic(f.data_ptr(), 0x5EED + fid, 12345.0, NX, NY, NZ, H, H, H, 1, None); fields.append(f)
specs = [(0, 0, 1), (1, 0, -1), (0, 1, -1), (1, 1, 1)]
flush = torch.zeros(1 << 27, dtype=torch.float64, device=dev)
stream = _lib.current_stream_ptr(dev)
fp = _lib.ptr_table(fields)
xl = (C.c_int8 * 4)(*[s[0] for s in specs]); yl = (C.c_int8 * 4)(*[s[1] for s in specs]); sg = (C.c_int32 * 4)(*[s[2] for s in specs])
def ev():
    e = C.c_void_p(); lib.tpg_event_create(C.byref(e)); return e
for kcount in (75, 72, 70, 64, 60, 75, 72):
    nbytes = 73.44e6 * kcount / 75
    waves = 4 * ((1800 * kcount + 255) // 256) * 4
    ts = []
    for rep in range(14):
        flush.sum()
        e0, e1 = ev(), ev()
        assert lib.tpg_zipper_fill_timed(fp, 4, xl, yl, sg, NX, NY, NZ, H, H, H, 1, kcount, 1, stream, e0, e1) == 0
        ms = C.c_float(); lib.tpg_event_elapsed_ms(e0, e1, C.byref(ms)); ts.append(ms.value * 1e3)
    ts = sorted(ts[2:]); med = ts[len(ts) // 2]
    print(f"levels={kcount:3d} waves={waves:5d} cold: median {med:6.2f} us (min {ts[0]:6.2f}) -> {nbytes / med / 1e3:6.0f} GB/s = {nbytes / med / 1e3 / 80:5.1f}% of 8 TB/s")
