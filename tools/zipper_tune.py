"""GPU-box tuning harness for the zipper kernel at config 3 (3600x1800x75, 4 fields).
Run under rocprofv3 --kernel-trace (tools/zipper_tune.sh) so that durations are device timestamps:
for each variant the launch sequence is [cold-dirty, cold-clean, warm] x ROUNDS, recognisable in the
trace by order.  cold-dirty: Infinity Cache full of another kernel's dirty lines (1 GiB in-place add);
cold-clean: full of clean lines (1 GiB read-only reduction); warm: back-to-back relaunch."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from orthogonalsphericalshellgrids.jl_amd import _lib

NX, NY, NZ, H = 3600, 1800, 75, 4
SPECS = [(0, 0, 1), (1, 0, -1), (0, 1, -1), (1, 1, 1)]
ROUNDS = 10
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
lib = _lib.lib()
shape = (NZ + 2 * H, NY + 2 * H, NX + 2 * H)
fields = []
for fid in range(4):
    f = torch.empty(shape, dtype=torch.float64, device=dev)
    lib.tpg_fill_synthetic(f.data_ptr(), 0x5EED + fid, 12345.0, NX, NY, NZ, H, H, H, 1, None); fields.append(f)
fp = _lib.ptr_table(fields); n = 4
xl = (C.c_int8 * n)(*[s[0] for s in SPECS]); yl = (C.c_int8 * n)(*[s[1] for s in SPECS]); sg = (C.c_int32 * n)(*[s[2] for s in SPECS])
flush = torch.zeros(1 << 27, dtype=torch.float64, device=dev)     # 1 GiB
stream = _lib.current_stream_ptr(dev)
zip_ = lambda: lib.tpg_zipper_fill(fp, n, xl, yl, sg, NX, NY, NZ, H, H, H, 1, NZ, 1, stream)
for v in [int(a) for a in sys.argv[1:]]:
    os.environ["TPG_ZIPPER_VARIANT"] = str(v)
    for _ in range(ROUNDS):
        flush.add_(1.0); zip_()
        s = flush.sum(); zip_()
        zip_()
    torch.cuda.synchronize()
print("done")
