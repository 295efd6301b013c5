#!/usr/bin/env python3
"""GPU box: interleaved A/B of the whole halo fill (tpg_fill_halo_regions) of large 3-D fields: two launches (zipper, then
periodic x) vs the merged single launch (TPG_FILL_MERGED).  Config 3 (4 fields, 3600x1800x75) and config 5 (5 fields,
8640x4320x100).  Times are hipEvent brackets around `reps` back-to-back fills, and single fills after a 1 GiB flush.
usage: python tools/fill_ab.py [rounds]"""
import ctypes as C, os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from orthogonalsphericalshellgrids.jl_amd import _lib
from tools import testlib           # knobs, synthetic fill, copy probe: the test library (same kernels)

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 12
torch.cuda.set_device(0); dev = torch.device("cuda", 0)
lib = testlib.lib()
flush = torch.zeros(1 << 27, dtype=torch.float64, device=dev)
ev = lambda: torch.cuda.Event(enable_timing=True)

for label, (NX, NY, NZ), specs in (("config 3", (3600, 1800, 75), [(0, 0, 1), (1, 0, -1), (0, 1, -1), (1, 1, 1)]),
                                   ("config 5", (8640, 4320, 100), [(1, 0, -1), (0, 1, -1), (0, 0, 1), (0, 0, 1), (0, 0, 1)])):
    H = 4
    fields = []
    for fid in range(len(specs)):
        f = torch.empty((NZ + 2 * H, NY + 2 * H, NX + 2 * H), dtype=torch.float64, device=dev)
        _lib.check(lib.tpg_fill_synthetic(f.data_ptr(), 0xAB + fid, 12345.0, NX, NY, NZ, H, H, H, 1, None)); fields.append(f)
    n = len(specs)
    fp = _lib.ptr_table(fields)
    xl = (C.c_int8 * n)(*[s[0] for s in specs]); yl = (C.c_int8 * n)(*[s[1] for s in specs]); sg = (C.c_int32 * n)(*[s[2] for s in specs])
    stream = _lib.current_stream_ptr(dev)
    fill = lambda: _lib.check(lib.tpg_fill_halo_regions(fp, n, xl, yl, sg, NX, NY, NZ, H, H, H, 1, 1, stream))
    res = {("0", "b2b"): [], ("1", "b2b"): [], ("0", "cold"): [], ("1", "cold"): []}
    for r in range(rounds + 2):
        for mode in ("0", "1"):
            os.environ["TPG_FILL_MERGED"] = mode; lib.tpg_reload_config()
            fill(); torch.cuda.synchronize()
            e0, e1 = ev(), ev()
            e0.record()
            for _ in range(10): fill()
            e1.record(); torch.cuda.synchronize()
            if r >= 2: res[(mode, "b2b")].append(e0.elapsed_time(e1) / 10 * 1e3)
            flush.add_(1.0); flush.sum()
            e0, e1 = ev(), ev()
            e0.record(); fill(); e1.record(); torch.cuda.synchronize()
            if r >= 2: res[(mode, "cold")].append(e0.elapsed_time(e1) * 1e3)
    for k, v in res.items():
        print(f"{label}: merged={k[0]} {k[1]:5s} median {statistics.median(v):8.2f} us  min {min(v):8.2f}")
    del fields
    torch.cuda.empty_cache()
os.environ.pop("TPG_FILL_MERGED", None)
print("done")
