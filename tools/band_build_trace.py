#!/usr/bin/env python3
"""GPU box, under `rocprofv3 --kernel-trace`: 60 builds each of the north, a middle and the south band of BASELINE config 4
(8 bands of 225 rows of the 3600 x 1800 grid), so that the trace shows what each band's launches cost kernel by kernel.
usage: rocprofv3 --kernel-trace --output-format csv -d OUT -o bb -- python3 tools/band_build_trace.py ; then summarise with
       python3 tools/band_build_trace.py --summarise OUT/bb_kernel_trace.csv"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if len(sys.argv) > 2 and sys.argv[1] == "--summarise":
    import csv, re, statistics
    rows = sorted(csv.DictReader(open(sys.argv[2])), key=lambda r: int(r["Start_Timestamp"]))
    rows = [r for r in rows if re.search(r"k_(tables|cells_tile|halos|south)", r["Kernel_Name"])]
    builds, cur = [], []
    for r in rows:
        if "k_tables" in r["Kernel_Name"] and cur:
            builds.append(cur); cur = []
        cur.append(r)
    builds.append(cur)
    # the launch sequence below: south, middle, north, then north, middle, south again -- whatever runs FIRST sits in the power-management
    # transient that follows the onset of this FP64-heavy kernel, so every band is measured early and late
    per = len(builds) // 6
    for k, label in enumerate(("south (1st)", "middle (2nd)", "north (3rd)", "north (4th)", "middle (5th)", "south (6th)")):
        bs = builds[k * per:(k + 1) * per]
        bs = bs[10:]                                        # warm-up
        names = [re.search(r"(k_[a-z_]+)", r["Kernel_Name"]).group(1) for r in bs[0]]
        durs = {n: statistics.median((int(b[i]["End_Timestamp"]) - int(b[i]["Start_Timestamp"])) / 1e3 for b in bs) for i, n in enumerate(names)}
        gaps = [statistics.median((int(b[i + 1]["Start_Timestamp"]) - int(b[i]["End_Timestamp"])) / 1e3 for b in bs) for i in range(len(names) - 1)]
        span = statistics.median((int(b[-1]["End_Timestamp"]) - int(b[0]["Start_Timestamp"])) / 1e3 for b in bs)
        print(label, {k: round(v, 2) for k, v in durs.items()}, "gaps", [round(g, 2) for g in gaps], "first start -> last end", round(span, 2))
    sys.exit(0)

import torch
import orthogonalsphericalshellgrids.jl_amd as osg
from orthogonalsphericalshellgrids.jl_amd import _lib
NX, NY, NZ, H, R = 3600, 1800, 75, 4, 8
ny = NY // R
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
lib = _lib.lib()
for rank in (0, R // 2, R - 1, R - 1, R // 2, 0):
    jstart, jend = 1 + ny * rank, ny * (rank + 1)
    p = _lib.TpgParams(NX, NY, NZ, H, H, H, -80.0, 55.0, 70.0, osg.R_Earth, _lib.TPG_F64, jstart, jend, 0)
    arrs = [torch.empty((ny + 2 * H, NX + 2 * H), dtype=torch.float64, device=dev) for _ in _lib.ARRAY_NAMES]
    ptrs = _lib.ptr_table(arrs)
    ws = torch.empty(int(lib.tpg_build_grid_workspace_bytes(C.byref(p))), dtype=torch.uint8, device=dev)
    for _ in range(60):
        _lib.check(lib.tpg_build_grid(C.byref(p), ptrs, ws.data_ptr(), ws.numel(), None))
    torch.cuda.synchronize()
