#!/usr/bin/env python3
"""GPU box, under `rocprofv3 --kernel-trace`: k_cells_tile over thin latitude bands (21 rows each, 29 evaluated) swept from south to
north over the 3600 x 1800 grid, 30 builds per band -- which rows of the globe are expensive for the cell kernel?
usage: rocprofv3 --kernel-trace --output-format csv -d OUT -o rs -- python3 tools/row_cost_sweep.py
       python3 tools/row_cost_sweep.py --summarise OUT/rs_kernel_trace.csv"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
NX, NY, H, ROWS, REPS = 3600, 1800, 4, 21, 30
BANDS = [(j, min(NY, j + ROWS - 1)) for j in range(1, NY + 1, ROWS)]

if len(sys.argv) > 2 and sys.argv[1] == "--summarise":
    import csv, statistics
    rows = sorted((r for r in csv.DictReader(open(sys.argv[2])) if "k_cells_tile" in r["Kernel_Name"]), key=lambda r: int(r["Start_Timestamp"]))
    assert len(rows) == REPS * len(BANDS), (len(rows), len(BANDS))
    for b, (j0, j1) in enumerate(BANDS):
        d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows[b * REPS + 5:(b + 1) * REPS]]
        print("rows %4d..%4d  median %6.2f us  min %6.2f" % (j0, j1, statistics.median(d), min(d)))
    sys.exit(0)

import torch
import orthogonalsphericalshellgrids.jl_amd as osg
from orthogonalsphericalshellgrids.jl_amd import _lib
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
lib = _lib.lib()
arrs = [torch.empty((ROWS + 2 * H, NX + 2 * H), dtype=torch.float64, device=dev) for _ in _lib.ARRAY_NAMES]
ptrs = _lib.ptr_table(arrs)
for j0, j1 in BANDS:
    p = _lib.TpgParams(NX, NY, 1, H, H, H, -80.0, 55.0, 70.0, osg.R_Earth, _lib.TPG_F64, j0, j1, 0)
    ws = torch.empty(int(lib.tpg_build_grid_workspace_bytes(C.byref(p))), dtype=torch.uint8, device=dev)
    for _ in range(REPS):
        _lib.check(lib.tpg_build_grid(C.byref(p), ptrs, ws.data_ptr(), ws.numel(), None))
    torch.cuda.synchronize()
