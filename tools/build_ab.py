#!/usr/bin/env python3
"""Interleaved A/B of tpg_build_grid between two settings of TPG_CELLS_VARIANT (test library): 2 = tile kernel + the halo pass k_halos
(K0 + K1 + K2: the product's build), 3 = k_cells_tile_push, the tile kernel writes the halo cells itself (K0 + K1; compiled into the test
library only: measured in round 6, not adopted -- profiles/r06/build_push_ab.txt).  Geometries: BASELINE config 2 (1/4 degree),
a 225-row band of config 4 (north, middle, south rank), the 1/10 degree globe.  usage: python tools/build_ab.py [rounds]"""
import ctypes as C
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from orthogonalsphericalshellgrids.jl_amd import _lib
from tools import testlib

torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
tl = testlib.lib()
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 12
GEOMS = {"config2_1440x720": (1440, 720, 1, 720), "band225_north": (3600, 1800, 1576, 1800), "band225_middle": (3600, 1800, 901, 1125),
         "band225_south": (3600, 1800, 1, 225), "globe_3600x1800": (3600, 1800, 1, 1800)}
out = {}
for name, (nx, ny, j0, j1) in GEOMS.items():
    H = 4
    p = _lib.TpgParams(nx, ny, 1, H, H, H, -80.0, 55.0, 70.0, 6371e3, 1, j0, j1, 0)
    rows = j1 - j0 + 1 + 2 * H
    arrs = [torch.empty((rows, nx + 2 * H), dtype=torch.float64, device=dev) for _ in range(20)]
    ptrs = _lib.ptr_table(arrs)
    ws = torch.empty(int(tl.tpg_build_grid_workspace_bytes(C.byref(p))), dtype=torch.uint8, device=dev)
    stream = _lib.current_stream_ptr(dev)
    reps = 40 if nx * (j1 - j0 + 1) > 2e6 else 100
    VARS = ["2", "3"]
    acc = {v: [] for v in VARS}
    ref = None
    for r in range(rounds + 2):
        for var in VARS if r % 2 == 0 else VARS[::-1]:
            os.environ["TPG_CELLS_VARIANT"] = var
            tl.tpg_reload_config()
            for _ in range(10):
                testlib.check(tl.tpg_build_grid(C.byref(p), ptrs, ws.data_ptr(), ws.numel(), stream))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                testlib.check(tl.tpg_build_grid(C.byref(p), ptrs, ws.data_ptr(), ws.numel(), stream))
            e1.record()
            torch.cuda.synchronize()
            if r >= 2:
                acc[var].append(e0.elapsed_time(e1) / reps * 1e3)
            if r == 0:                                         # both variants must write the same bits, halo cells included
                snap = [a.clone() for a in arrs]
                if ref is None:
                    ref = snap
                else:
                    assert all(torch.equal(a.view(torch.int64), b.view(torch.int64)) for a, b in zip(snap, ref)), name
                for a in arrs:
                    a.fill_(float("nan"))
    m2, m3 = statistics.median(acc["2"]), statistics.median(acc["3"])
    out[name] = {"tile_plus_k_halos_us": m2, "tile_pushes_halos_us": m3, "saved_us": m2 - m3, "rounds": rounds, "builds_per_round": reps,
                 "min_us": [min(acc["2"]), min(acc["3"])]}
    print(name, json.dumps(out[name]), flush=True)
print(json.dumps(out))
