#!/usr/bin/env python3
"""SURVEY.md 8f-1, BASELINE config 5: the halo fills of ONE baroclinic step of a hydrostatic model on the
1/24-degree, 100-level tripolar grid, single MI355X (the five 3-D fields alone are 162 GB of the 288 GB HBM).
The same fills are parity-tested against the oracle in tests/test_gpu_config5.py and timed in bench.py's `fill_step`.

Per step (examples/bickley_jet.jl:44-89 and test/runtests.jl:46-77 build this kind of model):
  * one tupled fill of the 3-D prognostic fields (u, v, T, S, c) -- ONE merged launch (fold + corners + periodic x);
  * `substeps` fills of the split-explicit free surface's 2-D fields (eta, U, V), which live on a grid whose north
    halo is extended to substeps + 1 rows (test/runtests.jl:61-71) -- one fused launch each, replayed from a HIP graph.
Prints the time of each part and what bounds it.  Run on an MI355X:  python examples/model_step_fills.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import orthogonalsphericalshellgrids.jl_amd as osg

SIZE = (8640, 4320, 100)
SUBSTEPS = 30


def timed(fn, reps):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps        # ms


def main():
    torch.cuda.set_device(0)
    grid = osg.TripolarGrid(size=SIZE)
    Nx, Ny, Nz = SIZE
    u, v = osg.XFaceField(grid), osg.YFaceField(grid)
    T, S, c = osg.CenterField(grid), osg.CenterField(grid), osg.CenterField(grid)
    ext = osg.with_halo((4, SUBSTEPS + 1, 4), osg.TripolarGrid(size=(Nx, Ny, 1)))      # the free surface's extended-halo grid
    eta = osg.Field((osg.Center, osg.Center, None), ext)
    U = osg.Field((osg.Face, osg.Center, None), ext)
    V = osg.Field((osg.Center, osg.Face, None), ext)
    for f in (u, v, T, S, c, eta, U, V):
        f.interior().uniform_(-1, 1)
    gb = sum(f.data.numel() * 8 for f in (u, v, T, S, c)) / 1e9

    plan3d = osg.halo_fill_plan((u, v, T, S, c))
    t3d = timed(plan3d, 10)
    zip_bytes = sum(Nx * Nz * 4 * 2 * 8 + (Nx // 2 * Nz * 2 * 8 if f.loc[1] is osg.Center else 0) for f in (u, v, T, S, c))
    rows = 5 * (Ny + 8) * (Nz + 8)
    graph2d = osg.halo_fill_plan((eta, U, V)).graph(repeat=SUBSTEPS)
    t2d = timed(graph2d.replay, 20)
    print(f"1/24 degree x {Nz} levels: 5 three-dimensional fields = {gb:.0f} GB")
    print(f"  tupled 3-D fill (zipper + periodic x): {t3d * 1e3:7.1f} us  (fold: {zip_bytes / 1e6:.0f} MB algorithmic; "
          f"periodic x: {rows / 1e6:.2f} M rows of two 64-B segments)")
    print(f"  {SUBSTEPS} sub-step fills of (eta, U, V), one replayed HIP graph: {t2d * 1e3:7.1f} us ({t2d / SUBSTEPS * 1e3:.1f} us per fill)")
    print(f"  halo fills per baroclinic step: {(t3d + t2d) * 1e3:7.1f} us")


if __name__ == "__main__":
    main()
