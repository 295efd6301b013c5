#!/usr/bin/env python3
"""Next-row sketch: the halo fills of a split-explicit barotropic sub-cycle.

Oceananigans' SplitExplicitFreeSurface advances eta, U, V (2-D, reduced in z) for `substeps`
sub-steps per baroclinic step and fills their halos every sub-step (test/runtests.jl:52-76 builds
exactly that model on a TripolarGrid; eta carries an extended north halo).  Each fill is tiny
(3 fields x Hy rows x Nx), so the loop is launch-bound: this script times the same sequence issued
eagerly and replayed from ONE captured HIP graph.  Run on an MI355X:  python examples/barotropic_substep_fills.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import orthogonalsphericalshellgrids.jl_amd as osg

SUBSTEPS = 30


def main():
    torch.cuda.set_device(0)
    grid = osg.TripolarGrid(size=(3600, 1800, 1))
    # eta: (Center, Center, Nothing); U: (Face, Center, Nothing); V: (Center, Face, Nothing)
    eta = osg.Field((osg.Center, osg.Center, None), grid)
    U = osg.Field((osg.Face, osg.Center, None), grid)
    V = osg.Field((osg.Center, osg.Face, None), grid)
    for f in (eta, U, V):
        f.interior().uniform_(-1, 1)

    def cycle():
        for _ in range(SUBSTEPS):
            osg.fill_halo_regions((eta, U, V))       # ONE fused launch (TPG_FILL_FUSED=0: zipper launch + periodic launch)

    cycle(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record()
    for _ in range(reps):
        cycle()
    e1.record(); torch.cuda.synchronize()
    eager = e0.elapsed_time(e1) / reps

    plan = osg.halo_fill_plan((eta, U, V))           # argument tables built once: one C call per fill
    plan(); torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        for _ in range(SUBSTEPS):
            plan()
    e1.record(); torch.cuda.synchronize()
    planned = e0.elapsed_time(e1) / reps

    graph = plan.graph(repeat=SUBSTEPS)              # the whole sub-cycle as one HIP graph
    graph.replay(); torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        graph.replay()
    e1.record(); torch.cuda.synchronize()
    replay = e0.elapsed_time(e1) / reps
    print(f"{SUBSTEPS} sub-step fills of (eta, U, V) on 3600x1800: eager {eager * 1e3:.0f} us "
          f"({eager / SUBSTEPS * 1e3:.1f} us per fill), with a HaloFillPlan {planned * 1e3:.0f} us "
          f"({planned / SUBSTEPS * 1e3:.1f} us per fill), graph replay {replay * 1e3:.0f} us "
          f"({replay / SUBSTEPS * 1e3:.1f} us per fill)")


if __name__ == "__main__":
    main()
