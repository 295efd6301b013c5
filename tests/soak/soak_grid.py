"""GPU box: extended randomised soak of tpg_build_grid against the oracle (bit-exact, whole padded arrays):
sizes up to 400 x 120, continuous random poles / south / first-pole longitude / radius, both element types,
the tile kernel with and without its own halo writes, and the thread-per-cell kernel.  usage: python tests/soak/soak_grid.py [trials] [seed] [big]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import orthogonalsphericalshellgrids.jl_amd as osg
from oracle import oracle
from tools import testlib
testlib.active().__enter__()          # the TPG_* knobs live in the test library: route the package through it
trials = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for t in range(trials):
    big = len(sys.argv) > 3 and sys.argv[3] == "big"
    Nx = 2 * int(rng.integers(1, 900 if big else 200)); Ny = int(rng.integers(2, 500 if big else 121))
    Hx = int(rng.integers(1, min(Nx, 6) + 1)); Hy = int(rng.integers(1, min(Ny, 6) + 1))
    kw = dict(size=(Nx, Ny, 1), halo=(Hx, Hy, 1), north_poles_latitude=float(np.round(rng.uniform(20, 88), int(rng.integers(0, 6)))),
              first_pole_longitude=float(np.round(rng.uniform(-200, 380), int(rng.integers(0, 6)))),
              southernmost_latitude=float(np.round(rng.uniform(-88, 15), int(rng.integers(0, 6)))), radius=float(rng.choice([1.0, 6371e3, 3389.5e3])))
    dtype, tdt = ((np.float64, torch.float64), (np.float32, torch.float32))[t % 5 == 0]
    os.environ["TPG_CELLS_VARIANT"] = ("0", "3", "2", "3")[t % 4]; testlib.lib().tpg_reload_config()      # 3: the tile kernel writes the halo cells too (default)
    ref = oracle.build_grid(dtype=dtype, **kw)
    g = osg.TripolarGrid(osg.GPU(0), tdt, **kw)
    for name, r in ref.items():
        if not np.array_equal(getattr(g, name).cpu().numpy(), r, equal_nan=True):
            bad += 1; print("MISMATCH", t, kw, name); break
    if t % 4 == 1 and Ny >= 4:                      # a random latitude band of the same grid
        R = int(rng.integers(2, min(Ny, 9) + 1)); rk = int(rng.integers(0, R))
        band = osg.TripolarGrid(osg.Distributed(osg.GPU(0), osg.Partition(y=R), local_rank=rk), tdt, **kw)
        j0, j1 = band.jrange
        for name, r in ref.items():
            if not np.array_equal(getattr(band, name).cpu().numpy(), r[j0 - 1:j1 + 2 * Hy], equal_nan=True):
                bad += 1; print("BAND MISMATCH", t, kw, rk, R, name); break
    if t % 50 == 49: print(f"{t + 1} trials, {bad} mismatches", flush=True)
print("done:", trials, "trials,", bad, "mismatches")
sys.exit(1 if bad else 0)
