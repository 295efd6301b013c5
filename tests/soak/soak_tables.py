"""GPU box: randomised soak of the TABLE-WORKSPACE reuse of the Python host (grids.py: TableWorkspace, TPG_BUILD_TABLES_VALID) against the
oracle, bit-exact on whole padded arrays.  A small pool of geometries is visited in random order by random operations -- fresh builds, random
latitude bands, with_halo (new Hx / Hz, sometimes a new Hy), reconstruct_global_grid, builds on a side stream -- while a random subset of the
grids stays alive, so that table keys repeat, workspaces are shared, dropped and re-created.  Every build is compared with the oracle; every
build also checks that the flag was set exactly when it should be: inside a `share_tables()` scope (half of the trials) when a live workspace
of its key existed, outside it only for the explicit hand-overs (with_halo of the same key, reconstruct_global_grid).  usage: python tests/soak/soak_tables.py [trials] [seed]"""
import contextlib, gc, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import orthogonalsphericalshellgrids.jl_amd as osg
from orthogonalsphericalshellgrids.jl_amd import grids
from oracle import oracle

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
torch.cuda.set_device(0)
POOL = [dict(size=(2 * int(rng.integers(4, 120)), int(rng.integers(8, 90)), 1), north_poles_latitude=float(rng.choice([55, 60.5, 35])),
             southernmost_latitude=float(rng.choice([-80, -75.5])), radius=float(rng.choice([1.0, 6371e3]))) for _ in range(6)]
alive, bad, reused, fresh = [], 0, 0, 0
side = torch.cuda.Stream()
ref_cache = {}


def ref_of(kw, halo, dtype, fpl):
    key = (kw["size"], kw["north_poles_latitude"], kw["southernmost_latitude"], kw["radius"], halo, dtype, fpl)
    if key not in ref_cache:
        ref_cache[key] = oracle.build_grid(dtype=dtype, halo=halo, first_pole_longitude=fpl, **kw)
    return ref_cache[key]


def check(g, kw, halo, dtype, fpl, what, band=None):
    global bad
    ref = ref_of(kw, halo, dtype, fpl)
    for name, r in ref.items():
        want = r if band is None else r[band[0] - 1:band[1] + 2 * halo[1]]
        if not np.array_equal(getattr(g, name).cpu().numpy(), want, equal_nan=True):
            bad += 1; print("MISMATCH", what, kw, halo, dtype, fpl, band, name, "reused" if g.tables_reused else "fresh", flush=True); return


for t in range(trials):
    kw = POOL[int(rng.integers(0, len(POOL)))]
    Nx, Ny, _ = kw["size"]
    halo = (int(rng.integers(1, min(Nx, 5) + 1)), int(rng.choice([2, 3, 4])) if Ny >= 8 else 2, int(rng.integers(0, 3)))
    dtype, tdt = ((np.float64, torch.float64), (np.float32, torch.float32))[int(rng.integers(0, 4)) == 0]
    fpl = float(rng.choice([70.0, 75.0, -12.25]))
    key = grids.table_key(Nx, Ny, halo[1], tdt, kw["southernmost_latitude"], kw["north_poles_latitude"], kw["radius"], torch.device("cuda", 0))
    shared = bool(rng.integers(0, 2))                                    # implicit sharing is explicit since round 6: only inside the scope
    expect = shared and any(g.workspace.key == key for g in alive)
    op = int(rng.integers(0, 5))
    ctx = torch.cuda.stream(side) if op == 4 else torch.cuda.stream(torch.cuda.current_stream())
    with ctx, (osg.share_tables() if shared else contextlib.nullcontext()):
        if op in (0, 4):                                                # a fresh serial build (op 4: on the side stream)
            g = osg.TripolarGrid(osg.GPU(0), tdt, halo=halo, first_pole_longitude=fpl, **kw)
            check(g, kw, halo, dtype, fpl, "build")
        elif op == 1:                                                   # a random latitude band
            R = int(rng.integers(2, min(Ny // 2, 8) + 1)); rk = int(rng.integers(0, R))
            g = osg.TripolarGrid(osg.Distributed(osg.GPU(0), osg.Partition(y=R), local_rank=rk), tdt, halo=halo, first_pole_longitude=fpl, **kw)
            check(g, kw, halo, dtype, fpl, "band", g.jrange)
            if rng.integers(0, 2):                                      # ... and the globe from it: always finds the band's tables
                full = osg.reconstruct_global_grid(g)
                # reconstruct_global_grid does not forward radius (the serial constructor's default applies): compare on that basis
                kw_full = dict(kw, radius=osg.R_Earth)
                check(full, kw_full, halo, dtype, fpl, "reconstruct")
                if kw["radius"] == osg.R_Earth and not full.tables_reused:
                    bad += 1; print("NOT REUSED: reconstruct_global_grid", kw, flush=True)
        else:                                                           # with_halo of a live grid of this geometry, if any
            olds = [g for g in alive if g.size == kw["size"] and g.global_size is None and g.dtype == tdt
                    and g.conformal_mapping.north_poles_latitude == kw["north_poles_latitude"] and g.radius == kw["radius"]
                    and g.conformal_mapping.southernmost_latitude == kw["southernmost_latitude"]]
            if not olds:
                olds = None
                continue
            old = olds[int(rng.integers(0, len(olds)))]
            g = osg.with_halo(halo, old)
            check(g, kw, halo, dtype, float(old.conformal_mapping.first_pole_longitude), "with_halo")
            expect = expect or old.workspace.key == key
    full = old = olds = None                                             # no grid may outlive the `alive` list unseen: it would keep its tables alive
    if g.tables_reused != expect:
        bad += 1; print("FLAG", "reused" if g.tables_reused else "fresh", "expected", expect, kw, halo, dtype, flush=True)
    reused += int(g.tables_reused); fresh += int(not g.tables_reused)
    alive.append(g)
    while len(alive) > 6 or (alive and rng.integers(0, 3) == 0):         # let grids (and with them workspaces) go
        alive.pop(int(rng.integers(0, len(alive))))
    del g
    if t % 7 == 0:
        gc.collect()
    if t % 50 == 49:
        print(f"{t + 1} trials, {bad} mismatches, {reused} builds reused tables, {fresh} computed them", flush=True)
print("done:", trials, "trials,", bad, "mismatches,", reused, "reused,", fresh, "fresh")
sys.exit(1 if bad else 0)
