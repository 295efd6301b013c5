"""GPU box: randomised soak of the latitude-band halo fill, R ranks emulated in one process with a loop-back
transport (real kernels: zipper on the north rank, periodic x, pack / unpack).  Every rank's slab must equal
rows jstart-Hy..jend+Hy of the serially filled global field (oracle).  usage: soak_distributed.py [trials] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import orthogonalsphericalshellgrids.jl_amd as osg
from oracle import oracle
trials = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
torch.cuda.set_device(0)
lib = osg._lib.lib()
SENT = 12345.0
bad = 0
for t in range(trials):
    R = int(rng.integers(2, 7))
    Hx, Hy, Hz = int(rng.integers(1, 6)), int(rng.integers(1, 6)), int(rng.integers(0, 3))
    # local rows > halo: with ny == Hy the north rank's y-Center fold reads its own south HALO row, which the
    # reference's order (zipper -> periodic x -> communication) has not received yet either
    ny = int(rng.integers(Hy + 1, Hy + 12)); Ny = ny * R
    Nx = 2 * int(rng.integers(max(1, Hx), 60)); Nz = int(rng.integers(1, 4))
    size, halo = (Nx, Ny, Nz), (Hx, Hy, Hz)
    dtype, tdt = ((np.float64, torch.float64), (np.float32, torch.float32))[t % 2]
    nf = int(rng.integers(1, 5))
    specs = [(int(rng.integers(0, 2)), int(rng.integers(0, 2))) for _ in range(nf)]
    globs = []
    for _ in specs:
        g = rng.uniform(-1, 1, (Nz + 2 * Hz, Ny + 2 * Hy, Nx + 2 * Hx)).astype(dtype)
        g[:, :Hy] = SENT; g[:, Hy + Ny:] = SENT
        globs.append(g)
    ranks = []
    for r in range(R):
        arch = osg.Distributed(osg.GPU(0), osg.Partition(y=R), local_rank=r)
        grid = osg.TripolarGrid(arch, tdt, size=size, halo=halo)
        jstart, jend = grid.jrange
        fs = []
        for (xl, yl), g in zip(specs, globs):
            f = osg.Field((osg.Face if xl else osg.Center, osg.Face if yl else osg.Center, osg.Center), grid)
            slab = g[:, jstart - 1:jend + 2 * Hy].copy()
            slab[:, :Hy] = SENT; slab[:, Hy + (jend - jstart + 1):] = SENT
            f.data.copy_(torch.from_numpy(slab)); fs.append(f)
        ranks.append((arch, grid, fs))
    mailbox = osg.LoopbackMailbox()
    plans = [osg.halo_fill_plan(fs, exchange=mailbox.endpoint(r)) for r, (arch, grid, fs) in enumerate(ranks)]
    for plan in plans: plan.begin()
    for plan in plans: plan.finish()
    torch.cuda.synchronize()
    for (xl, yl), g in zip(specs, globs):
        oracle.fill_halo_regions(g, xl, yl, -1 if xl != yl else 1, size, halo)      # default sign by location
    ok = True
    for r, (arch, grid, fs) in enumerate(ranks):
        jstart, jend = grid.jrange
        for f, g in zip(fs, globs):
            if not np.array_equal(f.data.cpu().numpy(), g[:, jstart - 1:jend + 2 * Hy]):
                ok = False; print("MISMATCH", t, size, halo, "R", R, "rank", r, f.loc, dtype.__name__); break
        if not ok: break
    bad += 0 if ok else 1
    if t % 100 == 99: print(f"{t + 1} trials, {bad} bad", flush=True)
print("done:", trials, "trials,", bad, "bad")
sys.exit(1 if bad else 0)
