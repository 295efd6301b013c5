"""GPU box: randomised soak of the halo fill (tpg_fill_halo_regions: zipper -> periodic x) against the oracle,
bit-exact on whole padded arrays: random geometry, locations, signs, element types, zipper variants, fused and
two-launch forms.  usage: python tests/soak/soak_fill.py [trials] [seed]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import orthogonalsphericalshellgrids.jl_amd as osg
from orthogonalsphericalshellgrids.jl_amd import _lib
from tools import testlib           # knobs, synthetic fill, copy probe: the test library (same kernels)
from oracle import oracle
trials = int(sys.argv[1]) if len(sys.argv) > 1 else 500
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
lib = testlib.lib()
bad = 0
for t in range(trials):
    Nx = 2 * int(rng.integers(1, 150)); Ny = int(rng.integers(1, 40)); Nz = int(rng.integers(1, 5))
    if t % 40 == 7: Nx = int(rng.choice([1440, 3600, 8640, 4322])); Ny = int(rng.integers(10, 30))      # full-width rows of the BASELINE grids
    Hx = int(rng.integers(0, min(Nx, 7) + 1)); Hy = int(rng.integers(0, min(Ny, 10) + 1)); Hz = int(rng.integers(0, 3))
    nf = int(rng.integers(1, 6))
    dt, tdt, ft = ((np.float64, torch.float64, 1), (np.float32, torch.float32, 0))[t % 3 == 0]
    os.environ["TPG_ZIPPER_VARIANT"] = str(int(rng.choice([0, 3])))
    mode = t % 4                                             # automatic / never fused / fused chunk items / fused one thread per cell
    if mode == 0: os.environ.pop("TPG_FILL_FUSED", None)
    else: os.environ["TPG_FILL_FUSED"] = str(mode - 1)
    os.environ["TPG_FILL_MERGED"] = str(int(rng.integers(0, 2)))
    lib.tpg_reload_config()                                  # the library reads its knobs once
    specs = [(int(rng.integers(0, 2)), int(rng.integers(0, 2)), int(rng.choice([1, -1, 3]))) for _ in range(nf)]
    if t % 11 == 5: nf = int(rng.integers(17, 40)); specs = [(int(rng.integers(0, 2)), int(rng.integers(0, 2)), int(rng.choice([1, -1]))) for _ in range(nf)]   # > TPG_MAX_FIELDS
    shape = (Nz + 2 * Hz, Ny + 2 * Hy, Nx + 2 * Hx)
    hosts = [rng.uniform(-1, 1, shape).astype(dt) for _ in specs]
    if t % 7 == 3:        # base pointers that are only element-aligned: views one element into a larger buffer
        n = int(np.prod(shape)); devs = []
        for h in hosts:
            buf = torch.empty(n + 1, dtype=tdt, device=dev); v = buf[1:].view(shape); v.copy_(torch.from_numpy(h)); devs.append(v)
    else:
        devs = [torch.from_numpy(h).to(dev) for h in hosts]
    xl = (C.c_int8 * nf)(*[s[0] for s in specs]); yl = (C.c_int8 * nf)(*[s[1] for s in specs]); sg = (C.c_int32 * nf)(*[s[2] for s in specs])
    zip_only = t % 5 == 4
    if zip_only:          # the fold alone, on a random range of levels (halo levels included)
        kstart = int(rng.integers(1 - Hz, Nz + 1)); kcount = int(rng.integers(0, Nz + Hz - kstart + 2))
        rc = lib.tpg_zipper_fill(_lib.ptr_table(devs), nf, xl, yl, sg, Nx, Ny, Nz, Hx, Hy, Hz, kstart, kcount, ft, None)
    else:
        rc = lib.tpg_fill_halo_regions(_lib.ptr_table(devs), nf, xl, yl, sg, Nx, Ny, Nz, Hx, Hy, Hz, 1, ft, None)
    torch.cuda.synchronize()
    if rc != 0:
        bad += 1; print("ERROR", t, rc, lib.tpg_last_error(), (Nx, Ny, Nz, Hx, Hy, Hz)); continue
    for d, h, (x, y, s) in zip(devs, hosts, specs):
        if zip_only: oracle.zipper_fill(h, x, y, s, (Nx, Ny, Nz), (Hx, Hy, Hz), kstart, kcount)
        else: oracle.fill_halo_regions(h, x, y, s, (Nx, Ny, Nz), (Hx, Hy, Hz))
        if not np.array_equal(d.cpu().numpy(), h):
            bad += 1; print("MISMATCH", t, (Nx, Ny, Nz, Hx, Hy, Hz), (x, y, s), dt.__name__, os.environ.get("TPG_FILL_FUSED"), os.environ["TPG_ZIPPER_VARIANT"]); break
    if t % 500 == 499: print(f"{t + 1} trials, {bad} bad", flush=True)
print("done:", trials, "trials,", bad, "bad")
sys.exit(1 if bad else 0)
