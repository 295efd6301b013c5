"""Developer diagnostic (GPU box): parity of build_grid / zipper against the oracle + rough timings."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import orthogonalsphericalshellgrids.jl_amd as osg
from oracle import oracle

def cmp_grid(size, halo=(4,4,4), dtype=torch.float64, **kw):
    npd = np.float64 if dtype == torch.float64 else np.float32
    t = time.time(); ref = oracle.build_grid(size, halo=halo, dtype=npd, **kw); tc = time.time() - t
    torch.cuda.synchronize(); t = time.time()
    g = osg.TripolarGrid(osg.GPU(0), dtype, size=size, halo=halo, **kw); torch.cuda.synchronize(); tg = time.time() - t
    worst, nbad = 0.0, 0
    for name, r in ref.items():
        got = getattr(g, name).cpu().numpy()
        same = (got == r) | (np.isnan(got) & np.isnan(r))
        nb = int((~same).sum())
        if nb:
            with np.errstate(divide="ignore", invalid="ignore"):
                rel = np.abs(got - r) / np.abs(r)
            rel[same] = 0
            w = float(np.nanmax(rel)); worst = max(worst, w); nbad += nb
            idx = np.unravel_index(np.nanargmax(rel), rel.shape)
            print(f"   {name}: {nb} differing, max rel {w:.3e} at j={idx[0]-halo[1]+1} i={idx[1]-halo[0]+1} got={got[idx]!r} ref={r[idx]!r}")
    print(f"grid {size} halo {halo} {dtype} {kw}: differing={nbad} max_rel={worst:.3e} cpu={tc:.3f}s gpu_wall={tg*1e3:.2f}ms")

def cmp_zip(size, halo, dtype=torch.float64):
    grid = osg.TripolarGrid(osg.GPU(0), dtype, size=size, halo=halo)
    rng = np.random.default_rng(1)
    ok = True
    fs, hs, meta = [], [], []
    for loc, (xl, yl) in (((osg.Center,osg.Center,osg.Center),(0,0)), ((osg.Face,osg.Center,osg.Center),(1,0)),
                          ((osg.Center,osg.Face,osg.Center),(0,1)), ((osg.Face,osg.Face,osg.Center),(1,1))):
        for sgn in (1, -1):
            f = osg.Field(loc, grid, boundary_conditions=osg.FieldBoundaryConditions(north=osg.ZipperBoundaryCondition(sgn)))
            h = rng.uniform(-1, 1, tuple(f.data.shape)).astype(np.float64 if dtype == torch.float64 else np.float32)
            f.data.copy_(torch.from_numpy(h)); fs.append(f); hs.append(h); meta.append((xl, yl, sgn))
    osg.fill_halo_regions(fs)
    for f, h, (xl, yl, sgn) in zip(fs, hs, meta):
        oracle.fill_halo_regions(h, xl, yl, sgn, size, halo)
        same = np.array_equal(f.data.cpu().numpy(), h)
        ok &= same
        if not same:
            d = np.argwhere(f.data.cpu().numpy() != h)
            print(f"   MISMATCH loc=({xl},{yl}) sign={sgn}: {len(d)} cells, first {d[:4].tolist()}")
    print(f"zipper {size} halo {halo} {dtype}: {'bit-exact' if ok else 'FAIL'}")

if __name__ == "__main__":
    print(torch.cuda.get_device_name(0))
    cmp_grid((60, 30, 1)); cmp_grid((4, 5, 1), north_poles_latitude=35, first_pole_longitude=75)
    cmp_grid((4, 5, 1), dtype=torch.float32, north_poles_latitude=35, first_pole_longitude=75)
    cmp_grid((10, 10, 1)); cmp_grid((360, 180, 1), north_poles_latitude=35, first_pole_longitude=75)
    cmp_grid((62, 31, 1), halo=(3, 2, 1)); cmp_grid((1440, 720, 1)); cmp_grid((60, 30, 1), dtype=torch.float32)
    cmp_zip((10, 10, 1), (4, 4, 4)); cmp_zip((60, 30, 3), (4, 4, 4)); cmp_zip((62, 31, 2), (3, 2, 1))
    cmp_zip((64, 30, 3), (4, 4, 4), torch.float32); cmp_zip((62, 30, 3), (4, 3, 2), torch.float32)
    cmp_zip((1440, 720, 5), (4, 4, 4))
    # timings
    for size in ((1440, 720, 1), (3600, 1800, 1)):
        g = osg.TripolarGrid(osg.GPU(0), size=size); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): g = osg.TripolarGrid(osg.GPU(0), size=size)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        print(f"build {size}: {ms:.3f} ms -> {size[0]*size[1]/ms*1e3:.3e} cells/s")
