"""bench_halo5.py -- halo fills at the reference's own MODEL geometry, halo = (5, 5, 5).

Both model examples of the reference build their grids with `halo = (5, 5, 5)` (examples/bickley_jet.jl:21,
examples/distributed_bickley_jet.jl:23): an odd Hx, for which a Float64 row no longer splits into 16-B chunks at the interior /
x-halo boundary.  This block times exactly those fills, with halo 4 measured by the SAME method beside them so that the two can
be compared per algorithmic byte:

  * headline geometry 3600 x 1800 x 75, fields c/u/v/zeta: the whole fill (tpg_fill_halo_regions) and the fold alone
    (tpg_zipper_fill), cold (after a 1 GiB read-only pass), median of 10 after 2 dropped;
  * BASELINE config 5's caller, 8640 x 4320 x 100: the tupled fill of (u, v, T, S, c) through a HaloFillPlan and the 30 sub-step
    fills of (eta, U, V) with the extended north halo replayed from one HIP graph -- as `fill_step`, at halo 5.

`fill_ms` is a stream-event bracket around the one C call (it covers every launch the call makes, one or two);
`fill_first_kernel_ms` is the first kernel's own start/stop events -- equal to the launch duration when the fill is ONE launch.
Imported by bench.py's auxiliary section (`fill_step_halo5`) and runnable alone:  python bench_halo5.py [--no-config5]
"""
import ctypes as C
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0
SPECS = [("c", 0, 0, 1), ("u", 1, 0, -1), ("v", 0, 1, -1), ("zeta", 1, 1, 1)]


def fold_bytes(nx, nz, hy, specs, s=8):
    """CF/FF Nx*Nz*Hy*2*s; CC/FC add the row-Ny substitution (Nx/2)*Nz*2*s"""
    return sum(nx * nz * hy * 2 * s + ((nx // 2) * nz * 2 * s if yl == 0 else 0) for _, _, yl, _ in specs)


def periodic_bytes(ny, nz, halo, nfields, s=8):
    """2 Hx elements read + 2 Hx written per row, every row and level of the parent"""
    hx, hy, hz = halo
    return nfields * (ny + 2 * hy) * (nz + 2 * hz) * 2 * hx * 2 * s


def headline_fill(torch, _lib, lib, tlib, dev, h, size=(3600, 1800, 75), reps=12, f32=False):
    """4-field fill and fold at `size`, halo (h, h, h), Float64 (or Float32), cold"""
    from tools import testlib
    nx, ny, nz = size
    geom = (nx, ny, nz, h, h, h)
    n = len(SPECS)
    tdt, ft, esz = (torch.float32, _lib.TPG_F32, 4) if f32 else (torch.float64, _lib.TPG_F64, 8)
    fields = [torch.empty((nz + 2 * h, ny + 2 * h, nx + 2 * h), dtype=tdt, device=dev) for _ in SPECS]
    for fid, f in enumerate(fields):
        testlib.check(tlib.tpg_fill_synthetic(f.data_ptr(), 0x5A10 + fid, 12345.0, *geom, ft, None))
    pt = _lib.ptr_table(fields)
    xl = (C.c_int8 * n)(*[s[1] for s in SPECS]); yl = (C.c_int8 * n)(*[s[2] for s in SPECS]); sg = (C.c_int32 * n)(*[s[3] for s in SPECS])
    stream = _lib.current_stream_ptr(dev)
    flush = torch.zeros(1 << 27, dtype=torch.float64, device=dev)          # 1 GiB: evicts L2 + Infinity Cache
    k0, k1 = C.c_void_p(), C.c_void_p()
    _lib.check(lib.tpg_event_create(C.byref(k0))); _lib.check(lib.tpg_event_create(C.byref(k1)))

    def kernel_ms():
        ms = C.c_float()
        _lib.check(lib.tpg_event_elapsed_ms(k0, k1, C.byref(ms)))
        return ms.value

    t_fill, t_first, t_fold = [], [], []
    for _ in range(reps):
        flush.sum()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _lib.check(lib.tpg_fill_halo_regions_timed(pt, n, xl, yl, sg, *geom, 1, ft, stream, k0, k1))
        e1.record()
        torch.cuda.synchronize()
        t_fill.append(e0.elapsed_time(e1)); t_first.append(kernel_ms())
        flush.sum()
        _lib.check(lib.tpg_zipper_fill_timed(pt, n, xl, yl, sg, *geom, 1, nz, ft, stream, k0, k1))
        t_fold.append(kernel_ms())
    lib.tpg_event_destroy(k0); lib.tpg_event_destroy(k1)
    del fields, flush
    torch.cuda.empty_cache()
    zb, pb = fold_bytes(nx, nz, h, SPECS, esz), periodic_bytes(ny, nz, (h, h, h), n, esz)
    med = lambda v: statistics.median(v[2:])
    fill, first, fold = med(t_fill), med(t_first), med(t_fold)
    return {"size": list(size), "halo": [h, h, h], "fields": [s[0] for s in SPECS], "eltype": "Float32" if f32 else "Float64",
            "fill_ms": fill, "fill_first_kernel_ms": first, "fill_algorithmic_bytes": zb + pb,
            "fill_ns_per_algorithmic_KB": fill * 1e6 / ((zb + pb) / 1e3),
            "fill_frac_of_hbm_peak": (zb + pb) / (fill * 1e-3) / 1e9 / HBM_PEAK_GBPS,
            "fill_first_kernel_frac_of_hbm_peak": (zb + pb) / (first * 1e-3) / 1e9 / HBM_PEAK_GBPS,
            "fold_ms": fold, "fold_algorithmic_bytes": zb, "fold_frac_of_hbm_peak": zb / (fold * 1e-3) / 1e9 / HBM_PEAK_GBPS}


def config5_fills(torch, osg, _lib, tlib, dev, h, substeps=30):
    """the fills of one baroclinic step at 8640 x 4320 x 100 (bench.py's `fill_step`) at halo (h, h, h)"""
    size = (8640, 4320, 100)
    nx, ny, nz = size
    free, _ = torch.cuda.mem_get_info(dev)
    need = 5 * (nx + 2 * h) * (ny + 2 * h) * (nz + 2 * h) * 8 + 16e9
    if free < need:
        return {"skipped": f"needs {need / 1e9:.0f} GB of free HBM, {free / 1e9:.0f} GB available"}
    grid = osg.TripolarGrid(osg.GPU(dev.index), torch.float64, size=size, halo=(h, h, h))
    ext = osg.TripolarGrid(osg.GPU(dev.index), torch.float64, size=(nx, ny, 1), halo=(h, substeps + 1, h))
    f3 = (osg.XFaceField(grid), osg.YFaceField(grid), osg.CenterField(grid), osg.CenterField(grid), osg.CenterField(grid))
    f2 = (osg.Field((osg.Center, osg.Center, None), ext), osg.Field((osg.Face, osg.Center, None), ext), osg.Field((osg.Center, osg.Face, None), ext))
    for k, f in enumerate(f3 + f2):
        assert tlib.tpg_fill_synthetic(f.data.data_ptr(), 0xF5 + k, 12345.0, f.Nx, f.Ny, f.Nz, f.Hx, f.Hy, f.Hz, _lib.TPG_F64, None) == 0

    def timed(fn, reps):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3                     # us

    plan3 = osg.halo_fill_plan(f3)
    batches = [timed(plan3, 10) for _ in range(3)]                # back to back (dirty predecessor lines): median of 3 batches of 10
    t3 = statistics.median(batches)
    flush = torch.zeros(1 << 27, dtype=torch.float64, device=dev)
    cold = []
    for _ in range(7):                                            # the same call after a 1 GiB read-only pass
        flush.sum()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); plan3(); e1.record(); torch.cuda.synchronize()
        cold.append(e0.elapsed_time(e1) * 1e3)
    t3_cold = statistics.median(cold[2:])
    del flush
    graph = osg.halo_fill_plan(f2).graph(repeat=substeps)
    t2 = timed(graph.replay, 20)
    specs3 = [("u", 1, 0, -1), ("v", 0, 1, -1), ("T", 0, 0, 1), ("S", 0, 0, 1), ("c", 0, 0, 1)]
    zb, pb = fold_bytes(nx, nz, h, specs3), periodic_bytes(ny, nz, (h, h, h), 5)
    out = {"size": list(size), "halo": [h, h, h], "fields_GB": sum(f.data.numel() for f in f3) * 8 / 1e9,
           "fill3d_us": t3, "fill3d_us_batches_of_10": batches, "fill3d_cold_us": t3_cold, "fill3d_algorithmic_bytes": zb + pb, "fill3d_ns_per_algorithmic_KB": t3 * 1e3 / ((zb + pb) / 1e3),
           "fill3d_algorithmic_frac_of_hbm_peak": (zb + pb) / (t3 * 1e-6) / 1e9 / HBM_PEAK_GBPS,
           "substep_fills_us": t2, "substeps": substeps, "substep_fill_us_each": t2 / substeps, "total_us": t3 + t2}
    del plan3, f3, f2, grid, ext, graph
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    return out


def fill_step_halo5(torch, osg, _lib, lib, tlib, dev, config5=True):
    """bench.py's `fill_step_halo5` object: halo 5 beside halo 4 (same method), headline size and config 5"""
    out = {"why": "examples/bickley_jet.jl:21 and examples/distributed_bickley_jet.jl:23 build their grids with halo = (5, 5, 5)",
           "headline_halo5": headline_fill(torch, _lib, lib, tlib, dev, 5),
           "headline_halo4_same_method": headline_fill(torch, _lib, lib, tlib, dev, 4)}
    a, b = out["headline_halo5"], out["headline_halo4_same_method"]
    out["headline_time_per_byte_halo5_over_halo4"] = a["fill_ns_per_algorithmic_KB"] / b["fill_ns_per_algorithmic_KB"]
    out["headline_fold_time_per_byte_halo5_over_halo4"] = (a["fold_ms"] / a["fold_algorithmic_bytes"]) / (b["fold_ms"] / b["fold_algorithmic_bytes"])
    # Float32: rows of 3610 elements are no 16-B rows (the GEN form with r = 1 outermost column per side)
    a, b = headline_fill(torch, _lib, lib, tlib, dev, 5, f32=True), headline_fill(torch, _lib, lib, tlib, dev, 4, f32=True)
    out["headline_float32"] = {"halo5": a, "halo4_same_method": b,
                               "time_per_byte_halo5_over_halo4": a["fill_ns_per_algorithmic_KB"] / b["fill_ns_per_algorithmic_KB"],
                               "fold_time_per_byte_halo5_over_halo4": (a["fold_ms"] / a["fold_algorithmic_bytes"]) / (b["fold_ms"] / b["fold_algorithmic_bytes"])}
    if config5:
        out["config5_halo5"] = config5_fills(torch, osg, _lib, tlib, dev, 5)
        out["config5_halo4_same_method"] = config5_fills(torch, osg, _lib, tlib, dev, 4)
        a, b = out["config5_halo5"], out["config5_halo4_same_method"]
        if "skipped" not in a and "skipped" not in b:
            out["config5_time_per_byte_halo5_over_halo4"] = a["fill3d_ns_per_algorithmic_KB"] / b["fill3d_ns_per_algorithmic_KB"]
            out["config5_cold_time_per_byte_halo5_over_halo4"] = (a["fill3d_cold_us"] / a["fill3d_algorithmic_bytes"]) / (b["fill3d_cold_us"] / b["fill3d_algorithmic_bytes"])
            out["config5_substep_fill_halo5_over_halo4"] = a["substep_fill_us_each"] / b["substep_fill_us_each"]
    out["method"] = ("Float64; headline: cold (after a 1 GiB read-only pass), median of 10; fill_ms = stream-event bracket around the one "
                     "tpg_fill_halo_regions call (all its launches), fill_first_kernel_ms / fold_ms = the kernel's own start/stop events; "
                     "config 5: as `fill_step` (plan call x 10, graph replay x 20, back to back)")
    return out


def main():
    import torch
    import orthogonalsphericalshellgrids.jl_amd as osg
    from orthogonalsphericalshellgrids.jl_amd import _lib
    from tools import testlib
    assert torch.cuda.is_available(), "needs a HIP device"
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    print(json.dumps(fill_step_halo5(torch, osg, _lib, _lib.lib(), testlib.lib(), dev, config5="--no-config5" not in sys.argv)))


if __name__ == "__main__":
    main()
