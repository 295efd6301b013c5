"""bench_aux.py -- the auxiliary measurements of `python bench.py` at N = 1.  They run AFTER the timed region and after every pass the
contract keys are read from (`value` is the same with --no-aux / --no-fill-step): the fold and the merged fill by cache state, the same-shape
copy ceiling, the 8- and 16-field batched fold, Float32, BASELINE config 2, the geometry utilities (`auxiliary`); the fills of one baroclinic
step of BASELINE config 5 at the default halo 4 (`fill_step_config5`) and at the reference's own model halo (5, 5, 5)
(`fill_step_halo5`, bench_halo5.py)."""
import ctypes as C
import statistics

from bench_common import HBM_PEAK_GBPS, NX, NY, NZ, H, SPECS, periodic_algorithmic_bytes, zipper_algorithmic_bytes
from bench_halo5 import fill_step_halo5                                # noqa: F401  (re-exported for bench.py)

AUX_PREROLL = 64


def auxiliary(torch, osg, _lib, lib, tlib, testlib, dev, fields, fptrs, xl, yl, sg, geom, p, out, out_ptrs, ws, hip_event, elapsed_ms):
    """Measurements of their own, N = 1 only, before the warm-up steps: the fold and the merged fill by cache state, the
    same-shape copy ceiling, Float32 fold / fill / build, BASELINE config 2, the geometry utilities."""
    n = len(SPECS)
    stream = _lib.current_stream_ptr(dev)
    ev = lambda: torch.cuda.Event(enable_timing=True)
    flush = torch.zeros(1 << 27, dtype=torch.float64, device=dev)          # 1 GiB: evicts L2 + Infinity Cache
    e0, e1 = hip_event(), hip_event()

    def fold(evs):
        _lib.check(lib.tpg_zipper_fill_timed(fptrs, n, xl, yl, sg, *geom, 1, NZ, _lib.TPG_F64, stream, evs[0], evs[1]))

    def merged(evs):
        _lib.check(lib.tpg_fill_halo_regions_timed(fptrs, n, xl, yl, sg, *geom, 1, _lib.TPG_F64, stream, evs[0], evs[1]))

    acc = {"cold_dirty": [], "cold_clean": [], "warm": [], "copy_cold_clean": [], "merged_cold_clean": [], "merged_cold_dirty": [], "merged_warm": []}
    for it in range(22):
        flush.add_(1.0)                                                     # predecessor leaves the caches full of dirty lines
        fold((e0, e1)); acc["cold_dirty"].append(elapsed_ms(e0, e1))
        flush.sum()                                                         # ... full of clean lines
        fold((e0, e1)); acc["cold_clean"].append(elapsed_ms(e0, e1))
        fold((e0, e1)); acc["warm"].append(elapsed_ms(e0, e1))              # back-to-back relaunch (Infinity-Cache resident)
        flush.sum()
        testlib.check(tlib.tpg_zipper_copy_probe(fptrs, n, yl, *geom, _lib.TPG_F64, stream, e0, e1))
        acc["copy_cold_clean"].append(elapsed_ms(e0, e1))
        flush.add_(1.0)
        merged((e0, e1)); acc["merged_cold_dirty"].append(elapsed_ms(e0, e1))
        flush.sum()
        merged((e0, e1)); acc["merged_cold_clean"].append(elapsed_ms(e0, e1))
        merged((e0, e1)); acc["merged_warm"].append(elapsed_ms(e0, e1))
    med = {k: statistics.median(v[2:]) for k, v in acc.items()}             # first 2 rounds dropped
    aux = {"zipper_cold_ms": med["cold_clean"], "zipper_cold_dirty_ms": med["cold_dirty"], "zipper_warm_ms": med["warm"],
           "zipper_copy_ceiling_ms": med["copy_cold_clean"],
           "fill_merged_cold_ms": med["merged_cold_clean"], "fill_merged_cold_dirty_ms": med["merged_cold_dirty"], "fill_merged_warm_ms": med["merged_warm"],
           "zipper_states_note": "kernel start/stop events, median of 20: after a 1 GiB read-only pass (cold), after a 1 GiB "
                                 "in-place write (cold_dirty), back-to-back relaunch (warm); zipper_* = the fold alone (k_zipper_cols), "
                                 "fill_merged_* = the whole fill (k_fill_merged); copy_ceiling = the fold's launch shape and bytes as a "
                                 "pure copy (tpg_zipper_copy_probe, test library), cold"}
    del flush
    for fid, f in enumerate(fields):                                        # the copy probe left unfolded halos behind
        testlib.check(tlib.tpg_fill_synthetic(f.data_ptr(), 0x5EED + fid, 12345.0, *geom, _lib.TPG_F64, None))

    # ---- Float32 (the reference tests FT in {Float32, Float64}, test/runtests.jl:10): fold, whole fill, build ----------------
    flush = torch.zeros(1 << 27, dtype=torch.float64, device=dev)
    f32 = [torch.empty((NZ + 2 * H, NY + 2 * H, NX + 2 * H), dtype=torch.float32, device=dev) for _ in SPECS]
    for fid, f in enumerate(f32):
        testlib.check(tlib.tpg_fill_synthetic(f.data_ptr(), 0xF32 + fid, 12345.0, *geom, _lib.TPG_F32, None))
    p32 = _lib.ptr_table(f32)
    a32 = {"fold": [], "fill": []}
    for it in range(12):
        flush.sum()
        _lib.check(lib.tpg_zipper_fill_timed(p32, n, xl, yl, sg, *geom, 1, NZ, _lib.TPG_F32, stream, e0, e1)); a32["fold"].append(elapsed_ms(e0, e1))
        flush.sum()
        _lib.check(lib.tpg_fill_halo_regions_timed(p32, n, xl, yl, sg, *geom, 1, _lib.TPG_F32, stream, e0, e1)); a32["fill"].append(elapsed_ms(e0, e1))
    del f32, flush
    zb32 = sum(zipper_algorithmic_bytes(NX, NZ, H, s=4).values())
    pb32 = periodic_algorithmic_bytes(NY, NZ, H, n, s=4)
    t_fold32, t_fill32 = statistics.median(a32["fold"][2:]), statistics.median(a32["fill"][2:])
    lib.tpg_event_destroy(e0); lib.tpg_event_destroy(e1)

    # ---- the fold (and the whole fill) of 8 and 16 fields in ONE launch: the caller's real regime (examples/bickley_jet.jl:44-55 fills
    # u, v, c, eta, U, V ... together; SURVEY.md 7 hard part 3 asks for a batched-fields figure beside cold / warm).  Same geometry as the
    # headline (3600 x 1800 x 75, halo 4), locations cycling c/u/v/zeta, cold, the kernel's own events, median of 10 after 2 dropped.
    # The 4-field `roofline_fold` stays the headline; this shows at which field count the fixed ramp + drain of a launch stops mattering.
    flush = torch.zeros(1 << 27, dtype=torch.float64, device=dev)
    e0, e1 = hip_event(), hip_event()
    nb = 16
    bf = [torch.empty((NZ + 2 * H, NY + 2 * H, NX + 2 * H), dtype=torch.float64, device=dev) for _ in range(nb)]
    for fid, f in enumerate(bf):
        testlib.check(tlib.tpg_fill_synthetic(f.data_ptr(), 0xBA7C + fid, 12345.0, *geom, _lib.TPG_F64, None))
    bspecs = [SPECS[i % len(SPECS)] for i in range(nb)]
    batched = []
    for nf in (8, 16):
        pt = _lib.ptr_table(bf[:nf])
        bx = (C.c_int8 * nf)(*[q[1] for q in bspecs[:nf]]); by = (C.c_int8 * nf)(*[q[2] for q in bspecs[:nf]]); bs = (C.c_int32 * nf)(*[q[3] for q in bspecs[:nf]])
        zb = sum(sum(zipper_algorithmic_bytes(NX, NZ, H, [q]).values()) for q in bspecs[:nf])
        pb = periodic_algorithmic_bytes(NY, NZ, H, nf)
        tf, tm = [], []
        for it in range(12):
            flush.sum()
            _lib.check(lib.tpg_zipper_fill_timed(pt, nf, bx, by, bs, *geom, 1, NZ, _lib.TPG_F64, stream, e0, e1)); tf.append(elapsed_ms(e0, e1))
            flush.sum()
            _lib.check(lib.tpg_fill_halo_regions_timed(pt, nf, bx, by, bs, *geom, 1, _lib.TPG_F64, stream, e0, e1)); tm.append(elapsed_ms(e0, e1))
        t_f, t_m = statistics.median(tf[2:]), statistics.median(tm[2:])
        batched.append({"fields": nf, "launch_ms": t_f, "algorithmic_bytes_per_launch": zb, "achieved": zb / (t_f * 1e-3) / 1e9,
                        "frac": zb / (t_f * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                        "merged_fill_launch_ms": t_m, "merged_fill_algorithmic_bytes": zb + pb,
                        "merged_fill_frac": (zb + pb) / (t_m * 1e-3) / 1e9 / HBM_PEAK_GBPS})
    aux["roofline_fold_batched"] = batched
    aux["roofline_fold_batched_note"] = ("k_zipper_cols<double,2,4> / k_fill_merged<double,2,4> over 8 and 16 fields of the headline geometry in ONE launch "
                                         "(locations cycling c/u/v/zeta), cold (after a 1 GiB read-only pass), kernel start/stop events, median of 10; unit GB/s "
                                         "against the 8000 GB/s peak")
    del bf, flush
    lib.tpg_event_destroy(e0); lib.tpg_event_destroy(e1)

    def builds():
        """the two build-only measurements: Float32 at 1/10 degree, Float64 at 1/4 degree (BASELINE config 2)"""
        pf = _lib.TpgParams(NX, NY, NZ, H, H, H, -80.0, 55.0, 70.0, osg.R_Earth, _lib.TPG_F32, 1, NY, 0)
        outf = [torch.empty((NY + 2 * H, NX + 2 * H), dtype=torch.float32, device=dev) for _ in _lib.ARRAY_NAMES]
        ptrf = _lib.ptr_table(outf)
        for _ in range(AUX_PREROLL):                                        # the same declared pre-roll as before the warm-up steps (Float64 builds)
            _lib.check(lib.tpg_build_grid(C.byref(p), out_ptrs, ws.data_ptr(), ws.numel(), stream))
        for _ in range(3):
            _lib.check(lib.tpg_build_grid(C.byref(pf), ptrf, ws.data_ptr(), ws.numel(), stream))
        b0, b1 = ev(), ev()
        b0.record()
        for _ in range(20):
            _lib.check(lib.tpg_build_grid(C.byref(pf), ptrf, ws.data_ptr(), ws.numel(), stream))
        b1.record(); torch.cuda.synchronize()
        usf = b0.elapsed_time(b1) / 20 * 1e3
        del outf
        aux["float32"] = {
            "fold_ms": t_fold32, "fold_algorithmic_bytes": zb32, "fold_frac_of_hbm_peak": zb32 / (t_fold32 * 1e-3) / 1e9 / HBM_PEAK_GBPS,
            "fold_kernel": "k_zipper_cols<float,4,4>, 4 fields x 75 levels, cold, kernel events, median of 10",
            "fill_ms": t_fill32, "fill_algorithmic_bytes": zb32 + pb32, "fill_frac_of_hbm_peak": (zb32 + pb32) / (t_fill32 * 1e-3) / 1e9 / HBM_PEAK_GBPS,
            "fill_kernel": "k_fill_merged<float,4,4>, same fields, cold, kernel events, median of 10",
            "build_us": usf, "build_cells_per_s": NX * NY / (usf * 1e-6), "build_store_GBps": 80.0 * (NX + 2 * H) * (NY + 2 * H) / (usf * 1e-6) / 1e9,
            "build_note": "3600x1800 Float32 grid: the Float64 pipeline on Float32-rounded lambda tables, rounded once at the store (SURVEY A-1); 20 builds "
                          f"back to back after a pre-roll of {AUX_PREROLL} Float64 builds + 3 untimed Float32 ones (sustained clocks, like the timed steps)",
            "preroll_builds": AUX_PREROLL}
        # BASELINE config 2: the 1/4 degree (1440 x 720) Float64 metric precompute alone, 20 back-to-back builds
        p2 = _lib.TpgParams(1440, 720, 1, H, H, H, -80.0, 55.0, 70.0, osg.R_Earth, _lib.TPG_F64, 1, 720, 0)
        out2 = [torch.empty((720 + 2 * H, 1440 + 2 * H), dtype=torch.float64, device=dev) for _ in _lib.ARRAY_NAMES]
        ptr2 = _lib.ptr_table(out2)
        ws2 = torch.empty(int(lib.tpg_build_grid_workspace_bytes(C.byref(p2))), dtype=torch.uint8, device=dev)
        for _ in range(3):
            _lib.check(lib.tpg_build_grid(C.byref(p2), ptr2, ws2.data_ptr(), ws2.numel(), stream))
        batches = []                                                        # 3 batches of 20 back-to-back builds; the median batch is the figure
        for _ in range(3):                                                  # (the FIRST use of freshly allocated arrays in the first GPU process of a
            b0, b1 = ev(), ev()                                             # fresh box has shown a one-off ~40 ms stall inside a batch)
            b0.record()
            for _ in range(20):
                _lib.check(lib.tpg_build_grid(C.byref(p2), ptr2, ws2.data_ptr(), ws2.numel(), stream))
            b1.record(); torch.cuda.synchronize()
            batches.append(b0.elapsed_time(b1) / 20 * 1e3)
        us2 = statistics.median(batches)
        aux["config2_quarter_degree_build"] = {"size": [1440, 720, 1], "us_per_build": us2, "cells_per_s": 1440 * 720 / (us2 * 1e-6),
                                               "store_GBps": 160.0 * 1448 * 728 / (us2 * 1e-6) / 1e9, "us_per_build_batches_of_20": batches}
        del out2, ws2

    def geometry():
        # SURVEY 8(f-4) geometry utilities at the bench's own size, on the grid arrays the warm-up build just has to produce
        _lib.check(lib.tpg_build_grid(C.byref(p), out_ptrs, ws.data_ptr(), ws.numel(), stream))
        arr = dict(zip(_lib.ARRAY_NAMES, out))
        angle = torch.empty((NY, NX), dtype=torch.float64, device=dev)
        uo, vo = torch.zeros_like(fields[0]), torch.zeros_like(fields[0])

        def timed_us(fn, reps):
            fn(); torch.cuda.synchronize()
            t0_, t1_ = ev(), ev()
            t0_.record()
            for _ in range(reps):
                fn()
            t1_.record(); torch.cuda.synchronize()
            return t0_.elapsed_time(t1_) / reps * 1e3

        t_ang = timed_us(lambda: _lib.check(lib.tpg_nonorthogonality_angle(arr["lambda_ff"].data_ptr(), arr["phi_ff"].data_ptr(), None,
                                                                           angle.data_ptr(), NX, NY, H, H, _lib.TPG_F64, stream)), 20)
        t_rot = timed_us(lambda: _lib.check(lib.tpg_convert_frame(arr["phi_cf"].data_ptr(), arr["phi_fc"].data_ptr(), arr["dy_cc"].data_ptr(),
                                                                  arr["dx_cc"].data_ptr(), fields[0].data_ptr(), fields[1].data_ptr(),
                                                                  uo.data_ptr(), vo.data_ptr(), 0, *geom, _lib.TPG_F64, stream)), 5)
        rot_bytes = 4 * NX * NY * NZ * 8                                        # 2 fields read + 2 written, interior cells
        aux["geometry_utilities"] = {
            "nonorthogonality_angle_us": t_ang, "nonorthogonality_max_abs_deg_unmasked": float(angle.abs().max()),
            "convert_frame_us": t_rot, "convert_frame_algorithmic_bytes": rot_bytes,
            "convert_frame_frac_of_hbm_peak": rot_bytes / (t_rot * 1e-6) / 1e9 / HBM_PEAK_GBPS}
        del angle, uo, vo

    geometry(); builds()
    return aux


def fill_step_config5(torch, osg, _lib, tlib, dev):
    """BASELINE config 5: the halo fills of ONE baroclinic step of a hydrostatic model with a split-explicit
    free surface on the 1/24 degree x 100 level tripolar grid (test/runtests.jl:46-77, examples/bickley_jet.jl:44-55):
      * one tupled fill of the 3-D prognostic fields (u, v, T, S, c): 5 x 32.3 GB of Float64 resident on one MI355X;
      * 30 sub-step fills of the 2-D fields (eta, U, V) with the extended north halo (Hy = 31), replayed from one HIP graph.
    Separate from the timed steps; parity of exactly these fills is tests/test_gpu_config5.py."""
    size, halo, substeps = (8640, 4320, 100), (4, 4, 4), 30
    Nx, Ny, Nz = size
    free, _ = torch.cuda.mem_get_info(dev)
    need = 5 * (Nx + 8) * (Ny + 8) * (Nz + 8) * 8 + 16e9
    if free < need:
        return {"skipped": f"needs {need / 1e9:.0f} GB of free HBM, {free / 1e9:.0f} GB available"}
    grid = osg.TripolarGrid(osg.GPU(dev.index), torch.float64, size=size, halo=halo)
    ext = osg.TripolarGrid(osg.GPU(dev.index), torch.float64, size=(Nx, Ny, 1), halo=(halo[0], substeps + 1, halo[2]))
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
    t3 = timed(plan3, 10)                                         # back to back: each launch runs behind its predecessor's dirty lines, as after a model's tendency kernels
    flush = torch.zeros(1 << 27, dtype=torch.float64, device=dev)
    cold = []
    for _ in range(7):                                            # the same launch after a 1 GiB read-only pass (clean caches)
        flush.sum()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); plan3(); e1.record(); torch.cuda.synchronize()
        cold.append(e0.elapsed_time(e1) * 1e3)
    t3_cold = statistics.median(cold[2:])
    del flush
    graph = osg.halo_fill_plan(f2).graph(repeat=substeps)
    t2 = timed(graph.replay, 20)
    specs3 = [("u", 1, 0, -1), ("v", 0, 1, -1), ("T", 0, 0, 1), ("S", 0, 0, 1), ("c", 0, 0, 1)]
    zb = sum(zipper_algorithmic_bytes(Nx, Nz, halo[1], specs3).values())
    pb = periodic_algorithmic_bytes(Ny, Nz, halo[0], 5)
    rows = 5 * (Ny + 2 * halo[1]) * (Nz + 2 * halo[2])
    out = {"workload": "1/24deg (8640x4320x100, halo 4, Float64): tupled fill_halo_regions!((u,v,T,S,c)) [one merged launch] + "
                       f"{substeps} sub-step fills of (eta,U,V) with north halo {substeps + 1} [one fused launch each, one HIP graph]",
           "fields_GB": sum(f.data.numel() for f in f3) * 8 / 1e9,
           "fill3d_us": t3, "fill3d_cold_us": t3_cold,
           "fill3d_states_note": "fill3d_us: 10 fills back to back (dirty predecessor lines in the caches: the state a model step leaves); fill3d_cold_us: "
                                 "the same call after a 1 GiB read-only pass, event bracket, median of 5",
           "substep_fills_us": t2, "substeps": substeps, "total_us": t3 + t2,
           "fill3d_algorithmic_bytes": zb + pb,
           "fill3d_algorithmic_frac_of_hbm_peak": (zb + pb) / (t3 * 1e-6) / 1e9 / HBM_PEAK_GBPS,
           # modelled, not counted: the periodic part touches 3 whole 128-B lines per row pair twice (fetch + write-back), the fold whole lines
           "fill3d_modelled_line_ops": (zb // 128) + rows * 3, "fill3d_modelled_lines_per_ns": ((zb // 128) + rows * 3) / (t3 * 1e3),
           "fill3d_cold_modelled_lines_per_ns": ((zb // 128) + rows * 3) / (t3_cold * 1e3),
           "substep_fill_us_each": t2 / substeps}
    del plan3, f3, f2, grid, ext, graph
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    return out
