#!/usr/bin/env python3
"""bench.py -- TripolarGrid metric precompute + zipper halo fill at 1/10 deg x 75 levels on MI355X.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--scaling strong|weak] [--exchange auto|monolithic|pipelined_1|pipelined_2]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...     (same thing, launcher supplied)
  python bench.py --loopback [--loopback-bands R] [--loopback-band r]                          (1 GPU: the RCCL branch on one rank)

This file is the N = 1 step, the contract line and the dispatch.  N > 1 (BASELINE config 4: the same globe as N latitude bands, one
process per GPU, RCCL seam exchange) and its one-GPU rehearsals live in bench_chain.py; the auxiliary measurements that follow the
timed region in bench_aux.py / bench_halo5.py; shared constants and byte counts in bench_common.py.

One "step" = one pass of the hot path over one batch of synthetic input, resident in HBM:
  (1) tpg_build_grid : coordinates + 12 staggered metrics of the whole 3600 x 1800 globe, Float64, halo 4 -> 20 padded arrays;
  (2) fill_halo_regions! of the 4 synthetic Float64 fields c(CC,+1) u(FC,-1) v(CF,-1) zeta(FF,+1), 75 levels, through the
      product's own entry point tpg_fill_halo_regions (ONE merged launch: zipper fold + periodic x).
`python bench.py --gpus N` with N > 1 and no launcher in the environment starts its own N workers (bench_chain.launch_workers: the
parent never imports torch and never touches a GPU); `--loopback` runs that whole branch on ONE GPU (a communicator of one rank whose
peers are the rank itself: a rehearsal of the code path, not a scaling measurement).

Order of a run (N = 1): set-up -> the DECLARED clock pre-roll (`clock_preroll`: P plain tpg_build_grid calls, ~35 ms, so that a short
run starts from the sustained FP64 clock state; `--preroll 0` = none) -> exactly W untimed warm-up steps -> exactly K timed steps
(`value`, `ms_per_step`) -> untimed instrumented passes (per-phase breakdown, the fold-only pass behind `roofline_fold`) -> K steps
started right after >= 50 ms of HBM-bound work, no pre-roll, no warm-up (`ms_per_step_cold_onset`: what a caller's FIRST builds cost)
-> the auxiliary measurements (fold / fill by cache state, copy ceiling, the 8- and 16-field batched fold, Float32, config 2, geometry,
config 5 at halo 4 = `fill_step`, and the halo fills at the reference's own model halo (5, 5, 5) = `fill_step_halo5`) -> cpu_baseline.
Nothing optional precedes the timed region: `value` is the same with --no-aux / --no-fill-step.

value = horizontal grid cells / step time.  `roofline` is the halo-fill kernel the step launches (k_fill_merged: the zipper halo fill as
the product issues it, HBM-bound); `roofline_fold` is the fold alone (k_zipper_cols, the kernel BASELINE.json's north_star puts the
70 % target on); the precompute kernel, which dominates the step time but is FP64-issue bound, is `roofline_precompute`.
"""
import argparse
import ctypes as C
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from bench_common import (HBM_PEAK_GBPS, METRIC, NX, NY, NZ, H, SPECS, libraries_built, load_traffic,     # noqa: E402,F401
                          periodic_algorithmic_bytes, precompute_roofline, sources_sha16, zipper_algorithmic_bytes)
from bench_chain import Watchdog, chain_layout, launch_workers                                              # noqa: E402,F401  (no torch at import)


def cpu_share():
    """CPUs this process may actually use: the affinity mask capped by the cgroup CPU quota (a GPU box shows all host
    cores in the mask but grants a share of them)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]                  # cgroup v2
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())                      # cgroup v1
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except (OSError, ValueError):
            pass
    return n


def cpu_baseline():
    """The oracle (CPU restatement, kind "port") timed on this host on a bounded sample of the same workload, single
    thread and all cores of this process's CPU share: median of 10 full 3600 x 1800 Float64 grid builds,
    and of 10 four-field zipper fills on a (3600, 64, 75) stand-in (the fold touches only the top Hy+1 rows of each level,
    so bytes per level are identical)."""
    import numpy as np
    from oracle import oracle
    ncpu = cpu_share()
    ny_s = 64
    size, halo = (NX, ny_s, NZ), (H, H, H)
    fields = [np.random.default_rng(i).uniform(-1, 1, (NZ + 2 * H, ny_s + 2 * H, NX + 2 * H)) for i in range(4)]
    zbytes = sum(zipper_algorithmic_bytes(NX, NZ, H).values())

    def measure(threads, reps):
        oracle.set_threads(threads)
        oracle.build_grid((360, 180, 1))                              # warm the library / thread pool
        tb = []
        for _ in range(reps):
            t0 = time.perf_counter()
            oracle.build_grid((NX, NY, 1))
            tb.append(time.perf_counter() - t0)
        for f, (_, xl, yl, sg) in zip(fields, SPECS):
            oracle.zipper_fill(f, xl, yl, sg, size, halo)             # warm-up
        tz = []
        for _ in range(reps):
            t0 = time.perf_counter()
            for f, (_, xl, yl, sg) in zip(fields, SPECS):
                oracle.zipper_fill(f, xl, yl, sg, size, halo)
            tz.append(time.perf_counter() - t0)
        return statistics.median(tb), statistics.median(tz)

    b1, z1 = measure(1, 10)
    bn, zn = measure(ncpu, 10)
    oracle.set_threads(1)
    return {
        "value": NX * NY / (b1 + z1), "unit": "cells/s", "cores": 1, "kind": "port",
        "sample": f"oracle/tpg_oracle.c, 1 thread: median of 10 full 3600x1800 Float64 grid builds ({b1:.3f} s) + median of 10 "
                  f"4-field zipper fills on a 3600x64x75 stand-in ({z1 * 1e3:.2f} ms, same bytes per level)",
        "precompute_cells_per_s": NX * NY / b1, "zipper_GBps": zbytes / z1 / 1e9,
        "all_cores": {"value": NX * NY / (bn + zn), "unit": "cells/s", "cores": ncpu, "nproc": os.cpu_count(),
                      "precompute_cells_per_s": NX * NY / bn, "zipper_GBps": zbytes / zn / 1e9,
                      "sample": f"same sample, {ncpu} OpenMP threads (the process's CPU share; the reference's ~40 serial full-array "
                                f"passes stay serial): build {bn:.3f} s, zipper {zn * 1e3:.2f} ms"},
    }


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--scaling", choices=("strong", "weak"), default="strong",
                    help="N > 1: strong = BASELINE config 4 (the 3600x1800x75 globe in N bands of 1800/N rows; default); weak = 1800 rows per rank")
    ap.add_argument("--exchange", choices=("auto", "monolithic", "pipelined_1", "pipelined_2"), default="auto",
                    help="N > 1: form of the RCCL seam exchange in the timed steps; auto = monolithic first, the pipelined forms probed afterwards")
    ap.add_argument("--loopback", action="store_true",
                    help="1 GPU: run the N > 1 (RCCL) branch on a one-rank communicator whose peers are the rank itself")
    ap.add_argument("--loopback-bands", type=int, default=8, help="--loopback: length R of the emulated latitude-band chain")
    ap.add_argument("--loopback-band", type=int, default=3, help="--loopback: which band of the chain this GPU plays (R-1 = the zipper band)")
    ap.add_argument("--deadline", type=float, default=float(os.environ.get("TPG_BENCH_DEADLINE_S", "120")),
                    help="seconds allowed for communicator bring-up and for the first seam exchange (N > 1)")
    ap.add_argument("--rendezvous-deadline", type=float, default=float(os.environ.get("TPG_BENCH_RENDEZVOUS_DEADLINE_S", "900")),
                    help="seconds allowed for the process-group rendezvous + RCCL communicator creation (covers a cold `import torch` on every rank)")
    ap.add_argument("--preroll", type=int, default=-1,
                    help="plain tpg_build_grid calls before the warm-up steps (the declared clock pre-roll); default 64 per 1800 rows of band, 0 = none")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fill-step", action="store_true", help="skip the config-5 (1/24 deg x 100 levels) measurements fill_step and fill_step_halo5")
    ap.add_argument("--no-cold-onset", action="store_true", help="skip the K cold-onset steps (profiling passes: keeps the 100 streaming launches out of the trace)")
    ap.add_argument("--no-aux", action="store_true", help="skip the auxiliary measurements (cache states, copy ceiling, Float32, config 2, geometry)")
    return ap.parse_args()


def main():
    args = parse_args()
    if args.loopback and args.gpus != 1:
        raise SystemExit("--loopback is a one-GPU mode (--gpus 1)")
    if args.loopback and not (0 <= args.loopback_band < args.loopback_bands and args.loopback_bands >= 2):
        raise SystemExit("--loopback-band must lie in 0 .. --loopback-bands - 1 (bands >= 2)")
    if args.gpus > 1 and int(os.environ.get("WORLD_SIZE", "1") or "1") == 1:
        # no launcher (WORLD_SIZE unset, or a single-process environment that exports WORLD_SIZE=1): become one.  Checked BEFORE torch is
        # imported or a device is touched.
        if args.scaling == "strong" and NY % args.gpus:
            raise SystemExit(f"--scaling strong needs {NY} % N == 0 (N = {args.gpus}); use N in 1,2,3,4,5,6,8,... or --scaling weak")
        sys.exit(launch_workers(args, sys.argv[1:]))

    # Whatever a library prints on stdout (RCCL writes a version banner there when a communicator comes up) must not reach OUR stdout:
    # the contract is ONE JSON line.  File descriptor 1 is pointed at stderr for the rest of the process; the line goes to a saved copy.
    contract_out = os.fdopen(os.dup(1), "w")
    sys.stdout.flush()
    os.dup2(2, 1)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")         # dmabuf IPC only on these hosts (RCCL's P2P set-up); read when HSA initialises
    if not libraries_built():
        if int(os.environ.get("LOCAL_RANK", "0")) == 0:            # fresh checkout: build once (hipcc, gcc)
            import __graft_entry__
            __graft_entry__.build()
        else:                                                       # the Makefile renames the finished libraries into place
            while not libraries_built():
                time.sleep(0.5)
    world = int(os.environ.get("WORLD_SIZE", "1"))                  # processes = GPUs
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} != WORLD_SIZE {world}")
    if world > 1 or args.loopback:
        import bench_chain
        sys.exit(bench_chain.run(args, contract_out))
    line = run_single(args)
    contract_out.write(json.dumps(line) + "\n")              # ASCII-escaped: safe under any stdout encoding
    contract_out.flush()


def run_single(args):
    """N = 1: the step on cuda:LOCAL_RANK, every figure of the contract line"""
    import torch
    import orthogonalsphericalshellgrids.jl_amd as osg
    from orthogonalsphericalshellgrids.jl_amd import _lib
    from tools import testlib                                       # synthetic fill + copy probe only; every step call is the product library's
    import bench_aux

    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a HIP device"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    lib, tlib = _lib.lib(), testlib.lib()

    # ---- resident inputs / outputs -------------------------------------------------------------
    p = _lib.TpgParams(NX, NY, NZ, H, H, H, -80.0, 55.0, 70.0, osg.R_Earth, _lib.TPG_F64, 1, NY, 0)
    out = [torch.empty((NY + 2 * H, NX + 2 * H), dtype=torch.float64, device=dev) for _ in _lib.ARRAY_NAMES]
    out_ptrs = _lib.ptr_table(out)
    ws = torch.empty(int(lib.tpg_build_grid_workspace_bytes(C.byref(p))), dtype=torch.uint8, device=dev)
    fields = []
    for fid, _ in enumerate(SPECS):
        f = torch.empty((NZ + 2 * H, NY + 2 * H, NX + 2 * H), dtype=torch.float64, device=dev)
        testlib.check(tlib.tpg_fill_synthetic(f.data_ptr(), 0x5EED + fid, 12345.0, NX, NY, NZ, H, H, H, _lib.TPG_F64, None))
        fields.append(f)
    fptrs = _lib.ptr_table(fields)
    n = len(SPECS)
    xl = (C.c_int8 * n)(*[s[1] for s in SPECS]); yl = (C.c_int8 * n)(*[s[2] for s in SPECS]); sg = (C.c_int32 * n)(*[s[3] for s in SPECS])
    geom = (NX, NY, NZ, H, H, H)
    stream = _lib.current_stream_ptr(dev)
    ev = lambda: torch.cuda.Event(enable_timing=True)

    def hip_event():
        e = C.c_void_p()
        _lib.check(lib.tpg_event_create(C.byref(e)))
        return e

    def elapsed_ms(e0, e1):
        ms = C.c_float()
        _lib.check(lib.tpg_event_elapsed_ms(e0, e1, C.byref(ms)))
        return ms.value

    def build():
        _lib.check(lib.tpg_build_grid(C.byref(p), out_ptrs, ws.data_ptr(), ws.numel(), stream))

    def fill(kev=None):
        """fill_halo_regions!: zipper -> periodic x, one merged launch; kev = the kernel's own start/stop events"""
        if kev is not None:
            _lib.check(lib.tpg_fill_halo_regions_timed(fptrs, n, xl, yl, sg, *geom, 1, _lib.TPG_F64, stream, kev[0], kev[1]))
        else:
            _lib.check(lib.tpg_fill_halo_regions(fptrs, n, xl, yl, sg, *geom, 1, _lib.TPG_F64, stream))

    def step(kev=None):
        build()
        fill(kev)

    torch.cuda.synchronize()

    # ---- declared clock pre-roll (not steps) ---------------------------------------------------------------------------------
    # The FP64-heavy cell kernel starts a power-management transient whenever it follows lighter work -- 535 us on its first launch, up
    # to 690 us a few launches later, its steady 490 us only after ~25 ms of sustained FP64 load (profiles/r03/cells_sequence_driver_args.txt).
    # A short run (`--steps 20 --warmup 5` = 14 ms) would time exactly that transient.  So the run declares what it does about it: P plain
    # tpg_build_grid calls (default: ~35 ms of them) immediately before the W warm-up steps, reported as `clock_preroll` {builds, ms};
    # `--preroll 0` switches it off.  Nothing else precedes the warm-up: every auxiliary measurement runs AFTER the timed and instrumented
    # passes.  The figure a caller sees on a FIRST build after HBM-bound work is in the line as well: `ms_per_step_cold_onset`.
    preroll_n = args.preroll if args.preroll >= 0 else 64
    preroll = {"builds": preroll_n, "ms": 0.0, "what": "plain tpg_build_grid calls of the timed geometry, back to back, immediately before the warm-up "
                                                       "steps: brings the clocks to the sustained FP64 state (not steps, not timed into `value`)"}
    if preroll_n:
        b0, b1 = ev(), ev()
        b0.record()
        for _ in range(preroll_n):
            build()
        b1.record()
        torch.cuda.synchronize()
        preroll["ms"] = b0.elapsed_time(b1)

    # ---- W warm-up steps, K timed steps ------------------------------------------------------------------------------------
    torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    # Timed region: K steps; the only instrumentation inside it is the fill kernel's own start/stop timestamps (they ride on its
    # dispatch packet).  Stream-marker events between the phases cost ~10 us of queue bubbles each, so the per-phase breakdown is
    # taken in separate, untimed passes below.
    kevs = [(hip_event(), hip_event()) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(kevs[k])
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0

    # ---- untimed instrumented pass: K steps with stream markers around the two phases ----------------------------------------------
    marks = [[ev() for _ in range(3)] for _ in range(args.steps)]
    for k in range(args.steps):
        m = marks[k]
        m[0].record(); build(); m[1].record(); fill(); m[2].record()
    torch.cuda.synchronize()
    avg = lambda a, b: sum(m[a].elapsed_time(m[b]) for m in marks) / len(marks)      # ms
    t_build, t_fill_bracket = avg(0, 1), avg(1, 2)
    fill_live = [elapsed_ms(e0, e1) for e0, e1 in kevs]
    t_fill_kernel = sum(fill_live) / len(fill_live)                                 # the merged kernel's own duration, timed steps
    for e0, e1 in kevs:
        lib.tpg_event_destroy(e0); lib.tpg_event_destroy(e1)

    # ---- fold-only pass, K old-style steps (build -> tpg_zipper_fill [k_zipper_cols] -> tpg_periodic_x_fill) ----------------------
    fevs = [(hip_event(), hip_event()) for _ in range(args.steps)]
    for k in range(args.steps):
        build()
        _lib.check(lib.tpg_zipper_fill_timed(fptrs, n, xl, yl, sg, *geom, 1, NZ, _lib.TPG_F64, stream, fevs[k][0], fevs[k][1]))
        _lib.check(lib.tpg_periodic_x_fill(fptrs, n, *geom, _lib.TPG_F64, stream))
    torch.cuda.synchronize()
    fold_live = [elapsed_ms(e0, e1) for e0, e1 in fevs]
    fold = sum(fold_live) / len(fold_live)
    for e0, e1 in fevs:
        lib.tpg_event_destroy(e0); lib.tpg_event_destroy(e1)

    # ---- the step right after HBM-bound work: K steps, no pre-roll, no warm-up, after >= 50 ms of streaming traffic ------------------------
    # (what the FIRST builds of a caller cost: the cell kernel's onset transient included.  Untimed region.)
    cold_onset = None
    if not args.no_cold_onset:
        flush = torch.zeros(1 << 27, dtype=torch.float64, device=dev)          # 1 GiB
        torch.cuda.synchronize()
        h0, h1 = ev(), ev()
        h0.record()
        for _ in range(100):                                                    # 100 x (1 GiB read + 1 GiB write) ~ 60 ms
            flush.add_(1.0)
        h1.record()
        torch.cuda.synchronize()
        t0c = time.perf_counter()
        for k in range(args.steps):
            step()
        torch.cuda.synchronize()
        cold_onset = {"ms_per_step": (time.perf_counter() - t0c) / args.steps * 1e3, "steps": args.steps, "preceded_by_ms_of_hbm_bound_work": h0.elapsed_time(h1),
                      "what": f"{args.steps} steps timed like the timed region, but started right after 100 in-place passes over 1 GiB instead of after the "
                              "pre-roll and the warm-up: the cell kernel's power-management onset transient is inside"}
        del flush

    # ---- auxiliary measurements (not steps; after everything that is timed into the contract keys) -------------------------------------------
    aux, fill_step, fill_step_halo5 = {}, None, None
    if not args.no_aux:
        aux = bench_aux.auxiliary(torch, osg, _lib, lib, tlib, testlib, dev, fields, fptrs, xl, yl, sg, geom, p, out, out_ptrs, ws, hip_event, elapsed_ms)
    if not args.no_fill_step:
        del fields, fptrs, out, out_ptrs
        torch.cuda.empty_cache()                                               # config 5 wants 165 GB + headroom
        fill_step = bench_aux.fill_step_config5(torch, osg, _lib, tlib, dev)
        fill_step_halo5 = bench_aux.fill_step_halo5(torch, osg, _lib, lib, tlib, dev)

    # ---- the line ------------------------------------------------------------------------------------------------------------------------------
    ms_per_step = elapsed / args.steps * 1e3
    cells = NX * NY
    zbytes = sum(zipper_algorithmic_bytes(NX, NZ, H).values())
    pbytes = periodic_algorithmic_bytes(NY, NZ, H, n)
    fill_bytes = zbytes + pbytes
    band_cells = (NY + 2 * H) * (NX + 2 * H)
    line = {
        "metric": METRIC,
        "value": cells / (elapsed / args.steps), "unit": "cells/s",
        "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
        "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "TripolarGrid 1/10deg metric precompute (3600x1800, Float64, halo 4) + fill_halo_regions! of 4 fields "
                               "c/u/v/zeta (3600x1800x75): zipper + periodic-x in one merged launch",
                   "global_size": [NX, NY, NZ], "local_size": [NX, NY, NZ], "rows_per_rank": NY, "halo": [H, H, H],
                   "fields": [s[0] for s in SPECS], "parallelism": "latitude-bands x1"},
        "clock_preroll": preroll,
        "ms_per_step_cold_onset": cold_onset["ms_per_step"] if cold_onset else None, "cold_onset": cold_onset,
        "precompute_cells_per_s": cells / (t_build * 1e-3),
        "precompute_ms": t_build, "fill_ms": t_fill_kernel, "fill_bracket_ms": t_fill_bracket,
        "fill_GBps": fill_bytes / (t_fill_kernel * 1e-3) / 1e9,
    }
    line.update(aux)
    if fill_step is not None:
        line["fill_step"] = fill_step
        line["fill_step_halo5"] = fill_step_halo5
    traffic = load_traffic()
    tm = traffic.get("k_fill_merged")
    line["roofline"] = {
        "kernel": "k_fill_merged<double,2,4> (tpg_fill_halo_regions: 4 fields x 83 levels, zipper fold + periodic x, one launch)", "bound": "hbm",
        "achieved": fill_bytes / (t_fill_kernel * 1e-3) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
        "frac": fill_bytes / (t_fill_kernel * 1e-3) / 1e9 / HBM_PEAK_GBPS, "traffic": tm,
        "algorithmic_bytes_per_launch": fill_bytes, "algorithmic_bytes_fold": zbytes, "algorithmic_bytes_periodic_x": pbytes,
        "launch_ms": t_fill_kernel, "measured": "the kernel's own start/stop events on every timed step (hipExtLaunchKernelGGL)",
        "launch_ms_median_min_max": [statistics.median(fill_live), min(fill_live), max(fill_live)],
        "traffic_frac": (tm / (t_fill_kernel * 1e-3) / 1e9 / HBM_PEAK_GBPS) if tm else None,
        "note": "the periodic-x part moves 128 B per row but must fetch and dirty 3 whole 128-B lines per row pair (row pitch 225.5 lines): "
                "counter traffic is ~1.5x algorithmic and the launch sits at the device's line rate (DESIGN.md 6)"}
    line["roofline_fold"] = {
        "kernel": "k_zipper_cols<double,2,4> (tpg_zipper_fill: the fold alone, 4 fields x 75 levels, one launch)", "bound": "hbm",
        "achieved": zbytes / (fold * 1e-3) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
        "frac": zbytes / (fold * 1e-3) / 1e9 / HBM_PEAK_GBPS, "traffic": traffic.get("k_zipper_cols"),
        "algorithmic_bytes_per_launch": zbytes, "launch_ms": fold,
        # the mean is the figure of record; median / min / max show whether a few event pairs are off (seen on one box: mean 19.1 us
        # live against 14.4 us for the same launches in a rocprofv3 trace)
        "launch_ms_median_min_max": [statistics.median(fold_live), min(fold_live), max(fold_live)],
        "measured": f"the kernel's own start/stop events over {args.steps} launches in step context (build -> fold -> periodic x), after the timed steps"}
    if aux:
        line["roofline_fold"]["copy_ceiling_ms"] = aux["zipper_copy_ceiling_ms"]
        line["roofline_fold"]["cold_launch_over_copy_ceiling"] = aux["zipper_cold_ms"] / aux["zipper_copy_ceiling_ms"]
    line["roofline_precompute"] = precompute_roofline(t_build, NX * NY, band_cells, traffic)
    if not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline()
    return line


if __name__ == "__main__":
    main()
