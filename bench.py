#!/usr/bin/env python3
"""bench.py -- TripolarGrid metric precompute + zipper halo fill at 1/10 deg x 75 levels on MI355X.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path over one batch of synthetic input, resident in HBM:
  (1) tpg_build_grid : coordinates + 12 staggered metrics of this rank's latitude band
                       (N = 1: the whole 3600 x 1800 globe, Float64, halo 4) -> 20 padded arrays;
  (2) fill_halo_regions! of the 4 synthetic 3600 x 1800 x 75 Float64 fields c(CC,+1) u(FC,-1)
      v(CF,-1) zeta(FF,+1): zipper fold (ONE batched launch; north rank only) + periodic x
      (+ for N > 1 the y-seam exchange of Hy rows with the neighbour ranks over RCCL send/recv).
N > 1 is WEAK scaling: every rank keeps a 3600 x 1800 x 75 band of a 3600 x (1800 N) x 75 global
tripolar grid (latitude bands, src/distributed_tripolar_grid.jl); no data-path collective.  For N > 1
the step is ordered zipper -> periodic x -> [seam exchange on a side stream || grid build]: the
exchange only needs the filled fields, the build only writes the grid arrays.

value = horizontal grid cells of all ranks / step time (max over ranks).  `roofline` is the zipper
kernel (the HBM-bound kernel BASELINE.json's north_star sets the 70 % target on); the precompute
kernel, which dominates the step time but is FP64-transcendental bound, is reported beside it.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

NX, NY, NZ, H = 3600, 1800, 75, 4
SPECS = [("c", 0, 0, 1), ("u", 1, 0, -1), ("v", 0, 1, -1), ("zeta", 1, 1, 1)]   # name, xloc, yloc, sign
HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FP64_VALU_PEAK_TFLOPS = 78.6    # MI355X vector FP64 (datasheet)


def zipper_algorithmic_bytes(nx, nz, hy, s=8):
    """CF/FF fields Nx*Nz*Hy*2*s; CC/FC add the row-Ny substitution (Nx/2)*Nz*2*s"""
    per_field = {}
    for name, xl, yl, _ in SPECS:
        b = nx * nz * hy * 2 * s
        if yl == 0:
            b += (nx // 2) * nz * 2 * s
        per_field[name] = b
    return per_field


def cpu_baseline(threads):
    """The oracle (CPU restatement, kind "port") timed on this host on a bounded sample of the same
    workload: the full 3600x1800 grid build, and the 4-field zipper on a (3600, 64, 75) stand-in
    (the fold touches only the top Hy+1 rows of each level, so bytes per level are identical)."""
    import numpy as np
    from oracle import oracle
    oracle.set_threads(threads)
    oracle.build_grid((360, 180, 1))                              # warm the library
    build_reps = 4                                                # ~12 s of CPU work on one core
    t0 = time.perf_counter()
    for _ in range(build_reps):
        oracle.build_grid((NX, NY, 1))
    t_build = (time.perf_counter() - t0) / build_reps
    ny_s = 64
    size, halo = (NX, ny_s, NZ), (H, H, H)
    fields = [np.random.default_rng(i).uniform(-1, 1, (NZ + 2 * H, ny_s + 2 * H, NX + 2 * H)) for i in range(4)]
    for f, (_, xl, yl, sg) in zip(fields, SPECS):
        oracle.zipper_fill(f, xl, yl, sg, size, halo)             # warm-up
    reps = 5
    t0 = time.perf_counter()
    for _ in range(reps):
        for f, (_, xl, yl, sg) in zip(fields, SPECS):
            oracle.zipper_fill(f, xl, yl, sg, size, halo)
    t_zip = (time.perf_counter() - t0) / reps
    zbytes = sum(zipper_algorithmic_bytes(NX, NZ, H).values())
    oracle.set_threads(1)
    return {
        "value": NX * NY / (t_build + t_zip), "unit": "cells/s", "cores": threads, "kind": "port",
        "sample": f"oracle/tpg_oracle.c, {threads} thread(s): full 3600x1800 Float64 grid build (mean of 4: {t_build:.3f} s) + "
                  f"4-field zipper on a 3600x64x75 stand-in ({t_zip * 1e3:.2f} ms, same bytes per level)",
        "precompute_cells_per_s": NX * NY / t_build, "zipper_GBps": zbytes / t_zip / 1e9,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: the GPU needs ~30 ms of work to settle (clock ramp, first touch of the outputs): 3 warm-up steps
    # measure 0.64 ms per step, 50 measure the steady 0.57-0.58 ms that 2000-step runs also show
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    if not os.path.exists(os.path.join(ROOT, "orthogonalsphericalshellgrids.jl_amd", "libtripolar_hip.so")):
        if int(os.environ.get("LOCAL_RANK", "0")) == 0:            # fresh checkout: build once (hipcc, gcc)
            import __graft_entry__
            __graft_entry__.build()
        else:
            while not os.path.exists(os.path.join(ROOT, "orthogonalsphericalshellgrids.jl_amd", "libtripolar_hip.so")):
                time.sleep(1.0)
            time.sleep(2.0)
    import orthogonalsphericalshellgrids.jl_amd as osg
    from orthogonalsphericalshellgrids.jl_amd import _lib
    from orthogonalsphericalshellgrids.jl_amd.distributed import exchange_y_halos

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N > 1 with: python -m torch.distributed.run --nnodes=1 --nproc-per-node N "
                             "--master-addr 127.0.0.1 --master-port P bench.py --gpus N ...")
        raise SystemExit(f"--gpus {args.gpus} != WORLD_SIZE {world}")
    assert torch.cuda.is_available(), "bench.py needs a HIP device"
    rehearse = os.environ.get("TPG_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # Rehearsal mode for a 1-GPU box (never used by the driver): TPG_BENCH_REHEARSE=1 runs the N-rank
    # code path with every rank on cuda:0 and the seam messages staged through host memory over gloo
    # (RCCL refuses two ranks on one device).  Timings of such a run are meaningless.
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    lib = _lib.lib()
    halo = (H, H, H)
    gsize = (NX, NY * world, NZ)                                   # weak scaling: 1800 rows per rank
    if world > 1:
        arch = osg.Distributed(osg.GPU(0 if rehearse else local_rank), osg.Partition(y=world), local_rank=rank)
        jstart, jend = osg.local_row_range(gsize[1], arch)
    else:
        arch, jstart, jend = osg.GPU(local_rank), 1, NY
    north_rank = rank == world - 1

    # ---- resident inputs / outputs -------------------------------------------------------------
    p = _lib.TpgParams(gsize[0], gsize[1], gsize[2], H, H, H, -80.0, 55.0, 70.0, osg.R_Earth, _lib.TPG_F64, jstart, jend, 0)
    rows = jend - jstart + 1 + 2 * H
    out = [torch.empty((rows, NX + 2 * H), dtype=torch.float64, device=dev) for _ in _lib.ARRAY_NAMES]
    out_ptrs = _lib.ptr_table(out)
    ws = torch.empty(int(lib.tpg_build_grid_workspace_bytes(C.byref(p))), dtype=torch.uint8, device=dev)
    shape = (NZ + 2 * H, NY + 2 * H, NX + 2 * H)
    fields = []
    for fid, _ in enumerate(SPECS):
        f = torch.empty(shape, dtype=torch.float64, device=dev)
        _lib.check(lib.tpg_fill_synthetic(f.data_ptr(), 0x5EED + fid + 16 * rank, 12345.0, NX, NY, NZ, H, H, H, _lib.TPG_F64, None))
        fields.append(f)
    fptrs = _lib.ptr_table(fields)
    n = len(SPECS)
    xl = (C.c_int8 * n)(*[s[1] for s in SPECS]); yl = (C.c_int8 * n)(*[s[2] for s in SPECS]); sg = (C.c_int32 * n)(*[s[3] for s in SPECS])
    geom = (NX, NY, NZ, H, H, H)

    class BandField:                                                # what exchange_y_halos needs of a Field
        def __init__(self, data):
            self.data, self.Nx, self.Ny, self.Nz, self.Hx, self.Hy, self.Hz = data, NX, NY, NZ, H, H, H
    band_fields = [BandField(f) for f in fields]

    stream = _lib.current_stream_ptr(dev)
    ev = lambda: torch.cuda.Event(enable_timing=True)

    transport = None
    if rehearse and world > 1:
        def transport(plan, send, recv, group):                     # host-staged stand-in for RCCL p2p
            hs = {k: v.cpu() for k, v in send.items()}
            hr = {k: torch.empty_like(v) for k, v in hs.items()}
            osg.torch_distributed_transport(plan, hs, hr, group)
            for k in recv:
                recv[k].copy_(hr[k])

    def hip_event():
        e = C.c_void_p()
        _lib.check(lib.tpg_event_create(C.byref(e)))
        return e

    def zipper(zev):
        if not north_rank:
            return
        if zev is not None:         # the kernel's own start/stop device timestamps (hipExtLaunchKernelGGL)
            _lib.check(lib.tpg_zipper_fill_timed(fptrs, n, xl, yl, sg, *geom, 1, NZ, _lib.TPG_F64, stream, zev[0], zev[1]))
        else:
            _lib.check(lib.tpg_zipper_fill(fptrs, n, xl, yl, sg, *geom, 1, NZ, _lib.TPG_F64, stream))

    def step_serial(marks=None, zev=None):
        """N = 1: build -> zipper -> periodic x, one stream"""
        if marks is not None: marks[0].record()
        _lib.check(lib.tpg_build_grid(C.byref(p), out_ptrs, ws.data_ptr(), ws.numel(), stream))
        if marks is not None: marks[1].record()
        zipper(zev)
        if marks is not None: marks[2].record()
        _lib.check(lib.tpg_periodic_x_fill(fptrs, n, *geom, _lib.TPG_F64, stream))
        if marks is not None: marks[3].record()

    main_stream = torch.cuda.current_stream(dev)
    side_stream = torch.cuda.Stream(dev) if world > 1 else None
    if world > 1:
        # the build shares the GPU with RCCL's send/recv workgroups.  The default tile kernel is made of
        # ~15 000 short blocks and simply yields them a few wave slots; the marching kernels
        # (TPG_CELLS_VARIANT=2/1) plan ONE resident round and are told to plan it for 90 % of the slots
        os.environ.setdefault("TPG_CELLS_CAPACITY", "0.9")

    def step_overlapped(marks=None, zev=None):
        """N > 1: the halo fill's seam exchange (pack -> RCCL send/recv -> unpack, side stream) runs
        concurrently with the grid build (main stream); the two touch disjoint memory.
        marks: [0] start, [1] after the zipper, [2] after periodic x, [3] end of the build (main stream)"""
        if marks is not None: marks[0].record()
        zipper(zev)
        if marks is not None: marks[1].record()
        _lib.check(lib.tpg_periodic_x_fill(fptrs, n, *geom, _lib.TPG_F64, stream))
        if marks is not None: marks[2].record()
        side_stream.wait_stream(main_stream)
        with torch.cuda.stream(side_stream):
            if marks is not None: marks[4].record()                     # side stream: exchange start / end
            exchange_y_halos(band_fields, arch, transport=transport)
            if marks is not None: marks[5].record()
        _lib.check(lib.tpg_build_grid(C.byref(p), out_ptrs, ws.data_ptr(), ws.numel(), stream))
        if marks is not None: marks[3].record()
        main_stream.wait_stream(side_stream)

    def step_serial_exchange(marks=None, zev=None):
        """N > 1 without overlap (TPG_BENCH_OVERLAP=0): same work, one stream"""
        if marks is not None: marks[0].record()
        zipper(zev)
        if marks is not None: marks[1].record()
        _lib.check(lib.tpg_periodic_x_fill(fptrs, n, *geom, _lib.TPG_F64, stream))
        if marks is not None: marks[2].record()
        exchange_y_halos(band_fields, arch, transport=transport)
        _lib.check(lib.tpg_build_grid(C.byref(p), out_ptrs, ws.data_ptr(), ws.numel(), stream))
        if marks is not None: marks[3].record()

    overlap = world > 1 and os.environ.get("TPG_BENCH_OVERLAP", "1") != "0"
    if world > 1 and not overlap:
        os.environ["TPG_CELLS_CAPACITY"] = "1.0"
    step = step_overlapped if overlap else (step_serial_exchange if world > 1 else step_serial)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # Device wake-up (part of the setup, not of the W warm-up steps): after the idle seconds of imports and
    # allocation the GPU needs ~30 ms of sustained work before its clocks settle; measured per step: 0.64 ms
    # right after idle, 0.57-0.58 ms from ~50 steps on and in 5000-step runs.  Reported as "prewarm_steps".
    PREWARM = 48
    for _ in range(PREWARM):
        step()
    sync()
    for _ in range(args.warmup):
        step()
    sync()
    # Timed region: K steps; the only instrumentation inside it is the zipper kernel's own start/stop
    # timestamps (they ride on its dispatch packet).  Stream-marker events between the phases cost
    # ~10 us of queue bubbles each (kernel trace: 0.3 us between kernels of one call, 9-11 us across a
    # marker), so the per-phase breakdown is taken in a second, untimed pass of the same K steps.
    zevs = [(hip_event(), hip_event()) for _ in range(args.steps)] if north_rank else [None] * args.steps
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(None, zevs[k])
    sync()
    elapsed = time.perf_counter() - t0
    marks = [[ev() for _ in range(6)] for _ in range(args.steps)]
    for k in range(args.steps):
        step(marks[k], None)
    sync()
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=None if rehearse else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    avg = lambda a, b: sum(m[a].elapsed_time(m[b]) for m in marks) / len(marks)      # ms
    t_exchange = None
    if world > 1:      # overlapped step: marks are [start, zipper, periodic, build end, exchange start, exchange end]
        t_zip_bracket, t_rest, t_build = avg(0, 1), avg(1, 2), avg(2, 3)
        if overlap:
            t_exchange = avg(4, 5)                                      # pack + send/recv + unpack on the side stream
            te = torch.tensor([t_exchange], dtype=torch.float64, device=None if rehearse else dev)
            dist.all_reduce(te, op=dist.ReduceOp.MAX)                   # the slowest rank's seams
            t_exchange = float(te.item())
    else:
        t_build, t_zip_bracket, t_rest = avg(0, 1), avg(1, 2), avg(2, 3)
    t_zip = t_zip_bracket
    if north_rank:
        tot = 0.0
        for e0, e1 in zevs:
            ms = C.c_float()
            _lib.check(lib.tpg_event_elapsed_ms(e0, e1, C.byref(ms)))
            tot += ms.value
            lib.tpg_event_destroy(e0); lib.tpg_event_destroy(e1)
        t_zip = tot / len(zevs)                                     # kernel duration, not the bracket
    if world > 1:
        # the zipper runs on the north (last) rank only: ship its launch time to rank 0 for the report
        tz = torch.tensor([t_zip], dtype=torch.float64, device=None if rehearse else dev)
        dist.broadcast(tz, src=world - 1)
        t_zip = float(tz.item())

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        cells = NX * NY * world
        zb = zipper_algorithmic_bytes(NX, NZ, H)
        zbytes = sum(zb.values())
        band_cells = (jend - jstart + 1 + 2 * H) * (NX + 2 * H)
        line = {
            "metric": "grid-cells/s metric precompute + zipper halo-fill GB/s, 1/10°×75z",
            "value": cells / (elapsed / args.steps), "unit": "cells/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "prewarm_steps": PREWARM, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "TripolarGrid 1/10deg metric precompute (3600x1800 per rank, Float64, halo 4) + "
                                   "fill_halo_regions! of 4 fields c/u/v/zeta (3600x1800x75 per rank): zipper + periodic-x"
                                   + (" + RCCL y-seam exchange" if world > 1 else ""),
                       "global_size": list(gsize), "local_size": [NX, NY, NZ], "halo": [H, H, H], "fields": [s[0] for s in SPECS],
                       "parallelism": f"latitude-bands x{world}"},
            "precompute_cells_per_s": NX * NY / (t_build * 1e-3),
            "precompute_ms": t_build, "zipper_ms": t_zip, "zipper_bracket_ms": t_zip_bracket,
            "periodic_x_ms" if world > 1 else "periodic_and_exchange_ms": t_rest,
            "overlap": "seam exchange on a side stream, concurrent with the grid build" if overlap else None,
            "exchange_ms": t_exchange,                                  # max over ranks; per seam direction: 4 fields x 9.58 MB
            "seam_GBps_per_direction": (4 * (NX + 2 * H) * H * (NZ + 2 * H) * 8 / (t_exchange * 1e-3) / 1e9) if t_exchange else None,
            "zipper_GBps": zbytes / (t_zip * 1e-3) / 1e9,
        }
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            with open(tpath) as f:
                traffic = json.load(f).get("k_zipper_cols_bytes_per_launch")
        if True:
            line["roofline"] = {"kernel": "k_zipper_cols<double,2,4> (4 fields x 75 levels, one launch" + (", on the north rank" if world > 1 else "") + ")", "bound": "hbm",
                                "achieved": zbytes / (t_zip * 1e-3) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                "frac": zbytes / (t_zip * 1e-3) / 1e9 / HBM_PEAK_GBPS, "traffic": traffic,
                                "algorithmic_bytes_per_launch": zbytes, "launch_ms": t_zip}
        flops = 2333.0 * NX * (jend - jstart + 1)                           # FP64 add/mul/fma (fma = 2) per cell, PMC-counted
        line["roofline_precompute"] = {
            "kernel": "tpg_build_grid (k_tables + k_cells_tile + k_halos)", "bound": "hbm",
            "achieved": 160.0 * band_cells / (t_build * 1e-3) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": 160.0 * band_cells / (t_build * 1e-3) / 1e9 / HBM_PEAK_GBPS, "traffic": None,
            "algorithmic_bytes_per_launch": 160 * band_cells,
            "fp64_tflops": flops / (t_build * 1e-3) / 1e12, "fp64_peak_tflops": FP64_VALU_PEAK_TFLOPS,
            "fp64_frac": flops / (t_build * 1e-3) / 1e12 / FP64_VALU_PEAK_TFLOPS,
            "note": "FP64-issue bound in practice (VALU busy 88 %%): ~%.1f TFLOP/s of the %.1f TFLOP/s vector FP64 peak at 2.33 kflop/cell (PMC count)"
                    % (flops / (t_build * 1e-3) / 1e12, FP64_VALU_PEAK_TFLOPS)}
        if world == 1 and not args.no_cpu_baseline:
            threads = 1
            line["cpu_baseline"] = cpu_baseline(threads)
        print(json.dumps(line))          # ASCII-escaped: safe under any stdout encoding
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
