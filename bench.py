#!/usr/bin/env python3
"""bench.py -- TripolarGrid metric precompute + zipper halo fill at 1/10 deg x 75 levels on MI355X.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path over one batch of synthetic input, resident in HBM:
  (1) tpg_build_grid : coordinates + 12 staggered metrics of this rank's latitude band
                       (N = 1: the whole 3600 x 1800 globe, Float64, halo 4) -> 20 padded arrays;
  (2) fill_halo_regions! of the 4 synthetic 3600 x 1800 x 75 Float64 fields c(CC,+1) u(FC,-1)
      v(CF,-1) zeta(FF,+1): zipper fold (ONE batched launch; north rank only) + periodic x
      (+ for N > 1 the y-seam exchange of Hy rows with the neighbour ranks: tpg_halo_exchange_y, RCCL send/recv).
N > 1 is WEAK scaling: every rank keeps a 3600 x 1800 x 75 band of a 3600 x (1800 N) x 75 global
tripolar grid (latitude bands, src/distributed_tripolar_grid.jl); no data-path collective.  For N > 1
the step is ordered zipper -> periodic x -> [seam exchange on a side stream || grid build]: the
exchange only needs the filled fields, the build only writes the grid arrays.

Exactly W untimed warm-up steps, then exactly K timed steps.  The auxiliary measurements the line also carries
(zipper launch duration from cold / warm caches, the same-shape copy ceiling, the config-5 `fill_step`) run BEFORE
the warm-up steps; they are separate measurements, not steps, and they leave the GPU at its steady clocks.

value = horizontal grid cells of all ranks / step time (max over ranks).  `roofline` is the zipper
kernel (the HBM-bound kernel BASELINE.json's north_star sets the 70 % target on); the precompute
kernel, which dominates the step time but is FP64-transcendental bound, is reported beside it.
"""
import argparse
import ctypes as C
import hashlib
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

NX, NY, NZ, H = 3600, 1800, 75, 4
SPECS = [("c", 0, 0, 1), ("u", 1, 0, -1), ("v", 0, 1, -1), ("zeta", 1, 1, 1)]   # name, xloc, yloc, sign
HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FP64_VALU_PEAK_TFLOPS = 78.6    # MI355X vector FP64 (datasheet)
LIB = os.path.join(ROOT, "orthogonalsphericalshellgrids.jl_amd", "libtripolar_hip.so")


def zipper_algorithmic_bytes(nx, nz, hy, specs=SPECS, s=8):
    """CF/FF fields Nx*Nz*Hy*2*s; CC/FC add the row-Ny substitution (Nx/2)*Nz*2*s"""
    per_field = {}
    for name, xl, yl, _ in specs:
        b = nx * nz * hy * 2 * s
        if yl == 0:
            b += (nx // 2) * nz * 2 * s
        per_field[name] = b
    return per_field


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fill-step", action="store_true", help="skip the config-5 (1/24 deg x 100 levels) fill_step measurement")
    ap.add_argument("--no-aux", action="store_true", help="skip the cold/warm zipper and copy-ceiling measurements")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    if not os.path.exists(LIB):
        if int(os.environ.get("LOCAL_RANK", "0")) == 0:            # fresh checkout: build once (hipcc, gcc)
            import __graft_entry__
            __graft_entry__.build()
        else:                                                       # the Makefile renames the finished library into place
            while not os.path.exists(LIB):
                time.sleep(0.5)
    import orthogonalsphericalshellgrids.jl_amd as osg
    from orthogonalsphericalshellgrids.jl_amd import _lib
    from orthogonalsphericalshellgrids.jl_amd.distributed import PendingExchange

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
    comm = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
            # the exchange itself is librccl through the C ABI (tpg_halo_exchange_y); should its communicator fail to come up
            # on ANY rank, every rank falls back to torch.distributed's batch_isend_irecv (also RCCL) and the line says so
            try:
                comm = osg.RcclComm.from_torch()
                comm_error = None
            except Exception as e:                                  # noqa: BLE001
                comm, comm_error = None, f"{type(e).__name__}: {e}"
            ok = torch.tensor([0 if comm is None else 1], dtype=torch.int32, device=dev)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 0 and comm is not None:
                comm.destroy(); comm = None
            if comm is None:
                print(f"[bench rank {rank}] tpg_comm_init_rank unavailable ({comm_error}); seam exchange over torch.distributed", file=sys.stderr)

    lib = _lib.lib()
    gsize = (NX, NY * world, NZ)                                   # weak scaling: 1800 rows per rank
    if world > 1:
        arch = osg.Distributed(osg.GPU(0 if rehearse else local_rank), osg.Partition(y=world), local_rank=rank, rccl_comm=comm)
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

    class BandField:                                                # what the seam exchange needs of a Field
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

    def exchange():
        PendingExchange(band_fields, arch, transport).begin().finish()

    def hip_event():
        e = C.c_void_p()
        _lib.check(lib.tpg_event_create(C.byref(e)))
        return e

    def elapsed_ms(e0, e1):
        ms = C.c_float()
        _lib.check(lib.tpg_event_elapsed_ms(e0, e1, C.byref(ms)))
        return ms.value

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

    def step_overlapped(marks=None, zev=None):
        """N > 1: the halo fill's seam exchange (pack -> RCCL send/recv -> unpack, side stream) runs
        concurrently with the grid build (main stream); the two touch disjoint memory.  The tile kernel of the build is
        ~15 000 short blocks, so RCCL's send/recv workgroups simply take a few wave slots from it.
        marks: [0] start, [1] after the zipper, [2] after periodic x, [3] end of the build (main stream),
               [4] / [5] exchange start / end (side stream)"""
        if marks is not None: marks[0].record()
        zipper(zev)
        if marks is not None: marks[1].record()
        _lib.check(lib.tpg_periodic_x_fill(fptrs, n, *geom, _lib.TPG_F64, stream))
        if marks is not None: marks[2].record()
        side_stream.wait_stream(main_stream)
        with torch.cuda.stream(side_stream):
            if marks is not None: marks[4].record()
            exchange()
            if marks is not None: marks[5].record()
        _lib.check(lib.tpg_build_grid(C.byref(p), out_ptrs, ws.data_ptr(), ws.numel(), stream))
        if marks is not None: marks[3].record()
        main_stream.wait_stream(side_stream)

    def step_serial_exchange(marks=None, zev=None):
        """N > 1 without overlap (TPG_BENCH_OVERLAP=0): same work, one stream; marks [4] / [5] bracket the exchange"""
        if marks is not None: marks[0].record()
        zipper(zev)
        if marks is not None: marks[1].record()
        _lib.check(lib.tpg_periodic_x_fill(fptrs, n, *geom, _lib.TPG_F64, stream))
        if marks is not None: marks[2].record(); marks[4].record()
        exchange()
        if marks is not None: marks[5].record()
        _lib.check(lib.tpg_build_grid(C.byref(p), out_ptrs, ws.data_ptr(), ws.numel(), stream))
        if marks is not None: marks[3].record()

    overlap = world > 1 and os.environ.get("TPG_BENCH_OVERLAP", "1") != "0"
    step = step_overlapped if overlap else (step_serial_exchange if world > 1 else step_serial)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- auxiliary measurements (not steps): zipper launch duration by cache state, copy ceiling, config-5 fills ----------
    aux = {}
    if world == 1 and not args.no_aux:
        flush = torch.zeros(1 << 27, dtype=torch.float64, device=dev)          # 1 GiB: evicts L2 + Infinity Cache
        e0, e1 = hip_event(), hip_event()
        acc = {"cold_dirty": [], "cold_clean": [], "warm": [], "copy_cold_clean": []}
        for it in range(22):
            flush.add_(1.0)                                                     # predecessor leaves the caches full of dirty lines
            zipper((e0, e1)); acc["cold_dirty"].append(elapsed_ms(e0, e1))
            flush.sum()                                                         # ... full of clean lines
            zipper((e0, e1)); acc["cold_clean"].append(elapsed_ms(e0, e1))
            zipper((e0, e1)); acc["warm"].append(elapsed_ms(e0, e1))            # back-to-back relaunch (Infinity-Cache resident)
            flush.sum()
            _lib.check(lib.tpg_zipper_copy_probe(fptrs, n, yl, *geom, _lib.TPG_F64, stream, e0, e1))
            acc["copy_cold_clean"].append(elapsed_ms(e0, e1))
        med = {k: statistics.median(v[2:]) for k, v in acc.items()}             # first 2 rounds dropped
        aux = {"zipper_cold_ms": med["cold_clean"], "zipper_cold_dirty_ms": med["cold_dirty"], "zipper_warm_ms": med["warm"],
               "zipper_copy_ceiling_ms": med["copy_cold_clean"],
               "zipper_states_note": "kernel start/stop events, median of 20: after a 1 GiB read-only pass (cold), after a 1 GiB "
                                     "in-place write (cold_dirty), back-to-back relaunch (warm); copy_ceiling = the same launch "
                                     "shape and bytes as a pure copy (tpg_zipper_copy_probe), cold"}
        lib.tpg_event_destroy(e0); lib.tpg_event_destroy(e1)
        del flush
        for fid, f in enumerate(fields):                                        # the copy probe left unfolded halos behind
            _lib.check(lib.tpg_fill_synthetic(f.data_ptr(), 0x5EED + fid, 12345.0, NX, NY, NZ, H, H, H, _lib.TPG_F64, None))
        # BASELINE config 2: the 1/4 degree (1440 x 720) Float64 metric precompute alone, 20 back-to-back builds
        p2 = _lib.TpgParams(1440, 720, 1, H, H, H, -80.0, 55.0, 70.0, osg.R_Earth, _lib.TPG_F64, 1, 720, 0)
        out2 = [torch.empty((720 + 2 * H, 1440 + 2 * H), dtype=torch.float64, device=dev) for _ in _lib.ARRAY_NAMES]
        ptr2 = _lib.ptr_table(out2)
        ws2 = torch.empty(int(lib.tpg_build_grid_workspace_bytes(C.byref(p2))), dtype=torch.uint8, device=dev)
        for _ in range(3):
            _lib.check(lib.tpg_build_grid(C.byref(p2), ptr2, ws2.data_ptr(), ws2.numel(), stream))
        b0, b1 = ev(), ev()
        b0.record()
        for _ in range(20):
            _lib.check(lib.tpg_build_grid(C.byref(p2), ptr2, ws2.data_ptr(), ws2.numel(), stream))
        b1.record(); torch.cuda.synchronize()
        us2 = b0.elapsed_time(b1) / 20 * 1e3
        aux["config2_quarter_degree_build"] = {"size": [1440, 720, 1], "us_per_build": us2, "cells_per_s": 1440 * 720 / (us2 * 1e-6),
                                               "store_GBps": 160.0 * 1448 * 728 / (us2 * 1e-6) / 1e9}
        del out2, ws2
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
    fill_step = None
    if world == 1 and not args.no_fill_step:
        torch.cuda.synchronize()
        torch.cuda.empty_cache()                                    # hand the auxiliary buffers back before sizing 162 GB of fields
        fill_step = fill_step_config5(torch, osg, _lib, dev)

    # ---- W warm-up steps, K timed steps ------------------------------------------------------------------------------------
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
    if world > 1:      # marks: [start, zipper, periodic, build end, exchange start, exchange end]
        t_zip_bracket, t_periodic = avg(0, 1), avg(1, 2)
        t_exchange = avg(4, 5)                                          # pack + send/recv + unpack
        t_build = avg(2, 3) if overlap else avg(5, 3)                   # serial order: the build starts after the exchange
        te = torch.tensor([t_exchange], dtype=torch.float64, device=None if rehearse else dev)
        dist.all_reduce(te, op=dist.ReduceOp.MAX)                       # the slowest rank's seams
        t_exchange = float(te.item())
    else:
        t_build, t_zip_bracket, t_periodic = avg(0, 1), avg(1, 2), avg(2, 3)
    t_zip = t_zip_bracket
    if north_rank:
        tot = 0.0
        for e0, e1 in zevs:
            tot += elapsed_ms(e0, e1)
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
        per_rows = (NY + 2 * H) * (NZ + 2 * H) * n
        line = {
            "metric": "grid-cells/s metric precompute + zipper halo-fill GB/s, 1/10°×75z",
            "value": cells / (elapsed / args.steps), "unit": "cells/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "TripolarGrid 1/10deg metric precompute (3600x1800 per rank, Float64, halo 4) + "
                                   "fill_halo_regions! of 4 fields c/u/v/zeta (3600x1800x75 per rank): zipper + periodic-x"
                                   + (" + RCCL y-seam exchange" if world > 1 else ""),
                       "global_size": list(gsize), "local_size": [NX, NY, NZ], "halo": [H, H, H], "fields": [s[0] for s in SPECS],
                       "parallelism": f"latitude-bands x{world}"},
            "precompute_cells_per_s": NX * NY / (t_build * 1e-3),
            "precompute_ms": t_build, "zipper_ms": t_zip, "zipper_bracket_ms": t_zip_bracket, "periodic_x_ms": t_periodic,
            "overlap": "seam exchange on a side stream, concurrent with the grid build" if overlap else None,
            "exchange_ms": t_exchange,                                  # max over ranks; per seam direction: 4 fields x 9.58 MB
            "exchange_transport": None if world == 1 else ("gloo, host-staged (rehearsal: timings meaningless)" if rehearse
                                                           else ("tpg_halo_exchange_y: librccl ncclSend/ncclRecv group, packed messages" if comm is not None
                                                                 else "torch.distributed batch_isend_irecv (nccl = RCCL), packed messages [fallback]")),
            "seam_GBps_per_direction": (4 * (NX + 2 * H) * H * (NZ + 2 * H) * 8 / (t_exchange * 1e-3) / 1e9) if t_exchange else None,
            "zipper_GBps": zbytes / (t_zip * 1e-3) / 1e9,
            # the periodic pass is bound by the 128-B lines it must touch, not by the bytes it needs from them:
            # per row pair 3 lines fetched + the same 3 dirtied (row pitch 225.5 lines) -> 384 B of line traffic per row
            "periodic_x": {"rows": per_rows, "algorithmic_bytes": per_rows * 2 * H * 2 * 8, "line_bytes": per_rows * 384,
                           "algorithmic_GBps": per_rows * 2 * H * 2 * 8 / (t_periodic * 1e-3) / 1e9,
                           "line_GBps": per_rows * 384 / (t_periodic * 1e-3) / 1e9,
                           "line_frac_of_hbm_peak": per_rows * 384 / (t_periodic * 1e-3) / 1e9 / HBM_PEAK_GBPS},
        }
        line.update(aux)
        if fill_step is not None:
            line["fill_step"] = fill_step
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):                                   # PMC traffic is only valid for the build it was measured on
            with open(tpath) as f:
                tj = json.load(f)
            src = os.path.join(ROOT, "orthogonalsphericalshellgrids.jl_amd", "csrc", "tpg_zipper.hip")
            if os.path.exists(src) and tj.get("zipper_source_sha16") == hashlib.sha256(open(src, "rb").read()).hexdigest()[:16]:
                traffic = tj.get("k_zipper_cols_bytes_per_launch")
        line["roofline"] = {"kernel": "k_zipper_cols<double,2,4> (4 fields x 75 levels, one launch" + (", on the north rank" if world > 1 else "") + ")", "bound": "hbm",
                            "achieved": zbytes / (t_zip * 1e-3) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                            "frac": zbytes / (t_zip * 1e-3) / 1e9 / HBM_PEAK_GBPS, "traffic": traffic,
                            "algorithmic_bytes_per_launch": zbytes, "launch_ms": t_zip}
        if aux:
            line["roofline"]["copy_ceiling_ms"] = aux["zipper_copy_ceiling_ms"]
            line["roofline"]["cold_launch_over_copy_ceiling"] = aux["zipper_cold_ms"] / aux["zipper_copy_ceiling_ms"]
        flops = 2333.0 * NX * (jend - jstart + 1)                           # FP64 add/mul/fma (fma = 2) per cell, PMC-counted
        line["roofline_precompute"] = {
            "kernel": "tpg_build_grid (k_tables + k_cells_tile + k_halos)", "bound": "hbm",
            "achieved": 160.0 * band_cells / (t_build * 1e-3) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": 160.0 * band_cells / (t_build * 1e-3) / 1e9 / HBM_PEAK_GBPS, "traffic": None,
            "algorithmic_bytes_per_launch": 160 * band_cells,
            "fp64_tflops": flops / (t_build * 1e-3) / 1e12, "fp64_peak_tflops": FP64_VALU_PEAK_TFLOPS,
            "fp64_frac": flops / (t_build * 1e-3) / 1e12 / FP64_VALU_PEAK_TFLOPS,
            "note": "FP64-issue bound in practice (VALU busy 91 %%): ~%.1f TFLOP/s of the %.1f TFLOP/s vector FP64 peak at 2.33 kflop/cell (PMC count)"
                    % (flops / (t_build * 1e-3) / 1e12, FP64_VALU_PEAK_TFLOPS)}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line))          # ASCII-escaped: safe under any stdout encoding
    if world > 1:
        dist.barrier()
        if comm is not None:
            comm.destroy()
        dist.destroy_process_group()


def fill_step_config5(torch, osg, _lib, dev):
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
    lib = _lib.lib()
    grid = osg.TripolarGrid(osg.GPU(dev.index), torch.float64, size=size, halo=halo)
    ext = osg.TripolarGrid(osg.GPU(dev.index), torch.float64, size=(Nx, Ny, 1), halo=(halo[0], substeps + 1, halo[2]))
    f3 = (osg.XFaceField(grid), osg.YFaceField(grid), osg.CenterField(grid), osg.CenterField(grid), osg.CenterField(grid))
    f2 = (osg.Field((osg.Center, osg.Center, None), ext), osg.Field((osg.Face, osg.Center, None), ext), osg.Field((osg.Center, osg.Face, None), ext))
    for k, f in enumerate(f3 + f2):
        _lib.check(lib.tpg_fill_synthetic(f.data.data_ptr(), 0xF5 + k, 12345.0, f.Nx, f.Ny, f.Nz, f.Hx, f.Hy, f.Hz, _lib.TPG_F64, None))

    def timed(fn, reps):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3                     # us

    t3 = timed(osg.halo_fill_plan(f3), 10)
    graph = osg.halo_fill_plan(f2).graph(repeat=substeps)
    t2 = timed(graph.replay, 20)
    specs3 = [("u", 1, 0, -1), ("v", 0, 1, -1), ("T", 0, 0, 1), ("S", 0, 0, 1), ("c", 0, 0, 1)]
    zb = sum(zipper_algorithmic_bytes(Nx, Nz, halo[1], specs3).values())
    rows = 5 * (Ny + 2 * halo[1]) * (Nz + 2 * halo[2])
    pb = rows * 2 * halo[0] * 2 * 8
    line_bytes = zb + rows * 384                                    # fold: whole lines anyway; periodic: 3 + 3 lines per row pair
    out = {"workload": "1/24deg (8640x4320x100, halo 4, Float64): tupled fill_halo_regions!((u,v,T,S,c)) [zipper + periodic x] + "
                       f"{substeps} sub-step fills of (eta,U,V) with north halo {substeps + 1} [one fused launch each, one HIP graph]",
           "fields_GB": sum(f.data.numel() for f in f3) * 8 / 1e9,
           "fill3d_us": t3, "substep_fills_us": t2, "substeps": substeps, "total_us": t3 + t2,
           "fill3d_algorithmic_bytes": zb + pb, "fill3d_line_bytes": line_bytes,
           "fill3d_algorithmic_frac_of_hbm_peak": (zb + pb) / (t3 * 1e-6) / 1e9 / HBM_PEAK_GBPS,
           "fill3d_line_frac_of_hbm_peak": line_bytes / (t3 * 1e-6) / 1e9 / HBM_PEAK_GBPS,
           "substep_fill_us_each": t2 / substeps}
    del f3, f2, grid, ext, graph
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    return out


if __name__ == "__main__":
    main()
