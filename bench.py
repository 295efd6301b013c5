#!/usr/bin/env python3
"""bench.py -- TripolarGrid metric precompute + zipper halo fill at 1/10 deg x 75 levels on MI355X.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--scaling strong|weak] [--exchange auto|monolithic|pipelined_1|pipelined_2]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...     (same thing, launcher supplied)
  python bench.py --loopback [--loopback-bands R] [--loopback-band r]                          (1 GPU: the RCCL branch on one rank)

`python bench.py --gpus N` with N > 1 and no launcher in the environment starts its own N workers: the parent -- which never
imports torch and never touches a GPU -- spawns one child per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 /
MASTER_PORT=<free> set), relays rank 0's ONE JSON line to its stdout and exits with the children's status.

One "step" = one pass of the hot path over one batch of synthetic input, resident in HBM:
  (1) tpg_build_grid : coordinates + 12 staggered metrics of this rank's latitude band
                       (N = 1: the whole 3600 x 1800 globe, Float64, halo 4) -> 20 padded arrays;
  (2) fill_halo_regions! of the 4 synthetic Float64 fields c(CC,+1) u(FC,-1) v(CF,-1) zeta(FF,+1), 75 levels, through the
      product's own entry point -- N = 1: tpg_fill_halo_regions (ONE merged launch: zipper fold + periodic x);
      N > 1: tpg_fill_halo_regions_distributed (zipper on the north rank -> periodic x -> RCCL y-seam exchange of Hy rows).

N > 1, default `--scaling strong` = BASELINE config 4: the SAME 3600 x 1800 x 75 globe split into N latitude bands of 1800/N
rows (N = 8: ny = 225, rank 7 owns the zipper; src/distributed_tripolar_grid.jl:36-49,75,143-147); no data-path collective,
point-to-point seams only.  `--scaling weak` keeps 1800 rows per rank of a 3600 x (1800 N) x 75 globe (not a BASELINE config).
For N > 1 the halo fill (local fill + seam exchange) runs on a side stream beside the grid build on the main stream: the
two touch disjoint memory.  Before the warm-up one fill runs under a host-side deadline: a stalled exchange ends the job
with a one-line JSON diagnostic on stderr and a non-zero exit instead of a silent hang.  The seam exchange has two forms with
identical results -- monolithic (pack all -> one RCCL group -> unpack all) and pipelined per field on a second stream
(tpg_halo_exchange_y_pipelined; in stages of 1 and of 2 fields).  The run is made with the MONOLITHIC form first (first contact,
bit-exact seam check, pre-pass, W + K steps, per-phase passes) and rank 0 assembles the line; only then are the pipelined forms
probed (`--exchange auto`, the default): first contact on fresh fields, seam check, pre-pass, per-phase pass; if one beats the
monolithic pre-pass (max over ranks) the W + K steps are run again with it and that is `value` (`ms_per_step_by_form` keeps
both; `exchange_ms_monolithic`, `exchange_ms_pipelined_1`, `exchange_ms_pipelined_2` beside `link_floor_ms`).  A stall or an
error inside the probe costs the probe, not the run: the line of the monolithic form is printed with `pipelined_probe.status`
"stalled" / "error" and every rank leaves with status 0.  `--exchange monolithic|pipelined_k` runs that form alone.
`--loopback` runs that whole N > 1 branch -- communicator bring-up under the watchdog, seam buffers, the one-call distributed
fill in both forms, the side-stream overlap with the build, the instrumented passes -- on ONE GPU: a communicator of one rank
whose south / north peer is the rank itself, for band r of a chain of R (default: band 3 of 8 = an interior band of BASELINE
config 4 with two seams; band R-1 = the zipper band).  Its seam "transfers" are device-local copies by RCCL's own kernels: a
rehearsal of the code path, not a scaling measurement, and the line says so.
Rehearsals for one-GPU boxes (never used by the driver; timings meaningless): TPG_BENCH_REHEARSE=1 (N <= 6 ranks on cuda:0, gloo, the
fallback transport), =shim (the same ranks through the PRODUCTION branch, librccl's entry points served by the test double tools/nccl_shim
behind the test library), =plan (start-up only, no device, any N: launcher, rendezvous, band layout, seam pairing).

Order of a run: set-up -> [N > 1: first seam exchange under a deadline, bit-exact seam check, exchange pre-pass] -> the DECLARED
clock pre-roll (`clock_preroll`: P plain tpg_build_grid calls, ~35 ms, so that a short run starts from the sustained FP64 clock
state; `--preroll 0` = none) -> exactly W untimed warm-up steps -> exactly K timed steps (`value`, `ms_per_step`) -> untimed
instrumented passes (per-phase breakdown, the fold-only pass behind `roofline_fold`) -> K steps started right after >= 50 ms of
HBM-bound work, no pre-roll, no warm-up (`ms_per_step_cold_onset`: what a caller's FIRST builds cost) -> the auxiliary
measurements (fold / fill by cache state, copy ceiling, the 8- and 16-field batched fold, Float32, config 2, geometry, config 5)
-> cpu_baseline.  Nothing optional precedes the timed region: `value` is the same with --no-aux / --no-fill-step.

value = horizontal grid cells of all ranks / step time (max over ranks).  `roofline` is the halo-fill kernel the step launches
(k_fill_merged: the zipper halo fill as the product issues it, HBM-bound); `roofline_fold` is the fold alone (k_zipper_cols, the
kernel BASELINE.json's north_star puts the 70 % target on); the precompute kernel, which dominates the step time but is
FP64-issue bound, is `roofline_precompute`.
"""
import argparse
import ctypes as C
import hashlib
import json
import os
import statistics
import sys
import threading
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


def periodic_algorithmic_bytes(ny, nz, h, nfields, s=8):
    """Oceananigans' periodic west/east fill: 2 Hx elements read + 2 Hx written per row, every row and level of the parent"""
    return nfields * (ny + 2 * h) * (nz + 2 * h) * 2 * h * 2 * s


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


class Watchdog:
    """Host-side deadline for the first contact with the other ranks (communicator bring-up, first seam exchange).  A
    mis-paired or stalled RCCL group blocks either the host (inside ncclGroupEnd) or the device (the stream never drains);
    a timer thread covers both: on expiry it prints ONE JSON line (rank, peers, transport, phase) to stderr and leaves with
    os._exit(3) -- no re-exec, no retry in this process: a fresh child is the only retry."""

    def __init__(self, seconds, info):
        self.seconds, self.info, self.phase, self._timer = seconds, dict(info), "idle", None
        self.soft = None                    # callable: what to do INSTEAD of failing (the pipelined probe: print the line already in hand)

    def _fire(self):
        if self.soft is not None:
            self.soft(self)
            os._exit(0)
        d = dict(self.info, event="bench_deadline_expired", phase=self.phase, deadline_s=self.seconds)
        print(json.dumps(d), file=sys.stderr, flush=True)
        os._exit(3)

    def arm(self, phase):
        self.disarm()
        self.phase = phase
        self._timer = threading.Timer(self.seconds, self._fire)
        self._timer.daemon = True
        self._timer.start()

    def set_phase(self, phase):
        self.phase = phase

    def disarm(self):
        if self._timer is not None:
            self._timer.cancel()
            self._timer = None
        self.phase = "idle"


def chain_layout(world, rank, scaling="strong", loopback=None):
    """The latitude-band chain as every worker derives it from (WORLD_SIZE, RANK) alone -- no device, no torch: one band per process
    (src/distributed_tripolar_grid.jl:36-49: Partition(y = R), rank 0 southernmost; :75,143-147: the last rank owns the zipper) or,
    `loopback = (R, r)`, band r of an emulated chain of R on one process whose peers are the rank itself.  Returns the band count, this
    band, rows per rank and the global row range, the global size, the RCCL peers (-1 = no seam on that side) and who zips."""
    bands, band = loopback if loopback else (world, rank)
    chain = bands > 1
    strong = chain and scaling == "strong"
    if strong and NY % bands:
        # the remainder rule of Oceananigans' local_size for Ny % R != 0 is unpinned: config 4 divides evenly
        raise SystemExit(f"--scaling strong needs {NY} % N == 0 (N = {bands}); use N in 1,2,3,4,5,6,8,... or --scaling weak")
    ny = NY // bands if strong else NY
    gsize = (NX, NY, NZ) if (strong or not chain) else (NX, NY * bands, NZ)
    if loopback:
        south_peer, north_peer = (0 if band > 0 else -1), (0 if band < bands - 1 else -1)
    else:
        south_peer, north_peer = (rank - 1 if rank > 0 else -1), (rank + 1 if rank < world - 1 else -1)
    return {"bands": bands, "band": band, "chain": chain, "strong": strong, "ny": ny, "gsize": gsize,
            "jstart": band * ny + 1, "jend": band * ny + ny, "south_peer": south_peer, "north_peer": north_peer,
            "north_is_zipper": band == bands - 1, "seams": int(south_peer >= 0) + int(north_peer >= 0)}


def plan_rehearsal(args, world, rank, contract_out):
    """TPG_BENCH_REHEARSE=plan: the start-up of an N-rank run WITHOUT a device, for N the box's process guard does not allow on one
    card (at most 6 processes may hold the GPU; BASELINE config 4 has 8).  Every worker runs what the real worker runs before its
    first kernel -- launcher environment, gloo rendezvous on 127.0.0.1, chain_layout, osg.local_row_range on the Distributed
    architecture, the seam plan -- then swaps seam-SHAPED host messages with its neighbours through the product's
    torch_distributed_transport (message [field][level][Hy][Nx+2Hx] of tags naming sender band, side and field) and checks what
    arrived, gathers every rank's record on rank 0 exactly as the real line's `per_rank` travels, and prints ONE line
    {"event": "bench_plan", ...}: a plan, not a measurement -- it carries no metric, value or time."""
    import torch
    import torch.distributed as dist
    import orthogonalsphericalshellgrids.jl_amd as osg
    from orthogonalsphericalshellgrids.jl_amd.distributed import exchange_plan, SOUTH, NORTH
    L = chain_layout(world, rank, args.scaling)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    arch = osg.Distributed(osg.CPU(), osg.Partition(y=L["bands"]), local_rank=L["band"])
    jstart, jend = osg.local_row_range(L["gsize"][1], arch)
    assert (jstart, jend) == (L["jstart"], L["jend"]) and jend - jstart + 1 == L["ny"], (jstart, jend, L)
    plan = exchange_plan(L["band"], L["bands"])
    n = len(SPECS)
    shape = (n, NZ + 2 * H, H, NX + 2 * H)
    tag = lambda b, side: (torch.arange(n, dtype=torch.float64).view(n, 1, 1, 1) + 16.0 * b + 4096.0 * side).expand(shape).contiguous()
    send = {m.side: tag(L["band"], m.side) for m in plan}              # "band b's rows next to `side`"
    recv = {m.side: torch.full(shape, -1.0, dtype=torch.float64) for m in plan}
    osg.torch_distributed_transport(plan, send, recv, None)
    ok = all(torch.equal(recv[m.side], tag(m.peer, NORTH if m.side == SOUTH else SOUTH)) for m in plan)
    mine = {"rank": rank, "band": L["band"], "rows": [jstart, jend], "seams": L["seams"], "zipper": L["north_is_zipper"],
            "peers": {"south": L["south_peer"], "north": L["north_peer"]}, "seam_tags_ok": ok,
            "seam_message_bytes_per_direction": n * (NX + 2 * H) * H * (NZ + 2 * H) * 8}
    per_rank = [None] * world
    dist.all_gather_object(per_rank, mine)
    dist.barrier()
    if rank == 0:
        per_band_hbm = (4 + 1) * (NZ + 2 * H) * (L["ny"] + 2 * H) * (NX + 2 * H) * 8 + 20 * (L["ny"] + 2 * H) * (NX + 2 * H) * 8
        contract_out.write(json.dumps({"event": "bench_plan", "n_gpus": world, "scaling": args.scaling, "global_size": list(L["gsize"]),
                                       "rows_per_rank": L["ny"], "per_rank": per_rank, "hbm_bytes_per_rank": per_band_hbm,
                                       "note": "device-free rehearsal of an N-rank start-up (launcher, rendezvous, band layout, seam pairing "
                                               "over gloo); no kernel ran, nothing was timed"}) + "\n")
        contract_out.flush()
    dist.destroy_process_group()
    return 0 if ok else 8


def free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def launch_workers(args, argv, script=None):
    """`python bench.py --gpus N` without a launcher: start the N workers ourselves.  This parent never imports torch and never
    touches a GPU (no HIP call before or after the spawn; the children are fresh processes, no exec of an initialised one).
    Rank 0's stdout is piped: its one JSON contract line is relayed to our stdout, anything else it prints goes to stderr; the
    other ranks' stdout goes to stderr.  Exit status: 0 only if every child exits 0.  A child that dies takes the job down: the
    survivors get 20 s (their watchdogs may still print a diagnostic), then SIGTERM, then SIGKILL -- by PID."""
    import subprocess
    n = args.gpus
    csrc = os.path.join(ROOT, "orthogonalsphericalshellgrids.jl_amd", "csrc")
    if not (os.path.exists(LIB) and os.path.exists(os.path.join(ROOT, "tools", "libtripolar_hip_test.so"))):
        subprocess.check_call(["make", "-C", csrc, "-j4"], stdout=sys.stderr)          # fresh checkout: hipcc only, no GPU needed
    port = free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), TPG_BENCH_SELF_LAUNCHED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC only on these hosts: RCCL's P2P setup needs it
        env.setdefault("OMP_NUM_THREADS", "1")
        procs.append(subprocess.Popen([sys.executable, script or os.path.abspath(__file__)] + argv, env=env, cwd=ROOT,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=True if r == 0 else None))
    relayed = []

    def relay():
        for ln in procs[0].stdout:
            if ln.lstrip().startswith(('{"metric"', '{"event": "bench_plan"')):
                relayed.append(ln)
                sys.stdout.write(ln); sys.stdout.flush()
            else:
                sys.stderr.write(ln); sys.stderr.flush()

    th = threading.Thread(target=relay, daemon=True)
    th.start()
    first_bad, t_bad = None, None
    while True:
        codes = [p.poll() for p in procs]
        if all(c is not None for c in codes):
            break
        bad = [c for c in codes if c not in (None, 0)]
        if bad and first_bad is None:
            first_bad, t_bad = bad[0], time.time()
            print(f"[bench launcher] a worker exited with status {first_bad}; waiting 20 s for the others", file=sys.stderr, flush=True)
        if first_bad is not None and time.time() - t_bad > 20:
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            time.sleep(5)
            for p in procs:
                if p.poll() is None:
                    p.kill()
        time.sleep(0.2)
    th.join(timeout=10)
    codes = [p.returncode for p in procs]
    rc = next((c for c in codes if c != 0), 0)
    if rc == 0 and len(relayed) != 1:
        print(f"[bench launcher] expected one contract line from rank 0, got {len(relayed)}", file=sys.stderr)
        rc = 5
    return rc if rc >= 0 else 128 - rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--scaling", choices=("strong", "weak"), default="strong",
                    help="N > 1: strong = BASELINE config 4 (the 3600x1800x75 globe in N bands of 1800/N rows; default); weak = 1800 rows per rank")
    ap.add_argument("--exchange", choices=("auto", "monolithic", "pipelined_1", "pipelined_2"), default="auto",
                    help="N > 1: form of the RCCL seam exchange in the timed steps; auto = whichever the pre-pass measures faster (both are always reported)")
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
    ap.add_argument("--no-fill-step", action="store_true", help="skip the config-5 (1/24 deg x 100 levels) fill_step measurement")
    ap.add_argument("--no-cold-onset", action="store_true", help="skip the K cold-onset steps (profiling passes: keeps the 100 streaming launches out of the trace)")
    ap.add_argument("--no-aux", action="store_true", help="skip the auxiliary measurements (cache states, copy ceiling, Float32, config 2, geometry)")
    args = ap.parse_args()

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
    import torch
    import torch.distributed as dist
    if not (os.path.exists(LIB) and os.path.exists(os.path.join(ROOT, "tools", "libtripolar_hip_test.so"))):
        if int(os.environ.get("LOCAL_RANK", "0")) == 0:            # fresh checkout: build once (hipcc, gcc)
            import __graft_entry__
            __graft_entry__.build()
        else:                                                       # the Makefile renames the finished libraries into place
            while not (os.path.exists(LIB) and os.path.exists(os.path.join(ROOT, "tools", "libtripolar_hip_test.so"))):
                time.sleep(0.5)
    import orthogonalsphericalshellgrids.jl_amd as osg
    from orthogonalsphericalshellgrids.jl_amd import _lib
    from orthogonalsphericalshellgrids.jl_amd.distributed import PendingExchange
    from tools import testlib                                       # synthetic fill + copy probe only; every step call is the product library's

    world = int(os.environ.get("WORLD_SIZE", "1"))                  # processes = GPUs
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} != WORLD_SIZE {world}")
    if os.environ.get("TPG_BENCH_REHEARSE") == "plan" and world > 1 and not args.loopback:
        sys.exit(plan_rehearsal(args, world, rank, contract_out))
    # the latitude-band chain: one band per process -- or, --loopback, band r of an emulated chain of R on this one process
    loopback = args.loopback
    L = chain_layout(world, rank, args.scaling, (args.loopback_bands, args.loopback_band) if loopback else None)
    bands, band, chain, strong = L["bands"], L["band"], L["chain"], L["strong"]
    south_peer, north_peer, north_is_zipper = L["south_peer"], L["north_peer"], L["north_is_zipper"]   # RCCL peers; -1 = no seam on that side
    # TPG_BENCH_REHEARSE (one-GPU boxes; never set by the driver): "1" = every rank on cuda:0, seams host-staged over gloo through the FALLBACK
    # transport (comm is None); "shim" = the same, but the PRODUCTION branch (comm is not None: the C ABI's one-call distributed fill in all
    # three exchange forms) runs, its librccl entry points served by the test double tools/nccl_shim (shared-memory mailboxes between the
    # processes) behind the TEST library; "plan" = start-up only, no device (plan_rehearsal above).  Timings of such runs mean nothing.
    shim = os.environ.get("TPG_BENCH_REHEARSE") == "shim" and not loopback
    rehearse = (os.environ.get("TPG_BENCH_REHEARSE") == "1" or shim) and not loopback
    # A node that shows fewer devices than ranks (a short node, a narrowed HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES) must end the job with
    # one readable line, not with N raw "invalid device ordinal" tracebacks.  device_count() does not initialise the GPU.
    visible = int(os.environ["TPG_BENCH_TEST_DEVICE_COUNT"]) if "TPG_BENCH_TEST_DEVICE_COUNT" in os.environ else torch.cuda.device_count()
    if not rehearse and visible < max(local_rank + 1, int(os.environ.get("LOCAL_WORLD_SIZE", world))):
        if rank == 0:
            print(json.dumps({"event": "too_few_devices", "visible": visible, "requested": world, "rank": rank, "local_rank": local_rank,
                              "HIP_VISIBLE_DEVICES": os.environ.get("HIP_VISIBLE_DEVICES"), "ROCR_VISIBLE_DEVICES": os.environ.get("ROCR_VISIBLE_DEVICES"),
                              "hint": "bench.py runs one process per GPU: --gpus N needs N visible HIP devices on this node "
                                      "(a one-GPU rehearsal of the N-rank code path: TPG_BENCH_REHEARSE=1 or =shim, N <= 6; its start-up only, any N: TPG_BENCH_REHEARSE=plan)"}), file=sys.stderr, flush=True)
        sys.exit(7)
    assert torch.cuda.is_available(), f"bench.py needs a HIP device (rank {rank} of {world})"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    peers = {"south": south_peer if south_peer >= 0 else None, "north": north_peer if north_peer >= 0 else None}
    dog = Watchdog(args.deadline, {"rank": rank, "world": world, "peers": peers, "device": local_rank})
    if loopback:
        dog.info.update(loopback={"bands": bands, "band": band})
    # Rehearsal mode for a 1-GPU box (never used by the driver): TPG_BENCH_REHEARSE=1 runs the N-rank
    # code path with every rank on cuda:0 and the seam messages staged through host memory over gloo
    # (RCCL refuses two ranks on one device).  Timings of such a run are meaningless.
    comm, comm_error = None, None
    if chain:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if loopback:
            os.environ.setdefault("MASTER_PORT", str(free_port()))
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
            if shim:
                # the package's and this file's C calls go through the TEST library (same objects as the product + knobs), whose exchange
                # binds the test double instead of librccl; TPG_RCCL_LIBRARY is read at the library's first call, i.e. below
                os.environ.setdefault("TPG_RCCL_LIBRARY", os.path.join(ROOT, "tools", "nccl_shim", "libnccl_shim.so"))
                _lib._lib = testlib.lib()
                dog.info["transport"] = "TEST DOUBLE of librccl (tools/nccl_shim) via tpg_comm_init_rank"
                dog.arm("RcclComm.from_torch over the nccl_shim test double")
                comm = osg.RcclComm.from_torch()
                dog.disarm()
        else:
            dog.info["transport"] = "librccl via tpg_comm_init_rank"
            # the rendezvous waits for the SLOWEST rank's `import torch`, and on a fresh node the first import pages the image in (1-2 minutes,
            # N processes at once): this phase gets its own, longer limit so that a cold start is not mistaken for a stalled exchange
            dog.seconds = max(args.deadline, args.rendezvous_deadline)
            dog.arm("torch.distributed init_process_group(nccl): rendezvous with the other ranks")
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
            # The exchange itself is librccl through the C ABI (tpg_halo_exchange_y).  RcclComm.from_torch first lets every rank report
            # whether it can bind librccl and agrees on that BEFORE the collective ncclCommInitRank; should the communicator still fail
            # to come up on any rank, every rank falls back to torch.distributed's batch_isend_irecv (also RCCL) and the line says so.
            dog.set_phase("RcclComm.from_torch (readiness agreement + ncclCommInitRank)")
            try:
                comm = osg.RcclComm.from_torch()
            except Exception as e:                                  # noqa: BLE001
                comm, comm_error = None, f"{type(e).__name__}: {e}"
            ok = torch.tensor([0 if comm is None else 1], dtype=torch.int32, device=dev)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 0 and comm is not None:
                comm.destroy(); comm = None
            dog.disarm()
            dog.seconds = args.deadline
            if comm is None:
                if loopback:
                    raise SystemExit(f"--loopback needs the C ABI's RCCL communicator: {comm_error}")
                dog.info["transport"] = "torch.distributed batch_isend_irecv"
                print(f"[bench rank {rank}] tpg_comm_init_rank unavailable ({comm_error}); seam exchange over torch.distributed", file=sys.stderr)

    lib, tlib = _lib.lib(), testlib.lib()
    ny, gsize = L["ny"], L["gsize"]                                 # rows of this rank's band, size of the global grid
    if chain:
        arch = osg.Distributed(osg.GPU(0 if rehearse else local_rank), osg.Partition(y=bands), local_rank=band, rccl_comm=comm)
        jstart, jend = osg.local_row_range(gsize[1], arch)
        assert (jstart, jend) == (L["jstart"], L["jend"]) and jend - jstart + 1 == ny, (jstart, jend, L)
    else:
        arch, jstart, jend = osg.GPU(local_rank), 1, NY

    # ---- resident inputs / outputs -------------------------------------------------------------
    p = _lib.TpgParams(gsize[0], gsize[1], gsize[2], H, H, H, -80.0, 55.0, 70.0, osg.R_Earth, _lib.TPG_F64, jstart, jend, 0)
    rows = ny + 2 * H
    out = [torch.empty((rows, NX + 2 * H), dtype=torch.float64, device=dev) for _ in _lib.ARRAY_NAMES]
    out_ptrs = _lib.ptr_table(out)
    ws = torch.empty(int(lib.tpg_build_grid_workspace_bytes(C.byref(p))), dtype=torch.uint8, device=dev)
    shape = (NZ + 2 * H, ny + 2 * H, NX + 2 * H)
    fields = []
    for fid, _ in enumerate(SPECS):
        f = torch.empty(shape, dtype=torch.float64, device=dev)
        testlib.check(tlib.tpg_fill_synthetic(f.data_ptr(), 0x5EED + fid + 16 * band, 12345.0, NX, ny, NZ, H, H, H, _lib.TPG_F64, None))
        fields.append(f)
    fptrs = _lib.ptr_table(fields)
    n = len(SPECS)
    xl = (C.c_int8 * n)(*[s[1] for s in SPECS]); yl = (C.c_int8 * n)(*[s[2] for s in SPECS]); sg = (C.c_int32 * n)(*[s[3] for s in SPECS])
    geom = (NX, ny, NZ, H, H, H)

    class BandField:                                                # what the seam exchange needs of a Field
        def __init__(self, data):
            self.data, self.Nx, self.Ny, self.Nz, self.Hx, self.Hy, self.Hz = data, NX, ny, NZ, H, H, H
    band_fields = [BandField(f) for f in fields]

    stream = _lib.current_stream_ptr(dev)
    ev = lambda: torch.cuda.Event(enable_timing=True)

    transport = None
    if rehearse and chain:
        def transport(plan, send, recv, group):                     # host-staged stand-in for RCCL p2p
            hs = {k: v.cpu() for k, v in send.items()}
            hr = {k: torch.empty_like(v) for k, v in hs.items()}
            osg.torch_distributed_transport(plan, hs, hr, group)
            for k in recv:
                recv[k].copy_(hr[k])

    # seam message buffers: owned here for the C call (RCCL path), by the PendingExchange otherwise
    seam, seam_ptr = None, [None] * 4
    if chain and comm is not None:
        nelem = int(lib.tpg_y_halo_buffer_elems(n, NX, NZ, H, H, H))
        seam = {k: torch.empty(nelem, dtype=torch.float64, device=dev) for k in ("ss", "sn", "rs", "rn")}
        seam_ptr = [seam["ss"].data_ptr() if south_peer >= 0 else None, seam["sn"].data_ptr() if north_peer >= 0 else None,
                    seam["rs"].data_ptr() if south_peer >= 0 else None, seam["rn"].data_ptr() if north_peer >= 0 else None]
    pending = PendingExchange(band_fields, arch, transport) if (chain and comm is None) else None

    def hip_event():
        e = C.c_void_p()
        _lib.check(lib.tpg_event_create(C.byref(e)))
        return e

    def elapsed_ms(e0, e1):
        ms = C.c_float()
        _lib.check(lib.tpg_event_elapsed_ms(e0, e1, C.byref(ms)))
        return ms.value

    main_stream = torch.cuda.current_stream(dev)
    # the halo fill of a distributed step runs on a side stream, the RCCL groups of the pipelined exchange on a third one; both at high
    # priority (TPG_BENCH_SIDE_PRIORITY, default -1): the exchange is the long pole of a band's step, its few workgroups should never queue
    # behind the ~2000 blocks of the build
    prio = int(os.environ.get("TPG_BENCH_SIDE_PRIORITY", "-1"))
    side_stream = torch.cuda.Stream(dev, priority=prio) if chain else None
    comm_stream = torch.cuda.Stream(dev, priority=prio) if (chain and comm is not None) else None
    comm_stream_ptr = C.c_void_p(comm_stream.cuda_stream) if comm_stream is not None else None
    overlap = chain and os.environ.get("TPG_BENCH_OVERLAP", "1") != "0"
    # exchange forms timed on every run: monolithic, and pipelined in stages of 1 and of 2 fields (4 and 2 stages of the 4 bench fields)
    FORMS = ("monolithic", "pipelined_1", "pipelined_2") if comm is not None else ("monolithic",)
    stage_of = lambda form: int(form.split("_")[1])

    def local_fill(kev=None):
        """fill_halo_regions! without the seams: zipper (north band) -> periodic x; kev = the first kernel's own start/stop events"""
        s_ = _lib.current_stream_ptr(dev)
        if kev is not None:
            _lib.check(lib.tpg_fill_halo_regions_timed(fptrs, n, xl, yl, sg, *geom, 1 if north_is_zipper else 0, _lib.TPG_F64, s_, kev[0], kev[1]))
        else:
            _lib.check(lib.tpg_fill_halo_regions(fptrs, n, xl, yl, sg, *geom, 1 if north_is_zipper else 0, _lib.TPG_F64, s_))

    def exchange_only(form="monolithic"):
        s_ = _lib.current_stream_ptr(dev)
        if comm is None:
            pending.begin().finish()
        elif form.startswith("pipelined"):
            _lib.check(lib.tpg_halo_exchange_y_pipelined_peers(comm.handle, south_peer, north_peer, fptrs, n, *seam_ptr, *geom, _lib.TPG_F64,
                                                               s_, comm_stream_ptr, stage_of(form)))
        else:
            _lib.check(lib.tpg_halo_exchange_y_peers(comm.handle, south_peer, north_peer, fptrs, n, *seam_ptr, *geom, _lib.TPG_F64, s_))

    def distributed_fill(form="monolithic"):
        """the whole fill_halo_regions! of a DistributedTripolarGrid: ONE C call on the RCCL path, in either exchange form"""
        s_ = _lib.current_stream_ptr(dev)
        if comm is None:
            local_fill()
            exchange_only()
        elif form.startswith("pipelined"):
            _lib.check(lib.tpg_fill_halo_regions_distributed_pipelined_peers(comm.handle, south_peer, north_peer, 1 if north_is_zipper else 0,
                                                                             fptrs, n, xl, yl, sg, *seam_ptr, *geom, _lib.TPG_F64,
                                                                             s_, comm_stream_ptr, stage_of(form)))
        else:
            _lib.check(lib.tpg_fill_halo_regions_distributed_peers(comm.handle, south_peer, north_peer, 1 if north_is_zipper else 0,
                                                                   fptrs, n, xl, yl, sg, *seam_ptr, *geom, _lib.TPG_F64, s_))

    def build():
        _lib.check(lib.tpg_build_grid(C.byref(p), out_ptrs, ws.data_ptr(), ws.numel(), stream))

    def step_serial(kev=None):
        """N = 1: build -> fill_halo_regions! (one merged launch), one stream"""
        build()
        local_fill(kev)

    used_form = ["monolithic"]

    def step_distributed(kev=None):
        """N > 1: the halo fill (zipper on the north rank -> periodic x -> seam exchange) on a side stream beside the grid
        build on the main stream; the two touch disjoint memory.  The tile kernel of the build is thousands of short blocks,
        so RCCL's send/recv workgroups simply take a few wave slots from it.  TPG_BENCH_OVERLAP=0: same work on one stream."""
        if overlap:
            side_stream.wait_stream(main_stream)
            with torch.cuda.stream(side_stream):
                distributed_fill(used_form[0])
            build()
            main_stream.wait_stream(side_stream)
        else:
            distributed_fill(used_form[0])
            build()

    step = step_distributed if chain else step_serial

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def reduce_max(x):
        if world == 1:
            return x
        tt = torch.tensor([x], dtype=torch.float64, device=None if rehearse else dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item())

    # the synthetic fields were written on the NULL stream; the side / comm streams are non-blocking streams and do not wait for it
    torch.cuda.synchronize()

    # ---- N > 1 helpers ---------------------------------------------------------------------------------------------------------------
    # The exchange form of record is `primary`: monolithic, unless --exchange names another.  The other forms are probed in an EPILOGUE, after
    # everything the contract line needs has been measured with the primary form and the line has been assembled: the pipelined forms have
    # never met a second RCCL rank, and a stall in one of them must cost their figures, not the run (see `pipelined probe` below).
    primary = "monolithic" if args.exchange == "auto" else args.exchange
    if chain and primary not in FORMS:
        raise SystemExit(f"--exchange {args.exchange}: not available on this transport ({dog.info.get('transport')})")

    def first_contact(form):
        dog.arm(f"first seam exchange ({form}): enqueue (host inside ncclGroupEnd / batch_isend_irecv)")
        with torch.cuda.stream(side_stream):
            distributed_fill(form)
        dog.set_phase(f"first seam exchange ({form}): device (stream not drained: a peer never posted its half of the group?)")
        torch.cuda.synchronize()
        if world > 1:
            dog.set_phase(f"barrier after the first seam exchange ({form})")
            dist.barrier()
        dog.disarm()

    def verify_seams():
        """(every rank's seams bit-exact?, this rank's record): collective"""
        chk = {"sides": 0, "fields": n, "bit_exact": True, "bad": []}
        scratch = torch.empty(shape, dtype=torch.float64, device=dev)
        for side, nb in (("south", band - 1), ("north", band + 1)):
            if not (0 <= nb < bands):
                continue
            chk["sides"] += 1
            owner = band if loopback else nb                          # loop-back: the "neighbour" on either side is this band itself
            for fid, (name, fxl, fyl, fsg) in enumerate(SPECS):
                testlib.check(tlib.tpg_fill_synthetic(scratch.data_ptr(), 0x5EED + fid + 16 * owner, 12345.0, NX, ny, NZ, H, H, H, _lib.TPG_F64, None))
                one = _lib.ptr_table([scratch])
                _lib.check(lib.tpg_fill_halo_regions(one, 1, (C.c_int8 * 1)(fxl), (C.c_int8 * 1)(fyl), (C.c_int32 * 1)(fsg), *geom,
                                                     1 if owner == bands - 1 else 0, _lib.TPG_F64, None))
                torch.cuda.synchronize()
                # a neighbour sends the interior rows next to the shared seam.  Loop-back with ONE seam (an end band of the emulated
                # chain): the rank's only send (its own rows next to that side) pairs with its only receive (the halo of that side)
                one_seam_loop = loopback and (south_peer < 0 or north_peer < 0)
                if side == "south":                                   # my halo rows j = 1-Hy..0  <-  its interior rows j = ny-Hy+1..ny
                    got, want = fields[fid][:, :H], (scratch[:, H:2 * H] if one_seam_loop else scratch[:, ny:ny + H])
                else:                                                 # my halo rows j = ny+1..ny+Hy  <-  its interior rows j = 1..Hy
                    got, want = fields[fid][:, ny + H:], (scratch[:, ny:ny + H] if one_seam_loop else scratch[:, H:2 * H])
                if not torch.equal(got, want):
                    chk["bit_exact"] = False
                    ne = (got != want).nonzero()
                    chk["bad"].append({"side": side, "field": name, "cells": int(ne.shape[0]),
                                       "first_level_row_col": ne[0].tolist(), "last_level_row_col": ne[-1].tolist()})
        del scratch
        agree = torch.tensor([1 if chk["bit_exact"] else 0], dtype=torch.int32, device=None if rehearse else dev)
        if world > 1:
            dist.all_reduce(agree, op=dist.ReduceOp.MIN)
        return int(agree.item()) == 1, chk

    def prepass_time(form):
        """fill + exchange alone on the side stream, no build beside it: 6 back-to-back fills after 2 untimed ones, max over ranks"""
        sync()
        with torch.cuda.stream(side_stream):
            for _ in range(2):
                distributed_fill(form)
            e0, e1 = ev(), ev()
            e0.record()
            for _ in range(6):
                distributed_fill(form)
            e1.record()
        torch.cuda.synchronize()
        return reduce_max(e0.elapsed_time(e1) / 6)

    def instrument_form(form, with_kernel_events):
        """K x [local fill, exchange] alone on the side stream with an event pair around each part: (local ms, exchange ms, fill kernel ms)"""
        marks = [[ev() for _ in range(3)] for _ in range(args.steps)]
        kev2 = [(hip_event(), hip_event()) for _ in range(args.steps)] if with_kernel_events else None
        sync()
        with torch.cuda.stream(side_stream):
            for k in range(args.steps):
                m = marks[k]
                m[0].record(); local_fill(kev2[k] if kev2 else None); m[1].record(); exchange_only(form); m[2].record()
        sync()
        avg = lambda a, b: sum(m[a].elapsed_time(m[b]) for m in marks) / len(marks)
        tk = 0.0
        if kev2:
            tk = sum(elapsed_ms(e0, e1) for e0, e1 in kev2) / len(kev2)
            for e0, e1 in kev2:
                lib.tpg_event_destroy(e0); lib.tpg_event_destroy(e1)
        return avg(0, 1), avg(1, 2), tk

    def timed_chain_steps():
        """exactly W warm-up + K timed steps of the N > 1 step with the form in used_form[0]; seconds for the K steps, max over ranks"""
        sync()
        for _ in range(args.warmup):
            step()
        sync()
        t0_ = time.perf_counter()
        for _ in range(args.steps):
            step()
        sync()
        return reduce_max(time.perf_counter() - t0_)

    # ---- N > 1: first contact with the neighbours under a deadline ---------------------------------------------------------
    if chain:
        dog.info.update(geometry=list(geom), seam_message_MB=4 * (NX + 2 * H) * H * (NZ + 2 * H) * 8 / 1e6)
        if os.environ.get("TPG_BENCH_TEST_STALL_RANK") == str(rank):      # tests/test_gpu_bench_contract.py: a rank that never posts its half
            time.sleep(3 * args.deadline)
            os._exit(4)
        first_contact(primary)
        # ---- the seams just exchanged, checked bit for bit.  Every field is synthetic with a seed that names its band, so this rank can
        # REBUILD what its neighbour owns: the neighbour's field, its local fill (periodic x; the zipper if it is the north band), and from
        # it the interior rows the neighbour sent.  They must equal the halo rows this rank received -- all columns incl. the x halos, all
        # levels incl. the z halos.  On the driver's multi-GPU run this is the first bit-exact check of the RCCL path between real ranks; a
        # mismatch ends the job (all ranks agree first, so nobody is left in a barrier) with a diagnostic and no contract line.
        dog.arm("seam verification after the first exchange")
        if os.environ.get("TPG_BENCH_TEST_CORRUPT_SEAM") == str(rank):     # tests/test_gpu_bench_contract.py: the check must have teeth
            fields[1][NZ // 2, (H - 1) if south_peer >= 0 else (ny + H), NX // 2] += 1.0
        all_ok, seam_check = verify_seams()
        if not seam_check["bit_exact"]:
            print(json.dumps(dict(dog.info, event="seam_mismatch", **seam_check)), file=sys.stderr, flush=True)
        if not all_ok:
            dog.disarm()
            os._exit(6)                                               # every rank leaves: the exchange delivered wrong halos somewhere
        dog.disarm()
        # the rest of the run (warm-up, timed and instrumented steps: a few seconds) stays under a generous second deadline, so that
        # an exchange that stalls LATER also ends with a diagnostic instead of the driver's kill
        dog.seconds = max(10 * args.deadline, 600.0)
        dog.arm("exchange pre-pass / warm-up / timed / instrumented steps (a seam exchange after the first one never completed)")

    # ---- N > 1: the exchange form of record (fill + exchange alone, pre-pass figure; the other forms follow in the epilogue) -----------
    prepass = {}
    if chain:
        prepass[primary] = prepass_time(primary)
        used_form[0] = primary

    # ---- declared clock pre-roll (not steps) ---------------------------------------------------------------------------------
    # The FP64-heavy cell kernel starts a power-management transient whenever it follows lighter work -- 535 us on its first launch, up
    # to 690 us a few launches later, its steady 490 us only after ~25 ms of sustained FP64 load (profiles/r03/cells_sequence_driver_args.txt).
    # A short run (`--steps 20 --warmup 5` = 14 ms) would time exactly that transient.  So the run declares what it does about it: P plain
    # tpg_build_grid calls of this rank's band (default: ~35 ms of them) immediately before the W warm-up steps, reported as `clock_preroll`
    # {builds, ms}; `--preroll 0` switches it off.  Nothing else precedes the warm-up: every auxiliary measurement runs AFTER the timed and
    # instrumented passes, so `value` does not depend on --no-aux / --no-fill-step.  The figure a caller sees on a FIRST build after
    # HBM-bound work is in the line as well: `ms_per_step_cold_onset` (below).
    preroll_n = args.preroll if args.preroll >= 0 else 64 * (bands if strong else 1)
    preroll = {"builds": preroll_n, "ms": 0.0, "what": "plain tpg_build_grid calls of the timed geometry, back to back, immediately before the warm-up "
                                                       "steps: brings the clocks to the sustained FP64 state (not steps, not timed into `value`)"}
    if preroll_n:
        sync()
        b0, b1 = ev(), ev()
        b0.record()
        for _ in range(preroll_n):
            build()
        b1.record()
        torch.cuda.synchronize()
        preroll["ms"] = b0.elapsed_time(b1)

    # ---- W warm-up steps, K timed steps ------------------------------------------------------------------------------------
    sync()
    for _ in range(args.warmup):
        step()
    sync()
    # Timed region: K steps; the only instrumentation inside it is the fill kernel's own start/stop timestamps (they ride on its
    # dispatch packet; N = 1).  Stream-marker events between the phases cost ~10 us of queue bubbles each, so the per-phase
    # breakdown is taken in separate, untimed passes below.
    kevs = [(hip_event(), hip_event()) for _ in range(args.steps)] if not chain else [None] * args.steps
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(kevs[k])
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=None if rehearse else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- untimed instrumented passes ----------------------------------------------------------------------------------------
    # N = 1: K steps with stream markers around the two phases (one stream anyway).
    # N > 1: no marker ever sits on the build's stream between kernels.  (a) K builds back to back on the main stream, ONE event pair
    # around the lot.  (b) per exchange form, K x [local fill, exchange] alone on the side stream with an event pair around each part
    # (hipEventRecord on the side stream only).  The overlap of (a) and (b) inside a step is then read off the timed steps.
    ex_ms, per_rank = {}, None
    if not chain:
        marks = [[ev() for _ in range(3)] for _ in range(args.steps)]
        for k in range(args.steps):
            m = marks[k]
            m[0].record(); build(); m[1].record(); local_fill(); m[2].record()
        sync()
        avg = lambda a, b: sum(m[a].elapsed_time(m[b]) for m in marks) / len(marks)      # ms
        t_build, t_fill_bracket, t_exchange, t_fillx = avg(0, 1), avg(1, 2), None, None
        fill_live = [elapsed_ms(e0, e1) for e0, e1 in kevs]
        t_fill_kernel = sum(fill_live) / len(fill_live)                                 # the merged kernel's own duration, timed steps
        for e0, e1 in kevs:
            lib.tpg_event_destroy(e0); lib.tpg_event_destroy(e1)
    else:
        sync()
        b0, b1 = ev(), ev()
        b0.record()
        for _ in range(args.steps):
            build()
        b1.record()
        sync()
        t_build = b0.elapsed_time(b1) / args.steps
        # the same past the power-management transient that follows the onset of the FP64-heavy cell kernel (~25 ms of sustained
        # load; with few steps everything above sits inside it): 300 more builds untimed, then 100 timed
        for _ in range(300):
            build()
        b0, b1 = ev(), ev()
        b0.record()
        for _ in range(100):
            build()
        b1.record()
        sync()
        t_build_steady = b0.elapsed_time(b1) / 100
        local_ms = {}
        local_ms[primary], ex_ms[primary], t_fill_kernel = instrument_form(primary, north_is_zipper)
        own_build, own_build_steady = t_build, t_build_steady
        t_build, t_build_steady = reduce_max(t_build), reduce_max(t_build_steady)
        t_fill_kernel = reduce_max(t_fill_kernel)                                    # only the zipper band has one
        cs = {}                                                                       # the chain's summary for the line (refreshed after the epilogue)

        def refresh_chain_summary():
            """collective: every band's own phase times travel to rank 0 (at N = 8 the interior ranks carry two seams, the end ranks one, and
            only the north rank folds); the figures of the form in used_form[0] become the line's exchange_ms / fill_plus_exchange_ms"""
            uf = used_form[0]
            mine = {"rank": rank, "band": band, "rows": [jstart, jend], "seams_bit_exact": seam_check["bit_exact"], "build_ms": own_build,
                    "build_steady_ms": own_build_steady, "local_fill_ms": local_ms[uf],
                    "exchange_ms": ex_ms[uf], "exchange_ms_by_form": dict(ex_ms), "fill_plus_exchange_ms": local_ms[uf] + ex_ms[uf],
                    "seams": int(south_peer >= 0) + int(north_peer >= 0), "zipper": north_is_zipper}
            pr = [None] * world
            if world > 1:
                dist.all_gather_object(pr, mine)
            else:
                pr = [mine]
            cs.update(per_rank=pr, t_fill_bracket=local_ms[uf], t_exchange=reduce_max(ex_ms[uf]), t_fillx=reduce_max(local_ms[uf] + ex_ms[uf]),
                      ex_ms_max={f: reduce_max(ex_ms[f]) for f in sorted(ex_ms)})

        refresh_chain_summary()
        t_fill_bracket, t_exchange, t_fillx = None, None, None                       # chain: read from `cs`

    # ---- N = 1: fold-only pass, K old-style steps (build -> tpg_zipper_fill [k_zipper_cols] -> tpg_periodic_x_fill) -------
    fold = None
    if not chain:
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
    # (what the FIRST builds of a caller cost: the cell kernel's onset transient included.  Untimed region, N = 1 only.)
    cold_onset = None
    if not chain and not args.no_cold_onset:
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
    aux, fill_step = {}, None
    if not chain and not args.no_aux:
        aux = auxiliary(torch, osg, _lib, lib, tlib, testlib, dev, fields, fptrs, xl, yl, sg, geom, p, out, out_ptrs, ws, hip_event, elapsed_ms)
    if not chain and not args.no_fill_step:
        torch.cuda.empty_cache()                                               # config 5 wants 162 GB + headroom
        fill_step = fill_step_config5(torch, osg, _lib, tlib, dev)

    def make_line(elapsed):
        ms_per_step = elapsed / args.steps * 1e3
        t_fill_bracket_, t_exchange, t_fillx = (cs["t_fill_bracket"], cs["t_exchange"], cs["t_fillx"]) if chain else (t_fill_bracket, None, None)
        ex_ms_max = cs["ex_ms_max"] if chain else {}
        per_rank = cs["per_rank"] if chain else None
        # cells of one step: the whole globe (all bands) -- in loop-back only this band's share of it exists
        cells = gsize[0] * gsize[1] if not loopback else NX * ny
        zbytes = sum(zipper_algorithmic_bytes(NX, NZ, H).values())
        pbytes = periodic_algorithmic_bytes(ny, NZ, H, n)
        fill_bytes = (zbytes if (north_is_zipper or not loopback) else 0) + pbytes      # the zipper band / N = 1
        band_cells = (ny + 2 * H) * (NX + 2 * H)
        jm_lo, jm_hi = max(1, jstart - H), min(gsize[1], jend + H)
        evaluated_cells = NX * (jm_hi - jm_lo + 1)                  # cells the cell kernel computes (the band + its seam halo rows)
        line = {
            "metric": "grid-cells/s metric precompute + zipper halo-fill GB/s, 1/10°×75z",
            "value": cells / (elapsed / args.steps), "unit": "cells/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak" if (not chain or not strong) else "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": ("TripolarGrid 1/10deg metric precompute (3600x1800, Float64, halo 4) + fill_halo_regions! of 4 fields "
                                    "c/u/v/zeta (3600x1800x75): zipper + periodic-x in one merged launch" if not chain else
                                    (f"LOOP-BACK REHEARSAL on one GPU (not a scaling measurement): band {band} of {bands} of " if loopback else "")
                                    + (f"BASELINE config 4: the 1/10deg globe (3600x1800x75, Float64, halo 4) as {bands} latitude bands of {ny} rows: per-band "
                                       "metric precompute + fill_halo_regions! of c/u/v/zeta (zipper on the north rank, periodic-x, RCCL y-seam exchange)"
                                       if strong else
                                       f"weak scaling (not a BASELINE config): {bands} bands of 1800 rows of a 3600x{NY * bands}x75 globe: per-band metric "
                                       "precompute + fill_halo_regions! of c/u/v/zeta (zipper on the north rank, periodic-x, RCCL y-seam exchange)")),
                       "global_size": list(gsize), "local_size": [NX, ny, NZ], "rows_per_rank": ny, "halo": [H, H, H],
                       "fields": [s[0] for s in SPECS], "parallelism": f"latitude-bands x{bands}" + (" (loop-back: one band on one GPU)" if loopback else "")},
            "clock_preroll": preroll,
            "ms_per_step_cold_onset": cold_onset["ms_per_step"] if cold_onset else None, "cold_onset": cold_onset,
            "precompute_cells_per_s": cells / (t_build * 1e-3),            # N > 1: all bands / the slowest rank's build
            "precompute_ms": t_build, "fill_ms": t_fill_kernel, "fill_bracket_ms": t_fill_bracket_,
            "fill_GBps": fill_bytes / (t_fill_kernel * 1e-3) / 1e9 if t_fill_kernel else None,
        }
        if chain:
            seam_bytes = 4 * (NX + 2 * H) * H * (NZ + 2 * H) * 8
            hidden = max(0.0, min(1.0, (t_build + t_fillx - ms_per_step) / max(1e-9, min(t_build, t_fillx))))
            transport_name = ("tpg_fill_halo_regions_distributed(_pipelined)_peers -> TEST DOUBLE of librccl (tools/nccl_shim: shared-memory mailboxes "
                              "between the processes of one GPU; rehearsal of the production branch: timings meaningless)" if shim
                              else "gloo, host-staged (rehearsal: timings meaningless)" if rehearse
                              else ("tpg_fill_halo_regions_distributed(_pipelined)_peers -> librccl ncclSend/ncclRecv groups, packed messages"
                                    if comm is not None else "torch.distributed batch_isend_irecv (nccl = RCCL), packed messages [fallback]"))
            line.update({
                "overlap": "halo fill (local fill + seam exchange) on a side stream, concurrent with the grid build" if overlap else None,
                "exchange_ms": t_exchange,                          # the form the timed steps used; pack + send/recv + unpack, slowest rank
                "exchange_form": used_form[0], "exchange_form_choice": args.exchange,
                "exchange_ms_monolithic": ex_ms_max.get("monolithic"),
                "exchange_ms_pipelined": min((v for f, v in ex_ms_max.items() if f.startswith("pipelined")), default=None),   # the better of the two stage sizes
                "exchange_ms_pipelined_1": ex_ms_max.get("pipelined_1"), "exchange_ms_pipelined_2": ex_ms_max.get("pipelined_2"),     # stages of 1 / 2 fields
                "exchange_prepass_fill_ms": dict(prepass),                # whole fill (local + exchange), back to back, per form: what `auto` chose on
                "link_floor_ms": seam_bytes / 153.6e9 * 1e3,        # one seam direction over one xGMI link at its ~153.6 GB/s spec figure
                # the band build past the cell kernel's power-management transient (300 untimed + 100 timed builds, slowest rank); `precompute_ms`
                # is K builds right after the timed steps, which with few steps still sit inside it
                "precompute_steady_ms": t_build_steady, "precompute_steady_cells_per_s": cells / (t_build_steady * 1e-3),
                "fill_plus_exchange_ms": t_fillx, "exchange_over_build": t_exchange / t_build,
                "overlap_hidden_frac": hidden if overlap else 0.0,  # share of the shorter of (build, fill + exchange) that the step hides
                "exchange_transport": transport_name + (" [loop-back: both peers are this rank, the transfers are device-local]" if loopback else ""),
                "per_rank": per_rank,
                "seam_check": "every rank rebuilt its neighbours' synthetic fields and compared the halo rows it received after the first exchanges "
                              f"(monolithic and pipelined) bit for bit: {n} fields x (Nx + 2Hx) x Hy x (Nz + 2Hz) per seam side; all ranks passed",
                "phase_timing": "build: one event pair around K back-to-back builds (main stream); local fill / exchange: hipEventRecord pairs on the "
                                "side stream in a pass without the build; no marker sits on the build's stream inside a step",
                "seam_message_bytes_per_direction": seam_bytes,
                "seam_GBps_per_direction": seam_bytes / (t_exchange * 1e-3) / 1e9,
                "note": ("loop-back rehearsal of the RCCL branch on one GPU: every code path of an N-rank run executes, no link is involved"
                         if loopback else "no multi-GPU curve exists until the driver runs one: this line is what each N prints")})
            if loopback:
                line["loopback"] = {"bands": bands, "band": band, "south_peer": south_peer, "north_peer": north_peer, "zipper": north_is_zipper}
        line.update(aux)
        if fill_step is not None:
            line["fill_step"] = fill_step
        traffic = load_traffic()
        if not chain:
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
            tz = traffic.get("k_zipper_cols")
            line["roofline_fold"] = {
                "kernel": "k_zipper_cols<double,2,4> (tpg_zipper_fill: the fold alone, 4 fields x 75 levels, one launch)", "bound": "hbm",
                "achieved": zbytes / (fold * 1e-3) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": zbytes / (fold * 1e-3) / 1e9 / HBM_PEAK_GBPS, "traffic": tz,
                "algorithmic_bytes_per_launch": zbytes, "launch_ms": fold,
                # the mean is the figure of record; median / min / max show whether a few event pairs are off (seen on one box: mean 19.1 us
                # live against 14.4 us for the same launches in a rocprofv3 trace)
                "launch_ms_median_min_max": [statistics.median(fold_live), min(fold_live), max(fold_live)],
                "measured": f"the kernel's own start/stop events over {args.steps} launches in step context (build -> fold -> periodic x), after the timed steps"}
            if aux:
                line["roofline_fold"]["copy_ceiling_ms"] = aux["zipper_copy_ceiling_ms"]
                line["roofline_fold"]["cold_launch_over_copy_ceiling"] = aux["zipper_cold_ms"] / aux["zipper_copy_ceiling_ms"]
        else:
            line["roofline"] = {
                "kernel": "k_fill_merged<double,2,4> on the north rank (zipper fold + periodic x of its band, one launch)", "bound": "hbm",
                "achieved": fill_bytes / (t_fill_kernel * 1e-3) / 1e9 if t_fill_kernel else None, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": fill_bytes / (t_fill_kernel * 1e-3) / 1e9 / HBM_PEAK_GBPS if t_fill_kernel else None, "traffic": None,
                "algorithmic_bytes_per_launch": fill_bytes, "launch_ms": t_fill_kernel,
                "measured": "the kernel's own start/stop events, instrumented pass after the timed steps"}
        # the precompute: FP64 VALU issue is what bounds it (VALU busy 91 %), so THAT is `bound` / `frac`; its store stream is secondary.
        # flops are per cell the kernel evaluates (band rows + the seam halo rows it computes), bytes per padded cell it stores
        flops = 2285.0 * evaluated_cells                            # FP64 add/mul/fma (fma = 2) per cell, PMC-counted on the round-3 kernel (profiles/r03/cells_trims.txt)
        tflops = flops / (t_build * 1e-3) / 1e12
        line["roofline_precompute"] = {
            "kernel": "tpg_build_grid (k_tables + k_cells_tile + k_halos)" + (", slowest rank" if world > 1 else ""), "bound": "fp64_valu",
            "achieved": tflops, "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tflops / FP64_VALU_PEAK_TFLOPS,
            "flop_per_cell": 2285.0, "evaluated_cells": evaluated_cells, "stored_cells": band_cells,
            "hbm_GBps": 160.0 * band_cells / (t_build * 1e-3) / 1e9, "hbm_frac": 160.0 * band_cells / (t_build * 1e-3) / 1e9 / HBM_PEAK_GBPS,
            "algorithmic_bytes_per_launch": 160 * band_cells, "traffic": traffic.get("k_cells_tile"),
            "note": "FP64-issue bound (VALU busy 91 %%): %.1f of the %.1f TFLOP/s vector FP64 peak at 2.29 kflop/cell (PMC count); the 160 B/cell store "
                    "stream is %.0f %%%% of HBM peak" % (tflops, FP64_VALU_PEAK_TFLOPS, 100 * 160.0 * band_cells / (t_build * 1e-3) / 1e9 / HBM_PEAK_GBPS)}
        return line

    # ---- N > 1: the pipelined probe (epilogue) -----------------------------------------------------------------------------------------------
    # Everything the line needs has been measured with the primary form, and rank 0 holds the line.  Only now are the other exchange forms
    # tried: first contact on freshly synthesised fields (so that a form that delivers nothing cannot pass on the primary form's halos), the
    # bit-exact seam check, the pre-pass figure, the per-phase pass.  If a probed form's pre-pass (max over ranks) beats the primary's, the W
    # + K steps are run again with it and THAT is `value` (`ms_per_step_by_form` keeps both).  A stall anywhere in here fires the watchdog in
    # its SOFT mode: one JSON diagnostic on stderr, rank 0 prints the line it already holds with `pipelined_probe.status = "stalled"`, every
    # rank leaves with status 0 -- the pipelined forms have never met a second RCCL rank, and their first contact must not cost the run.
    final_line = make_line(elapsed) if rank == 0 else None
    steps_by_form = {used_form[0]: elapsed / args.steps * 1e3} if chain else None
    if chain and rank == 0:
        final_line["ms_per_step_by_form"] = dict(steps_by_form)
        final_line["pipelined_probe"] = {"status": "not run", "why": "no C-ABI communicator" if comm is None else ("--exchange " + args.exchange)}
    if chain and comm is not None and args.exchange == "auto" and len(FORMS) > 1:
        probe = {"status": "ok", "forms": {}}

        def soft_expiry(d):
            print(json.dumps(dict(d.info, event="pipelined_probe_stalled", phase=d.phase, deadline_s=d.seconds)), file=sys.stderr, flush=True)
            if rank == 0:
                final_line["pipelined_probe"] = {"status": "stalled", "phase": d.phase, "forms": probe["forms"],
                                                 "note": "the line is the primary (monolithic) form's; a pipelined form did not complete in time"}
                contract_out.write(json.dumps(final_line) + "\n")
                contract_out.flush()

        try:
            dog.disarm()
            dog.soft, dog.seconds = soft_expiry, args.deadline
            for form in [f for f in FORMS if f != primary]:
                dog.arm(f"pipelined probe ({form}): first exchange on fresh fields")
                if os.environ.get("TPG_BENCH_TEST_STALL_PIPELINED") == str(rank):     # tests: a rank that never enters the probe
                    time.sleep(3 * args.deadline)
                for fid, f in enumerate(fields):
                    testlib.check(tlib.tpg_fill_synthetic(f.data_ptr(), 0x5EED + fid + 16 * band, 12345.0, NX, ny, NZ, H, H, H, _lib.TPG_F64, None))
                torch.cuda.synchronize()
                with torch.cuda.stream(side_stream):
                    distributed_fill(form)
                dog.set_phase(f"pipelined probe ({form}): device (stream not drained)")
                torch.cuda.synchronize()
                if world > 1:
                    dog.set_phase(f"pipelined probe ({form}): barrier after the first exchange")
                    dist.barrier()
                dog.set_phase(f"pipelined probe ({form}): seam verification")
                ok, chk = verify_seams()
                if not ok:
                    probe["forms"][form] = "seam_mismatch"
                    probe["status"] = "seam_mismatch"
                    if not chk["bit_exact"]:
                        print(json.dumps(dict(dog.info, event="seam_mismatch", form=form, **chk)), file=sys.stderr, flush=True)
                    for fid, f in enumerate(fields):                               # leave correct halos behind: the primary form again
                        testlib.check(tlib.tpg_fill_synthetic(f.data_ptr(), 0x5EED + fid + 16 * band, 12345.0, NX, ny, NZ, H, H, H, _lib.TPG_F64, None))
                    torch.cuda.synchronize()
                    with torch.cuda.stream(side_stream):
                        distributed_fill(primary)
                    torch.cuda.synchronize()
                    continue
                dog.arm(f"pipelined probe ({form}): pre-pass and per-phase pass")
                prepass[form] = prepass_time(form)
                local_ms[form], ex_ms[form], _ = instrument_form(form, False)
                probe["forms"][form] = "ok"
            best = min(prepass, key=lambda f: prepass[f])                           # the same on every rank: max-over-ranks values
            if best != primary:
                dog.arm(f"pipelined probe: W + K steps with {best}")
                used_form[0] = best
                steps_by_form[best] = timed_chain_steps() / args.steps * 1e3
            dog.arm("pipelined probe: gathering the ranks' figures")
            refresh_chain_summary()
            dog.disarm()
            dog.soft = None
            if rank == 0:
                final_line = make_line(steps_by_form[used_form[0]] * 1e-3 * args.steps)
                final_line["ms_per_step_by_form"] = dict(steps_by_form)
                final_line["pipelined_probe"] = probe
        except Exception as e:                                          # noqa: BLE001 -- an ERROR in a probed form (not a stall) must not cost the run either
            dog.disarm()
            print(json.dumps(dict(dog.info, event="pipelined_probe_failed", error=f"{type(e).__name__}: {e}"[:500], forms=probe["forms"])),
                  file=sys.stderr, flush=True)
            used_form[0] = primary
            if rank == 0:
                primary_line = make_line(elapsed)                       # the primary form's figures (the summary may be half refreshed: rebuild from `cs`)
                primary_line["ms_per_step_by_form"] = {primary: elapsed / args.steps * 1e3}
                primary_line["pipelined_probe"] = {"status": "error", "error": f"{type(e).__name__}: {e}"[:500], "forms": probe["forms"]}
                contract_out.write(json.dumps(primary_line) + "\n")
                contract_out.flush()
            os._exit(0)                                                 # the other ranks leave through their soft watchdogs
    if rank == 0:
        line = final_line
        if not chain and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
        contract_out.write(json.dumps(line) + "\n")          # ASCII-escaped: safe under any stdout encoding
        contract_out.flush()
    if chain:
        if world > 1:
            dist.barrier()
        dog.disarm()
        # every rank is past the last collective and the line is out: a communicator that will not shut down must not turn the run
        # into a failure (or keep the launcher waiting) -- but it must not pass unseen either: after 30 s the rank says on stderr which
        # call it is stuck in (one JSON line, event "teardown_stalled") and leaves with status 0, the measurement being complete
        pending = ["comm.destroy (ncclCommDestroy)" if comm is not None else "torch.distributed destroy_process_group"]

        def _stalled():
            print(json.dumps({"event": "teardown_stalled", "rank": rank, "world": world, "phase": "teardown", "pending_call": pending[0],
                              "after_s": 30, "exit_status": 0, "note": "the contract line was already written; only the shutdown hung"}),
                  file=sys.stderr, flush=True)
            os._exit(0)

        t_exit = threading.Timer(float(os.environ.get("TPG_BENCH_TEARDOWN_S", "30")), _stalled)
        t_exit.daemon = True
        t_exit.start()
        if os.environ.get("TPG_BENCH_TEST_STALL_TEARDOWN") == str(rank):      # tests: a shutdown that never returns
            time.sleep(3600)
        if comm is not None:
            comm.destroy()
        pending[0] = "torch.distributed destroy_process_group"
        dist.destroy_process_group()
        t_exit.cancel()


AUX_PREROLL = 64


def load_traffic():
    """PMC traffic per kernel (profiles/traffic.json), valid only for the build it was measured on: every kernel entry carries the
    file its kernel lives in and that file's hash at measurement time; an entry whose source has changed since is dropped."""
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.exists(tpath):
        return {}
    with open(tpath) as f:
        tj = json.load(f)
    out, hashes = {}, {}
    for k, v in tj.get("kernels", {}).items():
        srcs = tuple(v.get("sources") or ())
        if not srcs:
            continue
        if srcs not in hashes:
            hashes[srcs] = sources_sha16(srcs)
        if hashes[srcs] is not None and hashes[srcs] == v.get("sources_sha16"):
            out[k] = v.get("hbm_bytes_per_launch")
    return out


def sources_sha16(srcs):
    """one hash over the files (repo-relative) a kernel is compiled from; None if one is missing"""
    h = hashlib.sha256()
    for rel in srcs:
        path = os.path.join(ROOT, rel)
        if not os.path.exists(path):
            return None
        h.update(open(path, "rb").read())
    return h.hexdigest()[:16]


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
        b0, b1 = ev(), ev()
        b0.record()
        for _ in range(20):
            _lib.check(lib.tpg_build_grid(C.byref(p2), ptr2, ws2.data_ptr(), ws2.numel(), stream))
        b1.record(); torch.cuda.synchronize()
        us2 = b0.elapsed_time(b1) / 20 * 1e3
        aux["config2_quarter_degree_build"] = {"size": [1440, 720, 1], "us_per_build": us2, "cells_per_s": 1440 * 720 / (us2 * 1e-6),
                                               "store_GBps": 160.0 * 1448 * 728 / (us2 * 1e-6) / 1e9}
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

    t3 = timed(osg.halo_fill_plan(f3), 10)
    graph = osg.halo_fill_plan(f2).graph(repeat=substeps)
    t2 = timed(graph.replay, 20)
    specs3 = [("u", 1, 0, -1), ("v", 0, 1, -1), ("T", 0, 0, 1), ("S", 0, 0, 1), ("c", 0, 0, 1)]
    zb = sum(zipper_algorithmic_bytes(Nx, Nz, halo[1], specs3).values())
    pb = periodic_algorithmic_bytes(Ny, Nz, halo[0], 5)
    rows = 5 * (Ny + 2 * halo[1]) * (Nz + 2 * halo[2])
    out = {"workload": "1/24deg (8640x4320x100, halo 4, Float64): tupled fill_halo_regions!((u,v,T,S,c)) [one merged launch] + "
                       f"{substeps} sub-step fills of (eta,U,V) with north halo {substeps + 1} [one fused launch each, one HIP graph]",
           "fields_GB": sum(f.data.numel() for f in f3) * 8 / 1e9,
           "fill3d_us": t3, "substep_fills_us": t2, "substeps": substeps, "total_us": t3 + t2,
           "fill3d_algorithmic_bytes": zb + pb,
           "fill3d_algorithmic_frac_of_hbm_peak": (zb + pb) / (t3 * 1e-6) / 1e9 / HBM_PEAK_GBPS,
           # modelled, not counted: the periodic part touches 3 whole 128-B lines per row pair twice (fetch + write-back), the fold whole lines
           "fill3d_modelled_line_ops": (zb // 128) + rows * 3, "fill3d_modelled_lines_per_ns": ((zb // 128) + rows * 3) / (t3 * 1e3),
           "substep_fill_us_each": t2 / substeps}
    del f3, f2, grid, ext, graph
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    return out


if __name__ == "__main__":
    main()
