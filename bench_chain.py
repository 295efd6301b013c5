"""bench_chain.py -- the N > 1 branch of bench.py: BASELINE config 4, the 3600 x 1800 x 75 globe as N latitude bands, one process per GPU
(src/distributed_tripolar_grid.jl:36-49,75,143-147), and its one-GPU rehearsals (--loopback, TPG_BENCH_REHEARSE=1|shim|plan).

`python bench.py --gpus N` lands here: bench.py parses the arguments, points file descriptor 1 at stderr (the contract is ONE JSON line on the
saved copy) and calls `launch_workers` (no launcher in the environment: this parent never imports torch and spawns one worker per GPU) or
`run` (a worker).  Order of a worker's run: device-count preflight -> rendezvous (own deadline) -> RcclComm.from_torch (readiness agreed before
the collective init; every rank falls back to batch_isend_irecv if any fails) -> first fill with the MONOLITHIC exchange under the watchdog ->
bit-exact seam check (each rank rebuilds its neighbours' synthetic fields; a mismatch is exit 6 everywhere, no line) -> pre-pass -> declared
clock pre-roll -> W + K steps [distributed fill on a high-priority side stream || tpg_build_grid] -> instrumented passes -> rank 0 assembles the
line -> the pipelined probe (`--exchange auto`) -> line -> teardown under a limit.  Exit status: 0; 3 deadline expired; 6 seam mismatch at first
contact; 7 too few devices; 9 (PROBE_FAILED) the line was printed but a pipelined form stalled, failed or delivered wrong seams."""
import ctypes as C
import json
import os
import sys
import threading
import time

from bench_common import (HBM_PEAK_GBPS, LIB, METRIC, NX, NY, NZ, H, ROOT, SPECS, libraries_built, load_traffic, periodic_algorithmic_bytes,
                          precompute_roofline, zipper_algorithmic_bytes)

PROBE_FAILED = 9


class Watchdog:
    """Host-side deadline for the first contact with the other ranks (communicator bring-up, first seam exchange).  A
    mis-paired or stalled RCCL group blocks either the host (inside ncclGroupEnd) or the device (the stream never drains);
    a timer thread covers both: on expiry it prints ONE JSON line (rank, peers, transport, phase) to stderr and leaves with
    os._exit(3) -- no re-exec, no retry in this process: a fresh child is the only retry."""

    def __init__(self, seconds, info):
        self.seconds, self.info, self.phase, self._timer = seconds, dict(info), "idle", None
        self.soft = None                    # callable: what to do INSTEAD of failing (the pipelined probe: print the line already in hand)

    def _fire(self):
        if self.soft is not None:                  # the pipelined probe: the line in hand is printed, the status says the probe did not come back
            self.soft(self)
            os._exit(PROBE_FAILED)
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
    if not libraries_built():
        subprocess.check_call(["make", "-C", csrc, "-j4"], stdout=sys.stderr)          # fresh checkout: hipcc only, no GPU needed
    port = free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), TPG_BENCH_SELF_LAUNCHED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC only on these hosts: RCCL's P2P setup needs it
        env.setdefault("OMP_NUM_THREADS", "1")
        procs.append(subprocess.Popen([sys.executable, script or os.path.join(ROOT, "bench.py")] + argv, env=env, cwd=ROOT,
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


def run(args, contract_out):
    """one worker of the N > 1 run (or the --loopback rehearsal of it on one rank); returns the exit status"""
    import torch
    import torch.distributed as dist
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
        return plan_rehearsal(args, world, rank, contract_out)
    # the latitude-band chain: one band per process -- or, --loopback, band r of an emulated chain of R on this one process
    loopback = args.loopback
    L = chain_layout(world, rank, args.scaling, (args.loopback_bands, args.loopback_band) if loopback else None)
    bands, band, strong = L["bands"], L["band"], L["strong"]
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
        return 7
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
    arch = osg.Distributed(osg.GPU(0 if rehearse else local_rank), osg.Partition(y=bands), local_rank=band, rccl_comm=comm)
    jstart, jend = osg.local_row_range(gsize[1], arch)
    assert (jstart, jend) == (L["jstart"], L["jend"]) and jend - jstart + 1 == ny, (jstart, jend, L)

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
    if rehearse:
        def transport(plan, send, recv, group):                     # host-staged stand-in for RCCL p2p
            hs = {k: v.cpu() for k, v in send.items()}
            hr = {k: torch.empty_like(v) for k, v in hs.items()}
            osg.torch_distributed_transport(plan, hs, hr, group)
            for k in recv:
                recv[k].copy_(hr[k])

    # seam message buffers: owned here for the C call (RCCL path), by the PendingExchange otherwise
    seam, seam_ptr = None, [None] * 4
    if comm is not None:
        nelem = int(lib.tpg_y_halo_buffer_elems(n, NX, NZ, H, H, H))
        seam = {k: torch.empty(nelem, dtype=torch.float64, device=dev) for k in ("ss", "sn", "rs", "rn")}
        seam_ptr = [seam["ss"].data_ptr() if south_peer >= 0 else None, seam["sn"].data_ptr() if north_peer >= 0 else None,
                    seam["rs"].data_ptr() if south_peer >= 0 else None, seam["rn"].data_ptr() if north_peer >= 0 else None]
    pending = PendingExchange(band_fields, arch, transport) if comm is None else None

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
    side_stream = torch.cuda.Stream(dev, priority=prio)
    comm_stream = torch.cuda.Stream(dev, priority=prio) if comm is not None else None
    comm_stream_ptr = C.c_void_p(comm_stream.cuda_stream) if comm_stream is not None else None
    overlap = os.environ.get("TPG_BENCH_OVERLAP", "1") != "0"
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

    step = step_distributed

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
    if primary not in FORMS:
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
    prepass = {primary: prepass_time(primary)}
    used_form[0] = primary

    # ---- declared clock pre-roll (not steps) ---------------------------------------------------------------------------------
    # The FP64-heavy cell kernel starts a power-management transient whenever it follows lighter work -- 535 us on its first launch, up
    # to 690 us a few launches later, its steady 490 us only after ~25 ms of sustained FP64 load (profiles/r03/cells_sequence_driver_args.txt).
    # A short run (`--steps 20 --warmup 5` = 14 ms) would time exactly that transient.  So the run declares what it does about it: P plain
    # tpg_build_grid calls of this rank's band (default: ~35 ms of them) immediately before the W warm-up steps, reported as `clock_preroll`
    # {builds, ms}; `--preroll 0` switches it off.  Nothing else precedes the warm-up: every auxiliary measurement runs AFTER the timed and
    # instrumented passes.
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
    # Timed region: K steps, no instrumentation inside it.  Stream-marker events between the phases cost ~10 us of queue bubbles each, so
    # the per-phase breakdown is taken in separate, untimed passes below.
    t0 = time.perf_counter()
    for k in range(args.steps):
        step()
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=None if rehearse else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- untimed instrumented passes ----------------------------------------------------------------------------------------
    # No marker ever sits on the build's stream between kernels.  (a) K builds back to back on the main stream, ONE event pair
    # around the lot.  (b) per exchange form, K x [local fill, exchange] alone on the side stream with an event pair around each part
    # (hipEventRecord on the side stream only).  The overlap of (a) and (b) inside a step is then read off the timed steps.
    ex_ms = {}
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

    def make_line(elapsed):
        ms_per_step = elapsed / args.steps * 1e3
        t_exchange, t_fillx, ex_ms_max = cs["t_exchange"], cs["t_fillx"], cs["ex_ms_max"]
        # cells of one step: the whole globe (all bands) -- in loop-back only this band's share of it exists
        cells = gsize[0] * gsize[1] if not loopback else NX * ny
        zbytes = sum(zipper_algorithmic_bytes(NX, NZ, H).values())
        pbytes = periodic_algorithmic_bytes(ny, NZ, H, n)
        fill_bytes = (zbytes if (north_is_zipper or not loopback) else 0) + pbytes      # the zipper band
        band_cells = (ny + 2 * H) * (NX + 2 * H)
        jm_lo, jm_hi = max(1, jstart - H), min(gsize[1], jend + H)
        evaluated_cells = NX * (jm_hi - jm_lo + 1)                  # cells the cell kernel computes (the band + its seam halo rows)
        seam_bytes = 4 * (NX + 2 * H) * H * (NZ + 2 * H) * 8
        hidden = max(0.0, min(1.0, (t_build + t_fillx - ms_per_step) / max(1e-9, min(t_build, t_fillx))))
        transport_name = ("tpg_fill_halo_regions_distributed(_pipelined)_peers -> TEST DOUBLE of librccl (tools/nccl_shim: shared-memory mailboxes "
                          "between the processes of one GPU; rehearsal of the production branch: timings meaningless)" if shim
                          else "gloo, host-staged (rehearsal: timings meaningless)" if rehearse
                          else ("tpg_fill_halo_regions_distributed(_pipelined)_peers -> librccl ncclSend/ncclRecv groups, packed messages"
                                if comm is not None else "torch.distributed batch_isend_irecv (nccl = RCCL), packed messages [fallback]"))
        line = {
            "metric": METRIC,
            "value": cells / (elapsed / args.steps), "unit": "cells/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "strong" if strong else "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": (f"LOOP-BACK REHEARSAL on one GPU (not a scaling measurement): band {band} of {bands} of " if loopback else "")
                                   + (f"BASELINE config 4: the 1/10deg globe (3600x1800x75, Float64, halo 4) as {bands} latitude bands of {ny} rows: per-band "
                                      "metric precompute + fill_halo_regions! of c/u/v/zeta (zipper on the north rank, periodic-x, RCCL y-seam exchange)"
                                      if strong else
                                      f"weak scaling (not a BASELINE config): {bands} bands of 1800 rows of a 3600x{NY * bands}x75 globe: per-band metric "
                                      "precompute + fill_halo_regions! of c/u/v/zeta (zipper on the north rank, periodic-x, RCCL y-seam exchange)"),
                       "global_size": list(gsize), "local_size": [NX, ny, NZ], "rows_per_rank": ny, "halo": [H, H, H],
                       "fields": [s[0] for s in SPECS], "parallelism": f"latitude-bands x{bands}" + (" (loop-back: one band on one GPU)" if loopback else "")},
            "clock_preroll": preroll,
            "ms_per_step_cold_onset": None, "cold_onset": None,            # N = 1 only
            "precompute_cells_per_s": cells / (t_build * 1e-3),            # all bands / the slowest rank's build
            "precompute_ms": t_build, "fill_ms": t_fill_kernel, "fill_bracket_ms": cs["t_fill_bracket"],
            "fill_GBps": fill_bytes / (t_fill_kernel * 1e-3) / 1e9 if t_fill_kernel else None,
            "overlap": "halo fill (local fill + seam exchange) on a side stream, concurrent with the grid build" if overlap else None,
            "exchange_ms": t_exchange,                              # the form the timed steps used; pack + send/recv + unpack, slowest rank
            "exchange_form": used_form[0], "exchange_form_choice": args.exchange,
            "exchange_ms_monolithic": ex_ms_max.get("monolithic"),
            "exchange_ms_pipelined": min((v for f, v in ex_ms_max.items() if f.startswith("pipelined")), default=None),   # the better of the two stage sizes
            "exchange_ms_pipelined_1": ex_ms_max.get("pipelined_1"), "exchange_ms_pipelined_2": ex_ms_max.get("pipelined_2"),     # stages of 1 / 2 fields
            "exchange_prepass_fill_ms": dict(prepass),                    # whole fill (local + exchange), back to back, per form: what `auto` chose on
            "link_floor_ms": seam_bytes / 153.6e9 * 1e3,            # one seam direction over one xGMI link at its ~153.6 GB/s spec figure
            # the band build past the cell kernel's power-management transient (300 untimed + 100 timed builds, slowest rank); `precompute_ms`
            # is K builds right after the timed steps, which with few steps still sit inside it
            "precompute_steady_ms": t_build_steady, "precompute_steady_cells_per_s": cells / (t_build_steady * 1e-3),
            "fill_plus_exchange_ms": t_fillx, "exchange_over_build": t_exchange / t_build,
            "overlap_hidden_frac": hidden if overlap else 0.0,      # share of the shorter of (build, fill + exchange) that the step hides
            "exchange_transport": transport_name + (" [loop-back: both peers are this rank, the transfers are device-local]" if loopback else ""),
            # the IPC mode RCCL's P2P set-up ran under (these hosts support dmabuf IPC only: INTEGRATION.md 2, host requirements for N > 1)
            "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"),
            "per_rank": cs["per_rank"],
            "seam_check": "every rank rebuilt its neighbours' synthetic fields and compared the halo rows it received after the first exchanges "
                          f"(monolithic and pipelined) bit for bit: {n} fields x (Nx + 2Hx) x Hy x (Nz + 2Hz) per seam side; all ranks passed",
            "phase_timing": "build: one event pair around K back-to-back builds (main stream); local fill / exchange: hipEventRecord pairs on the "
                            "side stream in a pass without the build; no marker sits on the build's stream inside a step",
            "seam_message_bytes_per_direction": seam_bytes,
            "seam_GBps_per_direction": seam_bytes / (t_exchange * 1e-3) / 1e9,
            "note": ("loop-back rehearsal of the RCCL branch on one GPU: every code path of an N-rank run executes, no link is involved"
                     if loopback else "no multi-GPU curve exists until the driver runs one: this line is what each N prints")}
        if loopback:
            line["loopback"] = {"bands": bands, "band": band, "south_peer": south_peer, "north_peer": north_peer, "zipper": north_is_zipper}
        line["roofline"] = {
            "kernel": "k_fill_merged<double,2,4> on the north rank (zipper fold + periodic x of its band, one launch)", "bound": "hbm",
            "achieved": fill_bytes / (t_fill_kernel * 1e-3) / 1e9 if t_fill_kernel else None, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": fill_bytes / (t_fill_kernel * 1e-3) / 1e9 / HBM_PEAK_GBPS if t_fill_kernel else None, "traffic": None,
            "algorithmic_bytes_per_launch": fill_bytes, "launch_ms": t_fill_kernel,
            "measured": "the kernel's own start/stop events, instrumented pass after the timed steps"}
        line["roofline_precompute"] = precompute_roofline(t_build, evaluated_cells, band_cells, load_traffic(), slowest_rank=world > 1)
        return line

    # ---- the pipelined probe (epilogue) ------------------------------------------------------------------------------------------------------
    # Everything the line needs has been measured with the primary form, and rank 0 holds the line.  Only now are the other exchange forms
    # tried: first contact on freshly synthesised fields (so that a form that delivers nothing cannot pass on the primary form's halos), the
    # bit-exact seam check, the pre-pass figure, the per-phase pass.  If a probed form's pre-pass (max over ranks) beats the primary's, the W
    # + K steps are run again with it and THAT is `value` (`ms_per_step_by_form` keeps both).
    # A probe that does not come back clean -- a stall (the watchdog's SOFT mode), an error status, seams that are not bit-exact -- costs the
    # probe's figures, not the measurement: rank 0 still prints the line of the monolithic form with `pipelined_probe.status` "stalled" /
    # "error" / "seam_mismatch" and one JSON diagnostic goes to stderr.  But it is NOT a successful run: a product entry point
    # (tpg_fill_halo_regions_distributed_pipelined) hung, failed or delivered wrong halos, so every rank that sees it leaves with exit status
    # PROBE_FAILED (9) after the line is out, and the launcher passes that status on.
    final_line = make_line(elapsed) if rank == 0 else None
    steps_by_form = {used_form[0]: elapsed / args.steps * 1e3}
    probe_failed = False
    if rank == 0:
        final_line["ms_per_step_by_form"] = dict(steps_by_form)
        final_line["pipelined_probe"] = {"status": "not run", "why": "no C-ABI communicator" if comm is None else ("--exchange " + args.exchange)}
    if comm is not None and args.exchange == "auto" and len(FORMS) > 1:
        probe = {"status": "ok", "forms": {}}

        def soft_expiry(d):
            print(json.dumps(dict(d.info, event="pipelined_probe_stalled", phase=d.phase, deadline_s=d.seconds, exit_status=PROBE_FAILED)), file=sys.stderr, flush=True)
            if rank == 0:
                final_line["pipelined_probe"] = {"status": "stalled", "phase": d.phase, "forms": probe["forms"], "exit_status": PROBE_FAILED,
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
                if os.environ.get("TPG_BENCH_TEST_CORRUPT_PIPELINED_SEAM") == str(rank):    # tests: the probe's check must have teeth too
                    fields[1][NZ // 2, (H - 1) if south_peer >= 0 else (ny + H), NX // 2] += 1.0
                ok, chk = verify_seams()
                if not ok:
                    probe["forms"][form] = "seam_mismatch"
                    probe["status"] = "seam_mismatch"
                    if not chk["bit_exact"]:
                        print(json.dumps(dict(dog.info, event="seam_mismatch", form=form, exit_status=PROBE_FAILED, **chk)), file=sys.stderr, flush=True)
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
            probe_failed = probe["status"] != "ok"                      # seam_mismatch: agreed by all ranks (verify_seams is collective)
            if probe_failed:
                probe["exit_status"] = PROBE_FAILED
            if rank == 0:
                final_line = make_line(steps_by_form[used_form[0]] * 1e-3 * args.steps)
                final_line["ms_per_step_by_form"] = dict(steps_by_form)
                final_line["pipelined_probe"] = probe
        except Exception as e:                                          # noqa: BLE001 -- an ERROR in a probed form costs the probe's figures, not the line
            dog.disarm()
            print(json.dumps(dict(dog.info, event="pipelined_probe_failed", error=f"{type(e).__name__}: {e}"[:500], forms=probe["forms"],
                                  exit_status=PROBE_FAILED)), file=sys.stderr, flush=True)
            used_form[0] = primary
            if rank == 0:
                primary_line = make_line(elapsed)                       # the primary form's figures (the summary may be half refreshed: rebuild from `cs`)
                primary_line["ms_per_step_by_form"] = {primary: elapsed / args.steps * 1e3}
                primary_line["pipelined_probe"] = {"status": "error", "error": f"{type(e).__name__}: {e}"[:500], "forms": probe["forms"], "exit_status": PROBE_FAILED}
                contract_out.write(json.dumps(primary_line) + "\n")
                contract_out.flush()
            os._exit(PROBE_FAILED)                                      # the other ranks leave through their soft watchdogs, with the same status
    if rank == 0:
        contract_out.write(json.dumps(final_line) + "\n")   # ASCII-escaped: safe under any stdout encoding
        contract_out.flush()
    if world > 1:
        dist.barrier()
    dog.disarm()
    # every rank is past the last collective and the line is out: a communicator that will not shut down must not turn the run
    # into a failure (or keep the launcher waiting) -- but it must not pass unseen either: after 30 s the rank says on stderr which
    # call it is stuck in (one JSON line, event "teardown_stalled") and leaves with the status the run had earned, the measurement being complete
    status = PROBE_FAILED if probe_failed else 0
    pending = ["comm.destroy (ncclCommDestroy)" if comm is not None else "torch.distributed destroy_process_group"]

    def _stalled():
        print(json.dumps({"event": "teardown_stalled", "rank": rank, "world": world, "phase": "teardown", "pending_call": pending[0],
                          "after_s": 30, "exit_status": status, "note": "the contract line was already written; only the shutdown hung"}),
              file=sys.stderr, flush=True)
        os._exit(status)

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
    return status
