#!/usr/bin/env python3
"""GPU box: the C ABI's RCCL seam exchange (tpg_halo_exchange_y_peers) on a communicator of ONE rank whose south and
north peer are the rank itself: the rank's north interior rows arrive in its own south halo rows and vice versa -- the
data path of a seam (pack -> ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd -> unpack, and the pack-free form) on
real hardware, which a 1-GPU box cannot exercise with two ranks (RCCL refuses two ranks on one device).
Prints one JSON line.  Run it as a child process with a timeout: a mis-paired send/recv would hang."""
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import orthogonalsphericalshellgrids.jl_amd as osg
from orthogonalsphericalshellgrids.jl_amd import _lib


def main():
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    lib = _lib.lib()
    # the bootstrap bench.py uses for N > 1: torch.distributed ("nccl" = RCCL) ferries the unique id, librccl makes the communicator
    import socket
    import torch.distributed as dist
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    comm = osg.RcclComm.from_torch()
    out = {"ok": True, "cases": [], "from_torch": [comm.rank, comm.nranks]}
    rng = np.random.default_rng(5)
    for (Nx, Ny, Nz), (Hx, Hy, Hz), nf, dt, tdt in (((48, 40, 3), (4, 4, 2), 3, np.float64, torch.float64),
                                                   ((20, 12, 2), (3, 2, 1), 2, np.float32, torch.float32),
                                                   ((3600, 64, 75), (4, 4, 4), 4, np.float64, torch.float64),
                                                   ((3600, 225, 75), (4, 4, 4), 4, np.float64, torch.float64),      # BASELINE config 4's band
                                                   ((3600, 225, 75), (5, 5, 5), 4, np.float64, torch.float64)):     # ... at the halo of examples/distributed_bickley_jet.jl:23
        shape = (Nz + 2 * Hz, Ny + 2 * Hy, Nx + 2 * Hx)
        ft = 1 if dt == np.float64 else 0
        for packed in (True, False):
            hosts = [rng.uniform(-1, 1, shape).astype(dt) for _ in range(nf)]
            devs = [torch.from_numpy(h).to(dev) for h in hosts]
            ptrs = _lib.ptr_table(devs)
            n = lib.tpg_y_halo_buffer_elems(nf, Nx, Nz, Hx, Hy, Hz)
            bufs = [torch.empty(n, dtype=tdt, device=dev) for _ in range(4)] if packed else None
            bp = [b.data_ptr() for b in bufs] if packed else [None] * 4
            stream = _lib.current_stream_ptr(dev)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            rc = lib.tpg_halo_exchange_y_peers(comm.handle, 0, 0, ptrs, nf, *bp, Nx, Ny, Nz, Hx, Hy, Hz, ft, stream)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) * 1e3
            if rc != 0:
                out["ok"] = False
                out["cases"].append({"size": [Nx, Ny, Nz], "packed": packed, "rc": rc, "error": lib.tpg_last_error().decode()})
                continue
            good = True
            for h, d in zip(hosts, devs):
                want = h.copy()
                want[:, :Hy] = h[:, Ny:Ny + Hy]              # south halo  <- what was sent north (interior rows Ny-Hy+1..Ny)
                want[:, Ny + Hy:] = h[:, Hy:2 * Hy]          # north halo  <- what was sent south (interior rows 1..Hy)
                good = good and np.array_equal(d.cpu().numpy(), want)
            out["ok"] = out["ok"] and good
            # steady-state cost of one exchange call (host + device, 5 back-to-back calls; self loop-back: no link involved)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                lib.tpg_halo_exchange_y_peers(comm.handle, 0, 0, ptrs, nf, *bp, Nx, Ny, Nz, Hx, Hy, Hz, ft, stream)
            torch.cuda.synchronize()
            steady = (time.perf_counter() - t0) / 5 * 1e3
            out["cases"].append({"size": [Nx, Ny, Nz], "nfields": nf, "packed": packed, "bit_exact": good, "first_call_ms": round(ms, 3),
                                 "steady_call_ms": round(steady, 3)})
    # ---- the PIPELINED packed exchange (tpg_halo_exchange_y_pipelined_peers): stages of k fields, the RCCL groups on a second stream;
    #      must deliver exactly what the monolithic form delivers -- one stream and two, k = 1, 2, 3 (ragged last stage), all fields ----
    pcases = []
    comm_stream = torch.cuda.Stream(dev)
    for (Nx, Ny, Nz), (Hx, Hy, Hz), nf, dt, tdt in (((48, 40, 3), (4, 4, 2), 5, np.float64, torch.float64),
                                                   ((20, 12, 2), (3, 2, 1), 3, np.float32, torch.float32),
                                                   ((48, 40, 3), (5, 5, 5), 4, np.float64, torch.float64),
                                                   ((3600, 225, 75), (4, 4, 4), 4, np.float64, torch.float64)):
        shape = (Nz + 2 * Hz, Ny + 2 * Hy, Nx + 2 * Hx)
        ft = 1 if dt == np.float64 else 0
        nbuf = lib.tpg_y_halo_buffer_elems(nf, Nx, Nz, Hx, Hy, Hz)
        big = Nx >= 3600                                                 # config 4's band: 2.2 GB of host data per case, fewer variants
        for fps in ((1, 2) if big else sorted({1, 2, 3, nf})):
            for cs in ((comm_stream,) if big else (comm_stream, None)):
                for south, north in ((0, 0), (0, -1), (-1, 0)):
                    hosts = [rng.uniform(-1, 1, shape).astype(dt) for _ in range(nf)]
                    devs = [torch.from_numpy(h).to(dev) for h in hosts]
                    bufs = [torch.full((nbuf,), float("nan"), dtype=tdt, device=dev) for _ in range(4)]
                    stream = _lib.current_stream_ptr(dev)
                    rcs = []
                    for rep in range(2):                                   # twice: buffer reuse across calls on the same pair of streams
                        rcs.append(lib.tpg_halo_exchange_y_pipelined_peers(comm.handle, south, north, _lib.ptr_table(devs), nf,
                                                                           *[b.data_ptr() for b in bufs], Nx, Ny, Nz, Hx, Hy, Hz, ft, stream,
                                                                           C.c_void_p(cs.cuda_stream) if cs is not None else None, fps))
                    torch.cuda.synchronize()
                    good = all(r_ == 0 for r_ in rcs)
                    for h, d in zip(hosts, devs):
                        want = h.copy()
                        if south >= 0 and north >= 0:
                            want[:, :Hy] = h[:, Ny:Ny + Hy]; want[:, Ny + Hy:] = h[:, Hy:2 * Hy]
                        elif south >= 0:
                            want[:, :Hy] = h[:, Hy:2 * Hy]                 # the one send (south) pairs the one receive (from the south)
                        else:
                            want[:, Ny + Hy:] = h[:, Ny:Ny + Hy]
                        good = good and np.array_equal(d.cpu().numpy(), want)
                    pcases.append({"size": [Nx, Ny, Nz], "nfields": nf, "fields_per_stage": fps, "two_streams": cs is not None,
                                   "peers": [south, north], "bit_exact": bool(good)})
                    out["ok"] = out["ok"] and good
    out["pipelined"] = {"cases": len(pcases), "all_bit_exact": all(c["bit_exact"] for c in pcases),
                        "failed": [c for c in pcases if not c["bit_exact"]]}
    # cost of the two forms at config 4's band, interior rank (two seams), back to back on the loop-back (no link: RCCL's kernels copy on the device)
    (Nx, Ny, Nz), (Hx, Hy, Hz), nf = (3600, 225, 75), (4, 4, 4), 4
    shape = (Nz + 2 * Hz, Ny + 2 * Hy, Nx + 2 * Hx)
    devs = [torch.rand(shape, dtype=torch.float64, device=dev) for _ in range(nf)]
    nbuf = lib.tpg_y_halo_buffer_elems(nf, Nx, Nz, Hx, Hy, Hz)
    bufs = [torch.empty(nbuf, dtype=torch.float64, device=dev) for _ in range(4)]
    bp, ptrs, stream = [b.data_ptr() for b in bufs], _lib.ptr_table(devs), _lib.current_stream_ptr(dev)
    csp = C.c_void_p(comm_stream.cuda_stream)
    forms = {"monolithic": lambda: lib.tpg_halo_exchange_y_peers(comm.handle, 0, 0, ptrs, nf, *bp, Nx, Ny, Nz, Hx, Hy, Hz, 1, stream),
             "pipelined_1": lambda: lib.tpg_halo_exchange_y_pipelined_peers(comm.handle, 0, 0, ptrs, nf, *bp, Nx, Ny, Nz, Hx, Hy, Hz, 1, stream, csp, 1),
             "pipelined_2": lambda: lib.tpg_halo_exchange_y_pipelined_peers(comm.handle, 0, 0, ptrs, nf, *bp, Nx, Ny, Nz, Hx, Hy, Hz, 1, stream, csp, 2)}
    cost = {}
    for name, fn in forms.items():
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        host_ms = (time.perf_counter() - t0) / 20 * 1e3
        torch.cuda.synchronize()
        cost[name] = {"device_ms": round(e0.elapsed_time(e1) / 20, 4), "host_enqueue_ms": round(host_ms, 4)}
    out["exchange_cost_config4_band_loopback"] = cost

    # ---- tpg_fill_halo_regions_distributed_peers: the whole fill of a band in ONE call (zipper / periodic x / seams) ----------
    dcases = []
    for (Nx, Ny, Nz), (Hx, Hy, Hz) in (((48, 40, 3), (4, 4, 2)), ((48, 40, 3), (5, 5, 5))):        # the second: the distributed example's halo
        shape = (Nz + 2 * Hz, Ny + 2 * Hy, Nx + 2 * Hx)
        specs = [(0, 0, 1), (1, 0, -1), (0, 1, -1), (1, 1, 1)]
        xl = (C.c_int8 * 4)(*[s_[0] for s_ in specs]); yl = (C.c_int8 * 4)(*[s_[1] for s_ in specs]); sg = (C.c_int32 * 4)(*[s_[2] for s_ in specs])
        nbuf = lib.tpg_y_halo_buffer_elems(4, Nx, Nz, Hx, Hy, Hz)
        bufs = [torch.empty(nbuf, dtype=torch.float64, device=dev) for _ in range(4)]
        stream = _lib.current_stream_ptr(dev)
        # (a) a middle band: no zipper, both seams (peers = this rank): periodic x, then south halo <- own north interior rows etc.
        # (b) the north band: zipper + periodic x, then the south seam only (south halo <- own south interior rows: the one send pairs the one recv)
        for label, south, north, zipper, pipelined in (("middle", 0, 0, 0, False), ("north", 0, -1, 1, False),
                                                       ("middle", 0, 0, 0, True), ("north", 0, -1, 1, True)):
            hosts = [rng.uniform(-1, 1, shape) for _ in specs]
            devs = [torch.from_numpy(h).to(dev) for h in hosts]
            # expected: the product's own LOCAL fill (bit-exact against the oracle in tests/test_gpu_zipper.py) + the loop-back row moves
            refs = [d.clone() for d in devs]
            _lib.check(lib.tpg_fill_halo_regions(_lib.ptr_table(refs), 4, xl, yl, sg, Nx, Ny, Nz, Hx, Hy, Hz, zipper, 1, stream))
            bargs = (bufs[0].data_ptr(), bufs[1].data_ptr() if north >= 0 else None, bufs[2].data_ptr(), bufs[3].data_ptr() if north >= 0 else None)
            if pipelined:
                rc = lib.tpg_fill_halo_regions_distributed_pipelined_peers(comm.handle, south, north, zipper, _lib.ptr_table(devs), 4, xl, yl, sg,
                                                                           *bargs, Nx, Ny, Nz, Hx, Hy, Hz, 1, stream,
                                                                           C.c_void_p(comm_stream.cuda_stream), 1)
            else:
                rc = lib.tpg_fill_halo_regions_distributed_peers(comm.handle, south, north, zipper, _lib.ptr_table(devs), 4, xl, yl, sg,
                                                                 *bargs, Nx, Ny, Nz, Hx, Hy, Hz, 1, stream)
            torch.cuda.synchronize()
            good = rc == 0
            for r, d in zip(refs, devs):
                want = r.clone()
                if north >= 0:
                    want[:, :Hy] = r[:, Ny:Ny + Hy]; want[:, Ny + Hy:] = r[:, Hy:2 * Hy]
                else:
                    want[:, :Hy] = r[:, Hy:2 * Hy]
                good = good and bool(torch.equal(d, want))
            dcases.append({"band": label, "halo": [Hx, Hy, Hz], "pipelined": pipelined, "rc": rc, "bit_exact": bool(good)})
            out["ok"] = out["ok"] and good
    out["distributed_fill"] = dcases

    # ---- two exchanges of EQUAL geometry in flight on two streams, each with its own message buffers (seam-buffer ownership) ----
    (Nx, Ny, Nz), (Hx, Hy, Hz) = (3600, 64, 8), (4, 4, 4)
    shape = (Nz + 2 * Hz, Ny + 2 * Hy, Nx + 2 * Hx)
    nbuf = lib.tpg_y_halo_buffer_elems(2, Nx, Nz, Hx, Hy, Hz)
    sets = []
    for q in range(2):
        hosts = [rng.uniform(-1, 1, shape) for _ in range(2)]
        sets.append((hosts, [torch.from_numpy(h).to(dev) for h in hosts], [torch.empty(nbuf, dtype=torch.float64, device=dev) for _ in range(4)],
                     torch.cuda.Stream(dev)))
    torch.cuda.synchronize()
    rcs = []
    for rep in range(3):
        for hosts, devs, bb, st in sets:
            rcs.append(lib.tpg_halo_exchange_y_peers(comm.handle, 0, 0, _lib.ptr_table(devs), 2, *[b.data_ptr() for b in bb],
                                                     Nx, Ny, Nz, Hx, Hy, Hz, 1, C.c_void_p(st.cuda_stream)))
    torch.cuda.synchronize()
    good = all(r == 0 for r in rcs)
    for hosts, devs, bb, st in sets:
        for h, d in zip(hosts, devs):
            want = h.copy()
            want[:, :Hy] = h[:, Ny:Ny + Hy]; want[:, Ny + Hy:] = h[:, Hy:2 * Hy]
            good = good and np.array_equal(d.cpu().numpy(), want)
    out["two_streams_own_buffers_bit_exact"] = bool(good)
    out["ok"] = out["ok"] and good

    # ---- the capture fence: a capturing stream is refused (TPG_ERR_UNSUPPORTED), the capture itself stays valid -----------------
    d2 = torch.rand((4, 20, 24), dtype=torch.float64, device=dev)
    before = d2.clone()
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream(dev)
    side.wait_stream(torch.cuda.current_stream())
    bb = [torch.empty(lib.tpg_y_halo_buffer_elems(1, 16, 2, 4, 4, 1), dtype=torch.float64, device=dev) for _ in range(4)]
    with torch.cuda.stream(side), torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
        sp = C.c_void_p(side.cuda_stream)
        rc_fence = lib.tpg_halo_exchange_y_peers(comm.handle, 0, 0, _lib.ptr_table([d2]), 1, *[b.data_ptr() for b in bb], 16, 12, 2, 4, 4, 1, 1, sp)
        msg = lib.tpg_last_error().decode()
        rc_fence_p = lib.tpg_halo_exchange_y_pipelined_peers(comm.handle, 0, 0, _lib.ptr_table([d2]), 1, *[b.data_ptr() for b in bb], 16, 12, 2, 4, 4, 1, 1,
                                                             sp, C.c_void_p(comm_stream.cuda_stream), 1)
        rc_per = lib.tpg_periodic_x_fill(_lib.ptr_table([d2]), 1, 16, 12, 2, 4, 4, 1, 1, sp)        # a capturable call after the refusal
    torch.cuda.current_stream().wait_stream(side)
    g.replay()
    torch.cuda.synchronize()
    want = before.clone()
    want[:, :, :4] = before[:, :, 16:20]; want[:, :, 20:] = before[:, :, 4:8]
    out["capture_fence"] = {"rc": rc_fence, "rc_pipelined": rc_fence_p, "message": msg, "periodic_rc_in_capture": rc_per, "replay_bit_exact": bool(torch.equal(d2, want))}
    out["ok"] = out["ok"] and rc_fence == -5 and rc_fence_p == -5 and rc_per == 0 and out["capture_fence"]["replay_bit_exact"]

    # ---- a LATE failure of the pipelined exchange (test library, TPG_EXCHANGE_FAIL_STAGE=1: injected right after the RCCL group of stage 1
    #      went onto comm_stream): the call returns TPG_ERR_RCCL, and its post-condition must hold all the same -- `stream` is ordered after
    #      everything on comm_stream (synchronising `stream` alone leaves comm_stream idle), so the SAME buffers and streams serve the
    #      next, successful call.  Config 4's band, 4 stages of one field (a stage's transfers take tens of microseconds). ----------------
    from tools import testlib
    tl = testlib.lib()
    (Nx, Ny, Nz), (Hx, Hy, Hz), nf = (3600, 225, 75), (4, 4, 4), 4
    shape = (Nz + 2 * Hz, Ny + 2 * Hy, Nx + 2 * Hx)
    nbuf = lib.tpg_y_halo_buffer_elems(nf, Nx, Nz, Hx, Hy, Hz)
    bufs = [torch.empty(nbuf, dtype=torch.float64, device=dev) for _ in range(4)]
    bp, csp = [b.data_ptr() for b in bufs], C.c_void_p(comm_stream.cuda_stream)
    main_stream = torch.cuda.current_stream()
    stream = _lib.current_stream_ptr(dev)
    late = {"rcs": [], "comm_stream_idle_after_stream_sync": [], "messages": set()}
    os.environ["TPG_EXCHANGE_FAIL_STAGE"] = "1"
    testlib.check(tl.tpg_reload_config())
    devs = [torch.rand(shape, dtype=torch.float64, device=dev) for _ in range(nf)]
    torch.cuda.synchronize()
    for trial in range(5):
        rc = tl.tpg_halo_exchange_y_pipelined_peers(comm.handle, 0, 0, _lib.ptr_table(devs), nf, *bp, Nx, Ny, Nz, Hx, Hy, Hz, 1, stream, csp, 1)
        late["messages"].add(tl.tpg_last_error().decode())
        main_stream.synchronize()                                      # `stream` only
        late["comm_stream_idle_after_stream_sync"].append(bool(comm_stream.query()))
        late["rcs"].append(rc)
        torch.cuda.synchronize()
    del os.environ["TPG_EXCHANGE_FAIL_STAGE"]
    testlib.check(tl.tpg_reload_config())
    before = [d.clone() for d in devs]
    rc_after = tl.tpg_halo_exchange_y_pipelined_peers(comm.handle, 0, 0, _lib.ptr_table(devs), nf, *bp, Nx, Ny, Nz, Hx, Hy, Hz, 1, stream, csp, 1)
    torch.cuda.synchronize()
    good = rc_after == 0
    for b, d in zip(before, devs):
        want = b.clone()
        want[:, :Hy] = b[:, Ny:Ny + Hy]; want[:, Ny + Hy:] = b[:, Hy:2 * Hy]
        good = good and bool(torch.equal(d, want))
    late.update(rc_after=rc_after, reuse_bit_exact=bool(good), messages=sorted(late["messages"]))
    out["late_failure"] = late
    out["ok"] = out["ok"] and good and late["rcs"] == [-7] * 5 and all(late["comm_stream_idle_after_stream_sync"])

    # ---- the ordering-event pool belongs to its host thread: a short-lived thread runs a pipelined exchange and ends (its events are
    #      destroyed by the thread_local owner), the main thread's pool keeps working ---------------------------------------------------
    import threading
    th_rc = []

    def in_thread():
        torch.cuda.set_device(0)
        cs2 = torch.cuda.Stream(dev)
        with torch.cuda.stream(torch.cuda.Stream(dev)):
            th_rc.append(lib.tpg_halo_exchange_y_pipelined_peers(comm.handle, 0, 0, _lib.ptr_table(devs), nf, *bp, Nx, Ny, Nz, Hx, Hy, Hz, 1,
                                                                 _lib.current_stream_ptr(dev), C.c_void_p(cs2.cuda_stream), 2))
            torch.cuda.synchronize()

    for _ in range(3):
        t = threading.Thread(target=in_thread); t.start(); t.join()
    rc_main = lib.tpg_halo_exchange_y_pipelined_peers(comm.handle, 0, 0, _lib.ptr_table(devs), nf, *bp, Nx, Ny, Nz, Hx, Hy, Hz, 1, stream, csp, 2)
    torch.cuda.synchronize()
    out["event_pool_threads"] = {"thread_rcs": th_rc, "main_rc_after": rc_main}
    out["ok"] = out["ok"] and th_rc == [0, 0, 0] and rc_main == 0

    # the chain rule: a one-rank chain has no seam
    d = torch.zeros((1, 12, 12), dtype=torch.float64, device=dev)
    out["single_rank_chain_rc"] = lib.tpg_halo_exchange_y(comm.handle, 0, 1, _lib.ptr_table([d]), 1, None, None, None, None, 4, 4, 1, 4, 4, 0, 1, None)
    out["ok"] = out["ok"] and out["single_rank_chain_rc"] == 0
    comm.destroy()
    dist.destroy_process_group()
    print(json.dumps(out))
    return 0 if out["ok"] else 1


if __name__ == "__main__":
    sys.exit(main())
