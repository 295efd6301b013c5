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
                                                   ((3600, 64, 75), (4, 4, 4), 4, np.float64, torch.float64)):
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
