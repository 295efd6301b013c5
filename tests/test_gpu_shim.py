"""The C ABI's seam exchange between SEVERAL REAL PROCESSES on one GPU.

RCCL refuses two ranks on one device, so on a one-GPU box the production exchange -- tpg_halo_exchange_y (packed and pack-free),
tpg_halo_exchange_y_pipelined, tpg_fill_halo_regions_distributed(_pipelined) with a communicator of more than one rank -- could only run on a
one-rank communicator whose peers are the rank itself (tests/test_gpu_exchange.py).  Here the TEST library binds a test double of the ten
librccl entry points (tools/nccl_shim: shared-memory mailboxes between the processes, TPG_RCCL_LIBRARY) instead of librccl, and three
processes with DIFFERENT data exchange through exactly those entry points: which buffer goes to which peer, the order inside a group, the
pairing of the pipelined stage groups across processes, many messages per group (pack-free).  This validates the library's use of the API,
not RCCL and not the link; the product library never loads the double (it binds librccl by fixed names and reads no environment variable)."""
import ctypes as C
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SPECS = [(0, 0, 1), (1, 0, -1), (0, 1, -1), (1, 1, 1), (0, 0, 1)]          # xloc, yloc, sign


def _field(seed, shape, dtype):
    return np.random.default_rng(seed).uniform(-1, 1, shape).astype(dtype)


def _worker(rank, world, port, out):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), TPG_RCCL_LIBRARY=os.path.join(ROOT, "tools", "nccl_shim", "libnccl_shim.so"),
                          TPG_SHIM_DEADLINE_S="30")
        sys.path.insert(0, ROOT)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import orthogonalsphericalshellgrids.jl_amd as osg
        from orthogonalsphericalshellgrids.jl_amd import _lib
        from tools import testlib
        from oracle import oracle
        torch.cuda.set_device(0)
        dev = torch.device("cuda", 0)
        _lib._lib = testlib.lib()                                      # the package's calls -> the test library -> the test double
        lib = _lib.lib()
        comm = osg.RcclComm.from_torch()                               # readiness agreement + id broadcast over gloo, world 3
        assert (comm.rank, comm.nranks) == (rank, world)
        stream = _lib.current_stream_ptr(dev)
        cs = torch.cuda.Stream(dev)
        csp = C.c_void_p(cs.cuda_stream)
        cases = 0
        for ci, ((Nx, Ny, Nz), (Hx, Hy, Hz), nf, dt, tdt) in enumerate((((48, 20, 3), (4, 4, 2), 5, np.float64, torch.float64),
                                                                         ((20, 12, 2), (3, 2, 1), 3, np.float32, torch.float32),
                                                                         ((3600, 24, 6), (4, 4, 4), 4, np.float64, torch.float64))):
            shape = (Nz + 2 * Hz, Ny + 2 * Hy, Nx + 2 * Hx)
            ft = 1 if dt == np.float64 else 0
            everyone = [[_field(1000 * ci + 10 * r + f, shape, dt) for f in range(nf)] for r in range(world)]     # every rank knows every rank's data
            nbuf = lib.tpg_y_halo_buffer_elems(nf, Nx, Nz, Hx, Hy, Hz)
            bufs = [torch.empty(nbuf, dtype=tdt, device=dev) for _ in range(4)]
            bp = [b.data_ptr() for b in bufs]
            xl = (C.c_int8 * nf)(*[SPECS[f][0] for f in range(nf)]); yl = (C.c_int8 * nf)(*[SPECS[f][1] for f in range(nf)])
            sg = (C.c_int32 * nf)(*[SPECS[f][2] for f in range(nf)])

            def expect(local_fill):
                """this rank's fields after the exchange: its halo rows = the neighbours' interior rows (after THEIR local fill, if any)"""
                mine = [a.copy() for a in everyone[rank]]
                others = {r: [a.copy() for a in everyone[r]] for r in (rank - 1, rank + 1) if 0 <= r < world}
                if local_fill:
                    for r, fs in list(others.items()) + [(rank, mine)]:
                        for f, a in enumerate(fs):
                            if r == world - 1:
                                oracle.zipper_fill(a, SPECS[f][0], SPECS[f][1], SPECS[f][2], (Nx, Ny, Nz), (Hx, Hy, Hz))
                            oracle.periodic_x_fill(a, (Nx, Ny, Nz), (Hx, Hy, Hz))
                for f, a in enumerate(mine):
                    if rank > 0:
                        a[:, :Hy] = others[rank - 1][f][:, Ny:Ny + Hy]           # south halo <- the south neighbour's northernmost interior rows
                    if rank < world - 1:
                        a[:, Ny + Hy:] = others[rank + 1][f][:, Hy:2 * Hy]       # north halo <- the north neighbour's southernmost interior rows
                return mine

            forms = [("packed", lambda p: lib.tpg_halo_exchange_y(comm.handle, rank, world, p, nf, *bp, Nx, Ny, Nz, Hx, Hy, Hz, ft, stream), False),
                     ("pack-free", lambda p: lib.tpg_halo_exchange_y(comm.handle, rank, world, p, nf, None, None, None, None, Nx, Ny, Nz, Hx, Hy, Hz, ft, stream), False),
                     ("pipelined_1", lambda p: lib.tpg_halo_exchange_y_pipelined(comm.handle, rank, world, p, nf, *bp, Nx, Ny, Nz, Hx, Hy, Hz, ft, stream, csp, 1), False),
                     ("pipelined_2", lambda p: lib.tpg_halo_exchange_y_pipelined(comm.handle, rank, world, p, nf, *bp, Nx, Ny, Nz, Hx, Hy, Hz, ft, stream, csp, 2), False),
                     ("pipelined_one_stream", lambda p: lib.tpg_halo_exchange_y_pipelined(comm.handle, rank, world, p, nf, *bp, Nx, Ny, Nz, Hx, Hy, Hz, ft, stream, None, 3), False),
                     ("distributed_fill", lambda p: lib.tpg_fill_halo_regions_distributed(comm.handle, rank, world, p, nf, xl, yl, sg, *bp, Nx, Ny, Nz, Hx, Hy, Hz, ft, stream), True),
                     ("distributed_fill_pipelined", lambda p: lib.tpg_fill_halo_regions_distributed_pipelined(comm.handle, rank, world, p, nf, xl, yl, sg, *bp, Nx, Ny, Nz, Hx, Hy, Hz,
                                                                                                         ft, stream, csp, 2), True)]
            for name, call, local_fill in forms:
                if name == "pack-free" and Nx >= 3600:
                    continue                                           # hundreds of 115 KB messages through a host-staged double: covered by the small cases
                want = expect(local_fill)
                for rep in range(2):                                   # twice on fresh data: mailbox and buffer reuse, message order across calls
                    # (fresh data, not a second call on the result: the zipper is not idempotent -- the x-Face fold maps i = Nx/2 + 1 of row Ny
                    # onto itself with the field's sign)
                    devs = [torch.from_numpy(a).to(dev) for a in everyone[rank]]
                    rc = call(_lib.ptr_table(devs))
                    assert rc == 0, (name, rc, lib.tpg_last_error())
                    torch.cuda.synchronize()
                    for f, (d, w) in enumerate(zip(devs, want)):
                        got = d.cpu().numpy()
                        assert np.array_equal(got, w), f"rank {rank} case {ci} {name} rep {rep} field {f}: {int((got != w).sum())} cells differ"
                cases += 1
            dist.barrier()
        # ---- the Python production path: HaloFillPlan on a DistributedTripolarGrid whose architecture carries the communicator = ONE C call per
        # fill (tpg_fill_halo_regions_distributed[_pipelined]); its plan-build collective (all ranks must agree on the stage layout) runs for
        # real over the three processes.  Expected: every rank's slab == its rows of the serially filled GLOBAL field (oracle).
        gsize, ghalo = (48, 36, 2), (4, 4, 1)
        (Nx, Ny, Nz), (Hx, Hy, Hz) = gsize, ghalo
        specs4 = SPECS[:4]
        for stage in (0, 1, 3):
            rng = np.random.default_rng(4242)                           # the same global data on every rank
            globs = []
            for xl_, yl_, sg_ in specs4:
                g = rng.uniform(-1, 1, (Nz + 2 * Hz, Ny + 2 * Hy, Nx + 2 * Hx))
                g[:, :Hy] = 12345.0; g[:, Hy + Ny:] = 12345.0
                globs.append(g)
            arch = osg.Distributed(osg.GPU(0), osg.Partition(y=world), rccl_comm=comm)      # rank / world from torch.distributed (gloo)
            assert arch.local_rank == rank
            grid = osg.TripolarGrid(arch, torch.float64, size=gsize, halo=ghalo)
            j0, j1 = grid.jrange
            fs = []
            for (xl_, yl_, sg_), g in zip(specs4, globs):
                f = osg.Field((osg.Face if xl_ else osg.Center, osg.Face if yl_ else osg.Center, osg.Center), grid)
                slab = g[:, j0 - 1:j1 + 2 * Hy].copy()
                slab[:, :Hy] = 12345.0; slab[:, Hy + (j1 - j0 + 1):] = 12345.0
                f.data.copy_(torch.from_numpy(slab))
                fs.append(f)
            plan = osg.halo_fill_plan(fs, fields_per_stage=stage)
            assert plan.is_distributed
            plan()
            torch.cuda.synchronize()
            for (xl_, yl_, sg_), g in zip(specs4, globs):
                oracle.fill_halo_regions(g, xl_, yl_, sg_, gsize, ghalo)
            for f, g in zip(fs, globs):
                assert np.array_equal(f.data.cpu().numpy(), g[:, j0 - 1:j1 + 2 * Hy]), f"rank {rank} HaloFillPlan stage {stage} {f.loc}"
            cases += 1
        # ranks that disagree on fields_per_stage are caught at plan build, on every rank, before any exchange
        try:
            osg.halo_fill_plan(fs, fields_per_stage=2 if rank == 1 else 1)
            raise AssertionError("ranks disagreed on fields_per_stage and the plan was built")
        except ValueError as e:
            assert "disagree" in str(e)
        dist.barrier()
        # a size mismatch between a send and its receive is an error of the double, as it would be on the wire: rank 0 sends 3 fields, rank 1
        # expects 2 -> rank 1's receive fails (TPG_ERR_RCCL); rank 0's send completes (the message was put); nobody hangs
        if world >= 2 and rank <= 1:
            (Nx, Ny, Nz), (Hx, Hy, Hz) = (20, 12, 1), (2, 2, 0)
            shape = (Nz, Ny + 2 * Hy, Nx + 2 * Hx)
            nf = 3 if rank == 0 else 2
            devs = [torch.zeros(shape, dtype=torch.float64, device=dev) for _ in range(nf)]
            n = lib.tpg_y_halo_buffer_elems(nf, Nx, Nz, Hx, Hy, Hz)
            bb = [torch.zeros(n, dtype=torch.float64, device=dev) for _ in range(4)]
            south, north = (-1, 1) if rank == 0 else (0, -1)
            os.environ["TPG_SHIM_DEADLINE_S"] = "5"
            rc = lib.tpg_halo_exchange_y_peers(comm.handle, south, north, _lib.ptr_table(devs), nf, *[b.data_ptr() for b in bb], Nx, Ny, Nz, Hx, Hy, Hz, 1, stream)
            torch.cuda.synchronize()
            out.put((rank, "mismatch", rc, lib.tpg_last_error().decode()))
        dist.barrier()
        out.put((rank, "ok", cases, ""))
        os._exit(0)                                                    # the mismatch case left a message behind on purpose: no orderly teardown
    except Exception as e:                                             # noqa: BLE001
        import traceback
        out.put((rank, "error", -1, f"{type(e).__name__}: {e}\n{traceback.format_exc()[-1500:]}"))
        os._exit(1)


def test_exchange_entry_points_between_three_processes_over_the_test_double(gpu):
    world = 3
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    ctx = mp.get_context("spawn")
    out = ctx.SimpleQueue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
    for p in procs:
        if p.is_alive():
            p.kill()
    got = []
    while not out.empty():
        got.append(out.get())
    errors = [g for g in got if g[1] == "error"]
    assert not errors, errors
    oks = {g[0]: g[2] for g in got if g[1] == "ok"}
    assert oks == {0: 23, 1: 23, 2: 23}, got                          # 7 forms x 3 geometries minus the large pack-free case, + 3 HaloFillPlan runs
    mism = {g[0]: g for g in got if g[1] == "mismatch"}
    assert mism[1][2] == -7 and "nccl_shim" in mism[1][3] and mism[0][2] in (0, -7)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
