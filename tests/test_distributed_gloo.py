"""N>1 host logic on CPU: world_size 2 and 3 over gloo (127.0.0.1).

What runs here is the product's HOST protocol of a distributed halo fill -- partition rule
(local_row_range), exchange_plan, message layout and the batched point-to-point transport
(torch_distributed_transport, the same code that rides RCCL on the GPU box) -- with the device
kernels (zipper / periodic / pack / unpack) replaced, in this test only, by the oracle and numpy
slicing.  Expected result: every rank's padded slab equals the matching rows of the GLOBAL field
after a serial fill_halo_regions! (zipper + periodic)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

SIZE, HALO = (12, 17, 2), (3, 2, 1)
SENT = 12345.0
FIELDS = [(0, 0, 1), (1, 0, -1), (0, 1, -1), (1, 1, 1)]      # (xloc, yloc, sign): c, u, v, zeta


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, errors):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import sys
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import orthogonalsphericalshellgrids.jl_amd as osg
        from orthogonalsphericalshellgrids.jl_amd.distributed import SOUTH, NORTH
        from oracle import oracle

        (Nx, Ny, Nz), (Hx, Hy, Hz) = SIZE, HALO
        arch = osg.Distributed(osg.GPU(), osg.Partition(y=world))       # rank / world from torch.distributed
        assert arch.local_rank == rank and arch.ranks == (1, world, 1)
        jstart, jend = osg.local_row_range(Ny, arch)
        ny = jend - jstart + 1
        plan = osg.exchange_plan(rank, world)

        rng = np.random.default_rng(99)                                   # same global data on every rank
        locals_, globals_ = [], []
        for xl, yl, sg in FIELDS:
            glob = rng.uniform(-1, 1, (Nz + 2 * Hz, Ny + 2 * Hy, Nx + 2 * Hx))
            glob[:, :Hy] = SENT; glob[:, Hy + Ny:] = SENT
            loc = glob[:, jstart - 1:jstart - 1 + ny + 2 * Hy].copy()     # global rows jstart-Hy..jend+Hy
            loc[:, :Hy] = SENT; loc[:, Hy + ny:] = SENT                   # halo rows unknown before the fill
            oracle.fill_halo_regions(glob, xl, yl, sg, SIZE, HALO)
            locals_.append(loc); globals_.append(glob)

        lsize = (Nx, ny, Nz)
        for (xl, yl, sg), loc in zip(FIELDS, locals_):
            if rank == world - 1:                                         # zipper on the north rank only
                oracle.zipper_fill(loc, xl, yl, sg, lsize, HALO)
            oracle.periodic_x_fill(loc, lsize, HALO)
        # pack: [field][level][Hy][sx], interior rows next to each side
        rows = {SOUTH: slice(Hy, 2 * Hy), NORTH: slice(ny, ny + Hy)}
        halo_rows = {SOUTH: slice(0, Hy), NORTH: slice(Hy + ny, ny + 2 * Hy)}
        send = {m.side: torch.from_numpy(np.stack([l[:, rows[m.side]] for l in locals_])) for m in plan}
        recv = {m.side: torch.empty_like(send[m.side]) for m in plan}
        osg.torch_distributed_transport(plan, send, recv, None)
        for m in plan:
            for f, l in enumerate(locals_):
                l[:, halo_rows[m.side]] = recv[m.side][f].numpy()

        for loc, glob in zip(locals_, globals_):
            want = glob[:, jstart - 1:jstart - 1 + ny + 2 * Hy]
            if not np.array_equal(loc, want):
                bad = np.argwhere(loc != want)
                raise AssertionError(f"rank {rank}: {len(bad)} cells differ, first {bad[:3].tolist()}")
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:                                                # noqa: BLE001
        errors.put(f"rank {rank}: {type(e).__name__}: {e}")
        raise


@pytest.mark.parametrize("world", [2, 3])
def test_latitude_band_halo_exchange_over_gloo(world):
    ctx = mp.get_context("spawn")
    errors = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, errors)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
    msgs = []
    while not errors.empty():
        msgs.append(errors.get())
    assert not msgs, msgs
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]


def _readiness_worker(rank, world, port, results):
    """RcclComm.from_torch with librccl 'unavailable' on rank 1: every rank must leave with the same RuntimeError BEFORE anyone
    enters the collective ncclCommInitRank (round 2: the ranks that could bind librccl blocked inside it)"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import orthogonalsphericalshellgrids.jl_amd as osg
    osg.RcclComm.available = staticmethod(lambda: rank != 1)
    entered = []
    osg.RcclComm.create = classmethod(lambda cls, *a: entered.append(a) or (_ for _ in ()).throw(AssertionError("ncclCommInitRank entered")))
    try:
        osg.RcclComm.from_torch()
        results.put((rank, "no error"))
    except RuntimeError as e:
        results.put((rank, str(e)))
    assert not entered
    dist.barrier()
    dist.destroy_process_group()


def test_rccl_readiness_is_agreed_before_the_collective_init():
    world = 3
    ctx = mp.get_context("spawn")
    results = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_readiness_worker, args=(r, world, port, results)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    got = {}
    while not results.empty():
        r, msg = results.get()
        got[r] = msg
    assert set(got) == {0, 1, 2} and all("librccl unavailable on rank(s) [1]" in m for m in got.values()), got


def _stage_agreement_worker(rank, world, port, results):
    """HaloFillPlan's one collective at plan build: ranks that disagree on the stage layout of the pipelined seam exchange (group(k) of
    a rank pairs with group(k) of its neighbour, include/tripolar_hip.h) must ALL leave with a ValueError -- before any RCCL group"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import orthogonalsphericalshellgrids.jl_amd as osg
    from orthogonalsphericalshellgrids.jl_amd.fields import _agree_across_ranks
    arch = osg.Distributed(osg.GPU(), osg.Partition(y=world))
    _agree_across_ranks(arch, (4, 3600, (75, 4, 4, 4), False, 2), "stage layout")            # equal everywhere: passes
    try:
        _agree_across_ranks(arch, (4, 3600, (75, 4, 4, 4), False, 2 if rank != 1 else 1), "stage layout")
        results.put((rank, "no error"))
    except ValueError as e:
        results.put((rank, str(e)))
    longer = osg.Distributed(osg.GPU(), osg.Partition(y=8), local_rank=3)                    # not this group's chain: no collective, no error
    _agree_across_ranks(longer, rank, "anything")
    dist.barrier()
    dist.destroy_process_group()


def test_ranks_must_agree_on_the_stage_layout_of_the_pipelined_exchange():
    world = 3
    ctx = mp.get_context("spawn")
    results = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_stage_agreement_worker, args=(r, world, port, results)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    got = {}
    while not results.empty():
        r, msg = results.get()
        got[r] = msg
    assert set(got) == {0, 1, 2} and all("disagree on stage layout" in m for m in got.values()), got
