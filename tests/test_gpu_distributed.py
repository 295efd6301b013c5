"""Distributed latitude-band halo fill on ONE GPU: R ranks emulated in one process with a
loop-back transport, exercising the real device kernels (zipper on the north rank, periodic x,
tpg_pack_y_halo / tpg_unpack_y_halo) and the host protocol.  Expected: every rank's padded slab ==
rows jstart-Hy..jend+Hy of the serially filled global field (oracle)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
SENT = 12345.0
SPECS = [(0, 0, 1), (1, 0, -1), (0, 1, -1), (1, 1, 1)]


@pytest.mark.parametrize("stage", [0, 1, 3])      # 0: monolithic exchange; k: the pipelined form's stages of k fields (4 fields: 4 / 2 stages)
@pytest.mark.parametrize("R", [2, 4])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("halo", [(4, 4, 2), (5, 5, 5)], ids=["halo4", "halo5"])      # (5, 5, 5): examples/distributed_bickley_jet.jl:23
def test_band_halo_fill_with_loopback_transport(osg, oracle, gpu, R, dtype, stage, halo):
    size = (48, 40, 3)
    (Nx, Ny, Nz), (Hx, Hy, Hz) = size, halo
    tdt = torch.float64 if dtype == np.float64 else torch.float32
    rng = np.random.default_rng(17)
    globs = []
    for xl, yl, sg in SPECS:
        g = rng.uniform(-1, 1, (Nz + 2 * Hz, Ny + 2 * Hy, Nx + 2 * Hx)).astype(dtype)
        g[:, :Hy] = SENT; g[:, Hy + Ny:] = SENT
        globs.append(g)
    ranks = []
    for r in range(R):
        arch = osg.Distributed(osg.GPU(0), osg.Partition(y=R), local_rank=r)
        grid = osg.TripolarGrid(arch, tdt, size=size, halo=halo)
        jstart, jend = grid.jrange
        fs = []
        for (xl, yl, sg), g in zip(SPECS, globs):
            loc = (osg.Face if xl else osg.Center, osg.Face if yl else osg.Center, osg.Center)
            f = osg.Field(loc, grid)
            north = f.boundary_conditions.north
            assert osg.is_zipper(north) == (r == R - 1)                     # zipper only on the last rank
            if r == R - 1:
                assert north.condition == sg                                # default sign by location
            slab = g[:, jstart - 1:jend + 2 * Hy].copy()
            slab[:, :Hy] = SENT; slab[:, Hy + (jend - jstart + 1):] = SENT
            f.data.copy_(torch.from_numpy(slab))
            fs.append(f)
        ranks.append((arch, grid, fs))

    # phase 1 on every rank: zipper (north rank) + periodic x + pack + post; phase 2: delivery + the product's own unpack
    mailbox = osg.LoopbackMailbox()
    plans = [osg.halo_fill_plan(fs, exchange=mailbox.endpoint(r), fields_per_stage=stage) for r, (arch, grid, fs) in enumerate(ranks)]
    for plan in plans:
        plan.begin()
    for plan in plans:
        plan.finish()
    torch.cuda.synchronize()
    assert all(not q for q in mailbox.box.values())                                # every posted message was delivered

    for (xl, yl, sg), g in zip(SPECS, globs):
        oracle.fill_halo_regions(g, xl, yl, sg, size, halo)
    for r, (arch, grid, fs) in enumerate(ranks):
        jstart, jend = grid.jrange
        for f, g in zip(fs, globs):
            assert np.array_equal(f.data.cpu().numpy(), g[:, jstart - 1:jend + 2 * Hy]), (r, f.loc)


@pytest.mark.parametrize("tdt,size,halo,offset", [
    (torch.float64, (20, 12, 2), (4, 3, 1), 0),          # 16-B chunks
    (torch.float64, (20, 12, 2), (5, 5, 5), 0),          # the reference's model halo: Float64 rows stay 16-B rows
    (torch.float64, (20, 12, 2), (5, 3, 1), 1),          # field bases 8 B off the 16-B grid: element-aligned 16-B chunks of the slabs (k_pack_loose)
    (torch.float32, (20, 12, 2), (4, 3, 1), 0),          # sx = 28: 16-B chunks
    (torch.float32, (20, 12, 2), (5, 5, 5), 0),          # sx = 30 = 2 mod 4 (as Nx = 3600 at halo 5): k_pack_loose, a short last chunk per slab
    (torch.float32, (20, 12, 2), (5, 3, 1), 1),          # bases 4 B off: k_pack_loose
])
def test_pack_unpack_roundtrip_and_layout(osg, gpu, tdt, size, halo, offset):
    """message layout [field][level][Hy][sx]; pack reads interior rows, unpack writes halo rows -- on the 16-B grid (k_pack) and off it (k_pack_loose)"""
    lib = osg._lib.lib()
    (Nx, Ny, Nz), (Hx, Hy, Hz) = size, halo
    shape = (Nz + 2 * Hz, Ny + 2 * Hy, Nx + 2 * Hx)
    ft = osg._lib.ft_of(tdt)
    count = shape[0] * shape[1] * shape[2]
    fs = [torch.rand(count + offset, dtype=tdt, device=gpu)[offset:].view(shape) for _ in range(3)]
    ptrs = osg._lib.ptr_table(fs)
    n = lib.tpg_y_halo_buffer_elems(3, Nx, Nz, Hx, Hy, Hz)
    assert n == 3 * (Nx + 2 * Hx) * Hy * (Nz + 2 * Hz)
    for side, rows in ((0, slice(Hy, 2 * Hy)), (1, slice(Ny, Ny + Hy))):
        buf = torch.empty(n, dtype=tdt, device=gpu)
        assert lib.tpg_pack_y_halo(ptrs, 3, buf.data_ptr(), side, *size, *halo, ft, None) == 0
        torch.cuda.synchronize()
        want = torch.stack([f[:, rows] for f in fs]).flatten()
        assert torch.equal(buf, want)
    for side, rows in ((0, slice(0, Hy)), (1, slice(Ny + Hy, Ny + 2 * Hy))):
        buf = torch.rand(n, dtype=tdt, device=gpu)
        before = [f.clone() for f in fs]
        assert lib.tpg_unpack_y_halo(ptrs, 3, buf.data_ptr(), side, *size, *halo, ft, None) == 0
        torch.cuda.synchronize()
        msg = buf.view(3, Nz + 2 * Hz, Hy, Nx + 2 * Hx)
        for f, b, m in zip(fs, before, msg):
            assert torch.equal(f[:, rows], m)
            b[:, rows] = m
            assert torch.equal(f, b)


@pytest.mark.parametrize("halo", [(4, 4, 4), (5, 5, 5)], ids=["halo4", "halo5"])      # (5, 5, 5): the halo of the reference's distributed example
def test_config4_eight_bands_at_full_size(osg, gpu, tlib, halo):
    """BASELINE config 4 at its real geometry: the 1/10 degree grid (3600 x 1800 x 75, halo 4, Float64) as 8 latitude bands of
    ny = 225 rows, the four bench fields c / u / v / zeta, on ONE GPU with the 8 ranks emulated in this process (two-phase
    loop-back transport; the RCCL leg itself is tests/test_gpu_exchange.py).
      * every rank's band build equals rows jstart-Hy .. jend+Hy of the serially built global grid (the serial build is
        bit-exact against the oracle at this size: tests/test_gpu_grid.py);
      * after fill_halo_regions! on every rank (zipper on rank 7 only, periodic x, seam exchange of 4 x 9.58 MB per side) every
        rank's padded slab equals the same rows of the serially filled global field (the serial fill is bit-exact against the
        oracle at this size: tests/test_gpu_zipper.py::test_config3_tenth_degree_75_levels).
    Everything is compared on the device: 17 GB of global fields + 17 GB filled copies + 18 GB of slabs."""
    import ctypes as C
    import gc
    gc.collect(); torch.cuda.empty_cache()
    size, R = (3600, 1800, 75), 8
    (Nx, Ny, Nz), (Hx, Hy, Hz) = size, halo
    lib = osg._lib.lib()
    specs = [(0, 0, 1), (1, 0, -1), (0, 1, -1), (1, 1, 1)]
    serial = osg.TripolarGrid(size=size, halo=halo)
    globs, filled = [], []
    for fid, (xl, yl, sg) in enumerate(specs):
        loc = (osg.Face if xl else osg.Center, osg.Face if yl else osg.Center, osg.Center)
        f = osg.Field(loc, serial)
        assert tlib.tpg_fill_synthetic(f.data.data_ptr(), 0xC4 + fid, 12345.0, *size, *halo, 1, None) == 0
        globs.append(f.data.clone())
        filled.append(f)
    osg.fill_halo_regions(filled)                                   # the serial reference fill
    ranks = []
    for r in range(R):
        arch = osg.Distributed(osg.GPU(0), osg.Partition(y=R), local_rank=r)
        grid = osg.TripolarGrid(arch, size=size, halo=halo)
        jstart, jend = grid.jrange
        assert (jstart, jend) == (1 + 225 * r, 225 * (r + 1)) and grid.Ny == 225
        for name in osg._lib.ARRAY_NAMES:
            assert torch.equal(getattr(grid, name), getattr(serial, name)[jstart - 1:jend + 2 * Hy]), (r, name)
        fs = []
        for (xl, yl, sg), g in zip(specs, globs):
            loc = (osg.Face if xl else osg.Center, osg.Face if yl else osg.Center, osg.Center)
            f = osg.Field(loc, grid)
            assert osg.is_zipper(f.boundary_conditions.north) == (r == R - 1)
            f.data.copy_(g[:, jstart - 1:jend + 2 * Hy])
            f.data[:, :Hy] = 12345.0
            f.data[:, Hy + 225:] = 12345.0                          # halo rows unknown before the fill
            fs.append(f)
        ranks.append((grid, fs))
    mailbox = osg.LoopbackMailbox()
    plans = [osg.halo_fill_plan(fs, exchange=mailbox.endpoint(r)) for r, (grid, fs) in enumerate(ranks)]
    for plan in plans:
        plan.begin()
    for plan in plans:
        plan.finish()
    torch.cuda.synchronize()
    for r, (grid, fs) in enumerate(ranks):
        jstart, jend = grid.jrange
        for f, ref in zip(fs, filled):
            assert torch.equal(f.data, ref.data[:, jstart - 1:jend + 2 * Hy]), (r, f.loc)


class _PeerPeek:
    """A PLAIN-CALLABLE transport (no post/wait, no cloning) for R ranks emulated in one process: at finish() time it copies,
    batch by batch, straight out of the peer exchange's own send buffers.  Any sharing of message buffers between batches or
    between plans that are in flight together shows up as wrong halos (the module-global buffer cache did that)."""

    def __init__(self):
        self.exchanges = {}                      # (tag, rank) -> PendingExchange

    def endpoint(self, tag, me):
        state = {"calls": 0}

        def transport(plan, send, recv, group):
            mine = self.exchanges[(tag, me)]
            b = state["calls"] % len(mine.batches)
            state["calls"] += 1
            for m in plan:
                peer = self.exchanges[(tag, m.peer)].batches[b][2]
                recv[m.side].copy_(peer.send[1 - m.side])          # what the peer packed for the side that faces me
        return transport


def _band_fields(osg, grid, globs, specs, tdt, sent=SENT):
    jstart, jend = grid.jrange
    Hy = grid.Hy
    fs = []
    for (xl, yl, sg), g in zip(specs, globs):
        loc = (osg.Face if xl else osg.Center, osg.Face if yl else osg.Center, osg.Center)
        f = osg.Field(loc, grid)
        slab = g[:, jstart - 1:jend + 2 * Hy].copy()
        slab[:, :Hy] = sent; slab[:, Hy + (jend - jstart + 1):] = sent
        f.data.copy_(torch.from_numpy(slab))
        fs.append(f)
    return fs


def test_35_fields_through_a_plain_callable_transport(osg, oracle, gpu):
    """more fields than one batch (TPG_MAX_FIELDS = 16): three batches, two of them of equal shape, every batch with its own
    message buffers, all packed in begin() before any is delivered in finish()"""
    size, halo, R = (24, 16, 2), (4, 4, 1), 2
    (Nx, Ny, Nz), (Hx, Hy, Hz) = size, halo
    rng = np.random.default_rng(35)
    specs = [(int(rng.integers(0, 2)), int(rng.integers(0, 2)), 0) for _ in range(35)]
    specs = [(x, y, -1 if x != y else 1) for x, y, _ in specs]               # the default sign policy by location
    globs = []
    for _ in specs:
        g = rng.uniform(-1, 1, (Nz + 2 * Hz, Ny + 2 * Hy, Nx + 2 * Hx))
        g[:, :Hy] = SENT; g[:, Hy + Ny:] = SENT
        globs.append(g)
    peek = _PeerPeek()
    plans, ranks = [], []
    for r in range(R):
        grid = osg.TripolarGrid(osg.Distributed(osg.GPU(0), osg.Partition(y=R), local_rank=r), torch.float64, size=size, halo=halo)
        fs = _band_fields(osg, grid, globs, specs, torch.float64)
        plan = osg.halo_fill_plan(fs, exchange=peek.endpoint("a", r))
        (_, _, pending), = plan._steps
        assert len(pending.batches) == 3 and [len(b[0]) for b in pending.batches] == [16, 16, 3]
        bufs = [b[2].send[side].data_ptr() for b in pending.batches for side in b[2].send]
        assert len(set(bufs)) == len(bufs)                                      # no two batches share a message buffer
        peek.exchanges[("a", r)] = pending
        plans.append(plan); ranks.append((grid, fs))
    for plan in plans:
        plan.begin()
    for plan in plans:
        plan.finish()
    torch.cuda.synchronize()
    for (xl, yl, sg), g in zip(specs, globs):
        oracle.fill_halo_regions(g, xl, yl, sg, size, halo)
    for r, (grid, fs) in enumerate(ranks):
        jstart, jend = grid.jrange
        for k, (f, g) in enumerate(zip(fs, globs)):
            assert np.array_equal(f.data.cpu().numpy(), g[:, jstart - 1:jend + 2 * Hy]), (r, k, f.loc)


def test_two_plans_of_equal_geometry_in_flight_together(osg, oracle, gpu):
    """two HaloFillPlans over different fields of the SAME geometry, begun back to back and only then finished: each plan owns
    its message buffers, so the second pack cannot overwrite the first plan's posted message"""
    size, halo, R = (32, 24, 3), (4, 4, 2), 3
    (Nx, Ny, Nz), (Hx, Hy, Hz) = size, halo
    rng = np.random.default_rng(77)
    sets = {}
    for tag in ("a", "b"):
        globs = []
        for _ in SPECS:
            g = rng.uniform(-1, 1, (Nz + 2 * Hz, Ny + 2 * Hy, Nx + 2 * Hx))
            g[:, :Hy] = SENT; g[:, Hy + Ny:] = SENT
            globs.append(g)
        sets[tag] = globs
    peek = _PeerPeek()
    plans = {"a": [], "b": []}
    ranks = {"a": [], "b": []}
    for r in range(R):
        grid = osg.TripolarGrid(osg.Distributed(osg.GPU(0), osg.Partition(y=R), local_rank=r), torch.float64, size=size, halo=halo)
        for tag in ("a", "b"):
            fs = _band_fields(osg, grid, sets[tag], SPECS, torch.float64)
            plan = osg.halo_fill_plan(fs, exchange=peek.endpoint(tag, r))
            peek.exchanges[(tag, r)] = plan._steps[0][2]
            plans[tag].append(plan); ranks[tag].append((grid, fs))
    for r in range(R):
        a, b = peek.exchanges[("a", r)].batches[0][2], peek.exchanges[("b", r)].batches[0][2]
        assert not {t.data_ptr() for t in a.send.values()} & {t.data_ptr() for t in b.send.values()}
    for tag in ("a", "b"):                      # begin A on every rank, begin B on every rank ...
        for plan in plans[tag]:
            plan.begin()
    for tag in ("a", "b"):                      # ... and only then deliver
        for plan in plans[tag]:
            plan.finish()
    torch.cuda.synchronize()
    with pytest.raises(RuntimeError, match="begin"):
        plans["a"][1]._steps[0][2].finish()     # a second finish without begin is refused
    for tag in ("a", "b"):
        for (xl, yl, sg), g in zip(SPECS, sets[tag]):
            oracle.fill_halo_regions(g, xl, yl, sg, size, halo)
        for r, (grid, fs) in enumerate(ranks[tag]):
            jstart, jend = grid.jrange
            for f, g in zip(fs, sets[tag]):
                assert np.array_equal(f.data.cpu().numpy(), g[:, jstart - 1:jend + 2 * Hy]), (tag, r, f.loc)


def test_halo_fill_plan_marshals_the_one_call_distributed_fill(osg, gpu, monkeypatch):
    """With an RcclComm on the architecture a HaloFillPlan issues ONE C call per batch, tpg_fill_halo_regions_distributed.  No second
    RCCL rank exists on a one-GPU box, so the C function is replaced here by a recorder that checks every argument the plan hands
    over (order, kinds, which seam buffers are present for which rank, geometry, the zipper tables on the last rank only) and runs
    the local part of the fill; the C function itself runs against a real communicator in tools/rccl_selftest.py."""
    import ctypes as C
    from orthogonalsphericalshellgrids.jl_amd.distributed import RcclComm
    lib = osg._lib.lib()
    size, halo, R = (32, 24, 2), (4, 4, 1), 3
    calls = []

    def recorder(comm, rank, nranks, fields, nfields, xl, yl, sg, ss, sn, rs, rn, Nx, Ny, Nz, Hx, Hy, Hz, ft, stream):
        calls.append(dict(comm=comm, rank=rank, nranks=nranks, nfields=nfields, tables=(xl, yl, sg), bufs=(ss, sn, rs, rn),
                          geom=(Nx, Ny, Nz, Hx, Hy, Hz), ft=ft, stream=stream, field0=fields[0]))
        return lib.tpg_fill_halo_regions(fields, nfields, xl, yl, sg, Nx, Ny, Nz, Hx, Hy, Hz, 1 if rank == nranks - 1 else 0, ft, stream)

    monkeypatch.setattr(lib, "tpg_fill_halo_regions_distributed", recorder, raising=True)
    for r in range(R):
        comm = RcclComm(C.c_void_p(0xC0FFEE), r, R)
        arch = osg.Distributed(osg.GPU(0), osg.Partition(y=R), local_rank=r, rccl_comm=comm)
        grid = osg.TripolarGrid(arch, torch.float64, size=size, halo=halo)
        fs = [osg.CenterField(grid), osg.XFaceField(grid), osg.YFaceField(grid)]
        for f in fs:
            f.data.copy_(torch.rand_like(f.data))
        before = [f.data.clone() for f in fs]
        plan = osg.halo_fill_plan(fs)
        assert plan.is_distributed and all(p is None for _, _, p in plan._steps)          # one C call, no Python-side exchange object
        with pytest.raises(ValueError):
            plan.graph()                                                                    # a seam cannot be captured
        calls.clear()
        plan()
        torch.cuda.synchronize()
        assert len(calls) == 1
        c = calls[0]
        assert c["comm"].value == 0xC0FFEE and (c["rank"], c["nranks"], c["nfields"]) == (r, R, 3)
        assert c["geom"] == (32, 8, 2, 4, 4, 1) and c["ft"] == 1 and c["field0"] == fs[0].data.data_ptr()
        ss, sn, rs, rn = c["bufs"]
        assert (ss is not None) == (r > 0) == (rs is not None) and (sn is not None) == (r < R - 1) == (rn is not None)
        present = [b for b in c["bufs"] if b is not None]
        assert len(set(present)) == len(present)                                           # four distinct message buffers
        if r == R - 1:                                                                      # the zipper tables travel on the last rank only
            xl, yl, sg = c["tables"]
            assert list(xl) == [0, 1, 0] and list(yl) == [0, 0, 1] and list(sg) == [1, -1, -1]
        else:
            assert c["tables"] == (None, None, None)
        # the local part ran: x halos are periodic images, and only the last rank's north halo was folded
        for f, b in zip(fs, before):
            assert torch.equal(f.data[:, :, :4], f.data[:, :, 32:36])
            if r < R - 1:
                assert torch.equal(f.data[:, 12:, 4:36], b[:, 12:, 4:36])
    # pack-free plans hand over no buffers at all
    comm = RcclComm(C.c_void_p(0xC0FFEE), 1, R)
    grid = osg.TripolarGrid(osg.Distributed(osg.GPU(0), osg.Partition(y=R), local_rank=1, rccl_comm=comm), torch.float64, size=size, halo=halo)
    calls.clear()
    osg.halo_fill_plan([osg.CenterField(grid)], pack_free=True)()
    assert calls[0]["bufs"] == (None, None, None, None)


def test_halo_fill_plan_marshals_the_pipelined_distributed_fill(osg, gpu, monkeypatch):
    """fields_per_stage = k on a plan whose architecture carries an RcclComm: ONE call of tpg_fill_halo_regions_distributed_pipelined per
    batch, with the plan's own second stream and the stage size AFTER the caller's stream (the recorder checks order and kinds and runs
    the local fill; the C function itself runs against a real communicator in tools/rccl_selftest.py and bench.py --loopback);
    pack_free + pipelined is refused."""
    import ctypes as C
    from orthogonalsphericalshellgrids.jl_amd.distributed import RcclComm
    lib = osg._lib.lib()
    size, halo, R = (32, 24, 2), (4, 4, 1), 3
    calls = []

    def recorder(comm, rank, nranks, fields, nfields, xl, yl, sg, ss, sn, rs, rn, Nx, Ny, Nz, Hx, Hy, Hz, ft, stream, comm_stream, fps):
        calls.append(dict(rank=rank, nranks=nranks, nfields=nfields, bufs=(ss, sn, rs, rn), stream=stream, comm_stream=comm_stream, fps=fps))
        return lib.tpg_fill_halo_regions(fields, nfields, xl, yl, sg, Nx, Ny, Nz, Hx, Hy, Hz, 1 if rank == nranks - 1 else 0, ft, stream)

    monkeypatch.setattr(lib, "tpg_fill_halo_regions_distributed_pipelined", recorder, raising=True)
    monkeypatch.setattr(lib, "tpg_fill_halo_regions_distributed", lambda *a: pytest.fail("monolithic entry point called by a pipelined plan"), raising=True)
    comm = RcclComm(C.c_void_p(0xC0FFEE), 1, R)
    arch = osg.Distributed(osg.GPU(0), osg.Partition(y=R), local_rank=1, rccl_comm=comm)
    grid = osg.TripolarGrid(arch, torch.float64, size=size, halo=halo)
    fs = [osg.CenterField(grid), osg.XFaceField(grid), osg.YFaceField(grid)]
    plan = osg.halo_fill_plan(fs, fields_per_stage=2)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        plan()
        plan()
    torch.cuda.synchronize()
    assert len(calls) == 2 and all(c["fps"] == 2 and c["nfields"] == 3 and (c["rank"], c["nranks"]) == (1, R) for c in calls)
    assert all(c["stream"].value == side.cuda_stream for c in calls)                       # the caller's current stream ...
    cs = {c["comm_stream"].value for c in calls}
    assert len(cs) == 1 and cs != {side.cuda_stream} and None not in cs                    # ... and ONE second stream, owned by the plan
    assert all(b is not None for b in calls[0]["bufs"])                                    # a middle band: all four message buffers
    with pytest.raises(ValueError):
        osg.halo_fill_plan(fs, fields_per_stage=1, pack_free=True)
    with pytest.raises(ValueError):
        osg.halo_fill_plan(fs, fields_per_stage=-1)
