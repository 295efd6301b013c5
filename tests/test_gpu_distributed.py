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


@pytest.mark.parametrize("R", [2, 4])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_band_halo_fill_with_loopback_transport(osg, oracle, gpu, R, dtype):
    size, halo = (48, 40, 3), (4, 4, 2)
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
    plans = [osg.halo_fill_plan(fs, exchange=mailbox.endpoint(r)) for r, (arch, grid, fs) in enumerate(ranks)]
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


def test_pack_unpack_roundtrip_and_layout(osg, gpu):
    """message layout [field][level][Hy][sx]; pack reads interior rows, unpack writes halo rows"""
    lib = osg._lib.lib()
    size, halo = (20, 12, 2), (4, 3, 1)
    (Nx, Ny, Nz), (Hx, Hy, Hz) = size, halo
    shape = (Nz + 2 * Hz, Ny + 2 * Hy, Nx + 2 * Hx)
    fs = [torch.rand(shape, dtype=torch.float64, device=gpu) for _ in range(3)]
    ptrs = osg._lib.ptr_table(fs)
    n = lib.tpg_y_halo_buffer_elems(3, Nx, Nz, Hx, Hy, Hz)
    assert n == 3 * (Nx + 2 * Hx) * Hy * (Nz + 2 * Hz)
    for side, rows in ((0, slice(Hy, 2 * Hy)), (1, slice(Ny, Ny + Hy))):
        buf = torch.empty(n, dtype=torch.float64, device=gpu)
        assert lib.tpg_pack_y_halo(ptrs, 3, buf.data_ptr(), side, *size, *halo, 1, None) == 0
        torch.cuda.synchronize()
        want = torch.stack([f[:, rows] for f in fs]).flatten()
        assert torch.equal(buf, want)
    for side, rows in ((0, slice(0, Hy)), (1, slice(Ny + Hy, Ny + 2 * Hy))):
        buf = torch.rand(n, dtype=torch.float64, device=gpu)
        before = [f.clone() for f in fs]
        assert lib.tpg_unpack_y_halo(ptrs, 3, buf.data_ptr(), side, *size, *halo, 1, None) == 0
        torch.cuda.synchronize()
        msg = buf.view(3, Nz + 2 * Hz, Hy, Nx + 2 * Hx)
        for f, b, m in zip(fs, before, msg):
            assert torch.equal(f[:, rows], m)
            b[:, rows] = m
            assert torch.equal(f, b)
