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


def test_config4_eight_bands_at_full_size(osg, gpu):
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
    size, halo, R = (3600, 1800, 75), (4, 4, 4), 8
    (Nx, Ny, Nz), (Hx, Hy, Hz) = size, halo
    lib = osg._lib.lib()
    specs = [(0, 0, 1), (1, 0, -1), (0, 1, -1), (1, 1, 1)]
    serial = osg.TripolarGrid(size=size, halo=halo)
    globs, filled = [], []
    for fid, (xl, yl, sg) in enumerate(specs):
        loc = (osg.Face if xl else osg.Center, osg.Face if yl else osg.Center, osg.Center)
        f = osg.Field(loc, serial)
        assert lib.tpg_fill_synthetic(f.data.data_ptr(), 0xC4 + fid, 12345.0, *size, *halo, 1, None) == 0
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
