"""GPU parity of the Zipper halo fill (tpg_zipper_fill / tpg_fill_halo_regions) against the
oracle: bit-exact on the whole padded array (integer index map + sign)."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from helpers import synthetic_field

pytestmark = pytest.mark.gpu

LOCS = [(0, 0), (1, 0), (0, 1), (1, 1)]


def loc_types(osg, xl, yl):
    return (osg.Face if xl else osg.Center, osg.Face if yl else osg.Center, osg.Center)


def make_fields(osg, grid, specs, rng, dtype):
    fs, hosts = [], []
    for xl, yl, sg in specs:
        f = osg.Field(loc_types(osg, xl, yl), grid,
                      boundary_conditions=osg.FieldBoundaryConditions(north=osg.ZipperBoundaryCondition(sg)))
        h = rng.uniform(-1, 1, tuple(f.data.shape)).astype(dtype)
        f.data.copy_(torch.from_numpy(h))
        fs.append(f); hosts.append(h)
    return fs, hosts


def test_reference_zipper_testset(osg, gpu, kats):
    """test/test_zipper_boundary_conditions.jl:5-45 through the product API"""
    k = kats["zipper_10x10"]
    grid = osg.TripolarGrid(size=tuple(k["size"]))
    Nx, Ny, _ = grid.size
    Hx, Hy, Hz = grid.halo_size
    c, u, v = osg.CenterField(grid), osg.XFaceField(grid), osg.YFaceField(grid)
    for f, name in ((c, "c"), (u, "u"), (v, "v")):
        north = f.boundary_conditions.north
        assert isinstance(north.classification, osg.Zipper)                    # :14-16
        assert north.condition == k["default_sign"][name]                      # :21-23
        osg.set_(f, 1)
    osg.fill_halo_regions(c); osg.fill_halo_regions(u); osg.fill_halo_regions(v)
    north = lambda f: f.data[Hz, Ny + Hy:Ny + 2 * Hy, :]
    e = k["constant_one"]
    assert bool((north(c) == e["c_north_halo"]).all())                         # :35
    assert bool((north(v) == e["v_north_halo"]).all())                         # :36
    assert bool((north(u)[:, Hx + 1:Hx + Nx - 1] == e["u_north_halo_i_2_to_Nx_minus_1"]).all())   # :39-40
    assert bool((north(u)[:, Hx] == e["u_north_halo_i_1"]).all())              # :42,44
    assert bool((north(u)[:, Hx + Nx] == e["u_north_halo_i_Nx_plus_1"]).all())  # :43,45


def test_reference_row_symmetry_testset(osg, gpu):
    """:47-72: bottom_height-like reduced field, c(x), u(x) rows"""
    grid = osg.TripolarGrid(size=(10, 10, 1))
    bottom = osg.Field((osg.Center, osg.Center, None), grid)                   # (Center, Center, Nothing)
    bottom.set_(torch.rand(10, 10, dtype=torch.float64, device=gpu))
    osg.fill_halo_regions(bottom)
    row = bottom.interior()[0, 9]
    assert torch.equal(row, row.flip(0))                                       # :54
    c, u = osg.CenterField(grid), osg.XFaceField(grid)
    c.set_(lambda x, y, z: x); u.set_(lambda x, y, z: x)
    osg.fill_halo_regions([c, u])
    crow, urow = c.interior()[0, 9], u.interior()[0, 9]
    assert torch.equal(crow, crow.flip(0))                                     # :65
    assert torch.equal(urow[1:5], -urow[6:10].flip(0))                         # :68-72


GEOMS = [((10, 10, 1), (4, 4, 4)), ((60, 30, 3), (4, 4, 4)), ((62, 31, 2), (3, 2, 1)),   # odd Hx: chunks straddle the interior / x-halo boundary (GEN kernels)
         ((256, 40, 4), (4, 4, 2)), ((1000, 50, 2), (4, 4, 4)), ((130, 20, 3), (2, 5, 0)),
         ((6, 7, 1), (4, 4, 1)), ((4, 4, 2), (4, 4, 1)),
         ((64, 40, 1), (4, 13, 1)),      # extended north halo of the split-explicit free surface (test/runtests.jl:61-71): Hy > 8
         ((3600, 24, 2), (4, 4, 4)),     # full 1/10 degree rows
         # halo = (5, 5, 5): the reference's own model geometry (examples/bickley_jet.jl:21, examples/distributed_bickley_jet.jl:23)
         ((10, 12, 1), (5, 5, 5)), ((60, 30, 3), (5, 5, 5)), ((3600, 24, 2), (5, 5, 5)), ((64, 44, 1), (5, 13, 1)),
         # the other chunk geometries: Float32 8-B chunks (row pitch 2 mod 4), Nx = 2 mod 4 with an odd Hx, halo wider than one chunk + leftover
         ((128, 20, 2), (5, 4, 1)), ((130, 20, 3), (3, 5, 0)), ((66, 20, 2), (6, 3, 2)), ((64, 16, 2), (7, 8, 1)), ((12, 12, 2), (5, 5, 1)),
         ((10, 11, 2), (1, 1, 1))]


class _Knob:
    """TPG_FILL_FUSED through the environment + tpg_reload_config() (the library reads its knobs once)"""

    def __init__(self, lib):
        self.lib = lib

    def __setitem__(self, k, v):
        os.environ[k] = v
        self.lib.tpg_reload_config()

    def pop(self, k, *a):
        os.environ.pop(k, None)
        self.lib.tpg_reload_config()


@pytest.fixture
def fused_knob(osg, via_testlib):
    """the knobs live in the test library only: the package's calls go through it for the duration of the test"""
    saved = {k: os.environ.get(k) for k in ("TPG_FILL_FUSED", "TPG_FILL_MERGED")}
    yield _Knob(via_testlib)
    for k, v in saved.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v
    via_testlib.tpg_reload_config()


@pytest.mark.parametrize("fused,merged", [("0", "0"), ("1", "0"), ("0", "1"), ("2", "0")], ids=["two-launch", "fused", "merged", "fused-cells"])
@pytest.mark.parametrize("size,halo", GEOMS, ids=[f"{s}-{h}" for s, h in GEOMS])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_fill_halo_regions_parity(osg, oracle, gpu, fused_knob, size, halo, dtype, fused, merged):
    """zipper -> periodic x as two launches, as the single fused launch small fields take by default (chunk items; "fused-cells": its
    one-thread-per-cell cross-check form), and as the single merged launch large fields take (geometries a single-launch form does not
    cover -- Nx < 2 Hx + 2, Ny < 2 Hy + 2, Hy > 8 for the merged form -- fall back by themselves).  Every geometry has a chunked form:
    plain 16-B chunks for halo (4, 4, 4)-like geometries, the element-aligned GEN form for an odd Hx, Float32 with Nx = 2 mod 4, ..."""
    fused_knob["TPG_FILL_FUSED"] = fused
    fused_knob["TPG_FILL_MERGED"] = merged
    tdt = torch.float64 if dtype == np.float64 else torch.float32
    grid = osg.TripolarGrid(osg.GPU(0), tdt, size=size, halo=halo)
    specs = [(xl, yl, sg) for xl, yl in LOCS for sg in (1, -1)]
    fs, hosts = make_fields(osg, grid, specs, np.random.default_rng(5), dtype)
    osg.fill_halo_regions(fs)                                                  # one batched launch
    for f, h, (xl, yl, sg) in zip(fs, hosts, specs):
        oracle.fill_halo_regions(h, xl, yl, sg, size, halo)
        assert np.array_equal(f.data.cpu().numpy(), h), (xl, yl, sg)


def test_mixed_geometries_in_one_call(osg, oracle, gpu):
    """a tupled fill_halo_regions!((u, v, c, eta)) mixes 3-D fields with a reduced (Nz = 1, Hz = 0) one"""
    size, halo = (60, 30, 3), (4, 4, 4)
    grid = osg.TripolarGrid(size=size, halo=halo)
    rng = np.random.default_rng(8)
    u, v, c = osg.XFaceField(grid), osg.YFaceField(grid), osg.CenterField(grid)
    eta = osg.Field((osg.Center, osg.Center, None), grid)
    hosts = []
    for f in (u, v, c, eta):
        h = rng.uniform(-1, 1, tuple(f.data.shape)); f.data.copy_(torch.from_numpy(h)); hosts.append(h)
    osg.fill_halo_regions((u, v, c, eta))
    for f, h, (xl, yl, sg) in zip((u, v, c, eta), hosts, ((1, 0, -1), (0, 1, -1), (0, 0, 1), (0, 0, 1))):
        sz = (size[0], size[1], f.Nz); hl = (halo[0], halo[1], f.Hz)
        oracle.fill_halo_regions(h, xl, yl, sg, sz, hl)
        assert np.array_equal(f.data.cpu().numpy(), h), f.loc


def test_unhandled_boundary_conditions_are_refused(osg, gpu):
    """south / bottom / top conditions are Oceananigans' to fill: a field carrying one is refused, not returned stale"""
    grid = osg.TripolarGrid(size=(10, 10, 1))
    value = osg.BoundaryCondition(object(), 0.0)
    per = osg.PeriodicBoundaryCondition
    for side in ("south", "bottom", "top"):
        c = osg.CenterField(grid, boundary_conditions=osg.FieldBoundaryConditions(west=per(), east=per(), **{side: value}))
        with pytest.raises(NotImplementedError):
            osg.fill_halo_regions(c)
    c = osg.CenterField(grid, boundary_conditions=osg.FieldBoundaryConditions(west=value, east=per()))
    with pytest.raises(NotImplementedError):
        osg.fill_halo_regions(c)


def test_zipper_only_and_level_range(osg, oracle, gpu):
    """tpg_zipper_fill leaves x halos and unlisted levels untouched; kstart/kcount may address halo levels"""
    size, halo = (64, 20, 5), (4, 4, 2)
    lib = osg._lib.lib()
    rng = np.random.default_rng(3)
    for xl, yl in LOCS:
        h = rng.uniform(-1, 1, (5 + 4, 20 + 8, 64 + 8))
        d = torch.from_numpy(h).to(gpu)
        for kstart, kcount in ((2, 3), (-1, 9)):
            want = h.copy()
            oracle.zipper_fill(want, xl, yl, -1, size, halo, kstart, kcount)
            got = d.clone()
            rc = lib.tpg_zipper_fill(osg._lib.ptr_table([got]), 1, (C.c_int8 * 1)(xl), (C.c_int8 * 1)(yl), (C.c_int32 * 1)(-1),
                                     *size, *halo, kstart, kcount, 1, None)
            assert rc == 0
            torch.cuda.synchronize()
            assert np.array_equal(got.cpu().numpy(), want), (xl, yl, kstart)


def test_more_fields_than_one_launch_holds(osg, oracle, gpu):
    size, halo = (32, 12, 2), (4, 4, 4)
    grid = osg.TripolarGrid(size=size, halo=halo)
    specs = [(f % 2, (f // 2) % 2, 1 if f % 3 else -1) for f in range(37)]     # > 2 x TPG_MAX_FIELDS
    fs, hosts = make_fields(osg, grid, specs, np.random.default_rng(11), np.float64)
    osg.fill_halo_regions(fs)
    for f, h, (xl, yl, sg) in zip(fs, hosts, specs):
        oracle.fill_halo_regions(h, xl, yl, sg, size, halo)
        assert np.array_equal(f.data.cpu().numpy(), h)


def test_unaligned_base_pointer_takes_the_element_aligned_kernels(osg, oracle, gpu):
    """a field that is only 8-B aligned: the GEN (element-aligned) form of the column kernel, and of the merged / fused fill"""
    size, halo = (32, 12, 2), (4, 4, 4)
    shape = (2 + 8, 12 + 8, 32 + 8)
    n = int(np.prod(shape))
    raw = torch.zeros(n + 1, dtype=torch.float64, device=gpu)
    view = raw[1:]                                                              # 8-byte aligned only
    h = np.random.default_rng(2).uniform(-1, 1, shape)
    view.copy_(torch.from_numpy(h).flatten())
    lib = osg._lib.lib()
    ptr = (C.c_void_p * 1)(view.data_ptr())
    assert lib.tpg_zipper_fill(ptr, 1, (C.c_int8 * 1)(1), (C.c_int8 * 1)(0), (C.c_int32 * 1)(-1), *size, *halo, 1, 2, 1, None) == 0
    torch.cuda.synchronize()
    oracle.zipper_fill(h, 1, 0, -1, size, halo)
    assert np.array_equal(view.cpu().numpy().reshape(shape), h)
    h = np.random.default_rng(3).uniform(-1, 1, shape)
    view.copy_(torch.from_numpy(h).flatten())
    assert lib.tpg_fill_halo_regions(ptr, 1, (C.c_int8 * 1)(1), (C.c_int8 * 1)(0), (C.c_int32 * 1)(-1), *size, *halo, 1, 1, None) == 0
    torch.cuda.synchronize()
    oracle.fill_halo_regions(h, 1, 0, -1, size, halo)
    assert np.array_equal(view.cpu().numpy().reshape(shape), h)


def test_synthetic_fill_matches_host_twin(osg, gpu, tlib):
    size, halo = (24, 10, 3), (4, 4, 2)
    lib = tlib
    for dt, tdt, ft in ((np.float64, torch.float64, 1), (np.float32, torch.float32, 0)):
        d = torch.empty((3 + 4, 10 + 8, 24 + 8), dtype=tdt, device=gpu)
        assert lib.tpg_fill_synthetic(d.data_ptr(), 0x5EED + 1, 12345.0, *size, *halo, ft, None) == 0
        torch.cuda.synchronize()
        assert np.array_equal(d.cpu().numpy(), synthetic_field(0x5EED + 1, 12345.0, size, halo, dt))


@pytest.mark.parametrize("h", [4, 5], ids=["halo4", "halo5"])
def test_config3_tenth_degree_75_levels(osg, oracle, gpu, tlib, h):
    """BASELINE config 3: (3600,1800,75), halo 4 -- and halo 5, the reference's model halo (examples/bickley_jet.jl:21): k_zipper_cols in its
    GEN form --, Float64, fields c(CC,+1) u(FC,-1) v(CF,-1) zeta(FF,+1),
    splitmix64 interior / sentinel halos.  Full-size checks:
      * bit-exact parity of every row the fold can touch (rows Ny-Hy..Ny+Hy, all levels) against the
        oracle run on that slab as a short (Ny' = 2Hy+1) field -- the fold only looks Hy rows down;
      * everything below is untouched (checksum of the raw bits before/after);
      * x halos and z-halo levels of the north rows keep the sentinel (zipper only, no periodic pass).
    """
    size, halo = (3600, 1800, 75), (h, h, h)
    (Nx, Ny, Nz), (Hx, Hy, Hz) = size, halo
    lib = osg._lib.lib()
    specs = [(0, 0, 1), (1, 0, -1), (0, 1, -1), (1, 1, 1)]
    shape = (Nz + 2 * Hz, Ny + 2 * Hy, Nx + 2 * Hx)
    fields = []
    for fid, _ in enumerate(specs):
        d = torch.empty(shape, dtype=torch.float64, device=gpu)
        assert tlib.tpg_fill_synthetic(d.data_ptr(), 0x5EED + fid, 12345.0, *size, *halo, 1, None) == 0
        fields.append(d)
    top = slice(Ny - 1, Ny + 2 * Hy)            # parent rows of logical rows Ny-Hy .. Ny+Hy
    before = [f[:, top].cpu().numpy() for f in fields]
    low_sum = [int(f[:, :Ny - 1].view(torch.int64).sum()) for f in fields]
    n = len(specs)
    rc = lib.tpg_zipper_fill(osg._lib.ptr_table(fields), n, (C.c_int8 * n)(*[s[0] for s in specs]),
                             (C.c_int8 * n)(*[s[1] for s in specs]), (C.c_int32 * n)(*[s[2] for s in specs]),
                             *size, *halo, 1, Nz, 1, None)
    assert rc == 0
    torch.cuda.synchronize()
    slab_size = (Nx, Hy + 1, Nz)                # short field whose top row is the global row Ny
    for f, b, (xl, yl, sg), ls in zip(fields, before, specs, low_sum):
        # slab rows: [Ny-Hy .. Ny] interior-like + Hy halo rows; pad Hy dummy south-halo rows for the oracle
        want = np.concatenate([np.zeros_like(b[:, :Hy]), b], axis=1)
        oracle.zipper_fill(want, xl, yl, sg, slab_size, halo)
        got = f[:, top].cpu().numpy()
        assert np.array_equal(got, want[:, Hy:]), (xl, yl, sg)
        assert int(f[:, :Ny - 1].view(torch.int64).sum()) == ls
        assert bool((f[:, Ny + Hy:, :Hx] == 12345.0).all()) and bool((f[:Hz, Ny + Hy:] == 12345.0).all())
        assert not bool((f[Hz:Hz + Nz, Ny + Hy:, Hx:Hx + Nx] == 12345.0).any())     # every halo cell of the fold written


@pytest.mark.parametrize("h", [4, 5], ids=["halo4", "halo5"])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_config3_merged_fill(osg, oracle, gpu, tlib, dtype, h):
    """BASELINE config 3 through tpg_fill_halo_regions -- the ONE merged launch (k_fill_merged: zipper fold + periodic x) that
    bench.py's step issues and `roofline` is quoted on -- at full size (3600, 1800, 75), halo 4 (and halo 5, the reference's model
    halo: the GEN form, Float32 in 8-B chunks), the four bench fields, against
    the oracle DIRECTLY (round 3 reached this kernel at this size only through the HIP serial fill):
      * every row the fold or a corner can touch (logical rows Ny-Hy .. Ny+Hy, ALL levels incl. the z halos, x halos included)
        is compared bit for bit with oracle.fill_halo_regions run on that slab as a short (Ny' = Hy+1) field: zipper on
        k = 1..Nz, then periodic x on every level (src/zipper_boundary_condition.jl:70-155, order pinned by
        test/test_zipper_boundary_conditions.jl:42-45);
      * below the slab, on the device: interior bits untouched (checksum), x halos == the wrapped interior columns on every row
        and level, no sentinel left in them."""
    size, halo = (3600, 1800, 75), (h, h, h)
    (Nx, Ny, Nz), (Hx, Hy, Hz) = size, halo
    lib = osg._lib.lib()
    tdt, ft, ity = (torch.float64, 1, torch.int64) if dtype == np.float64 else (torch.float32, 0, torch.int32)
    specs = [(0, 0, 1), (1, 0, -1), (0, 1, -1), (1, 1, 1)]
    shape = (Nz + 2 * Hz, Ny + 2 * Hy, Nx + 2 * Hx)
    fields = []
    for fid, _ in enumerate(specs):
        d = torch.empty(shape, dtype=tdt, device=gpu)
        assert tlib.tpg_fill_synthetic(d.data_ptr(), 0x5EED + fid, 12345.0, *size, *halo, ft, None) == 0
        fields.append(d)
    top = slice(Ny - 1, Ny + 2 * Hy)            # parent rows of logical rows Ny-Hy .. Ny+Hy
    before = [f[:, top].cpu().numpy() for f in fields]
    low_sum = [int(f.view(ity)[:, :Ny - 1, Hx:Hx + Nx].to(torch.int64).sum()) for f in fields]
    n = len(specs)
    rc = lib.tpg_fill_halo_regions(osg._lib.ptr_table(fields), n, (C.c_int8 * n)(*[s[0] for s in specs]),
                                   (C.c_int8 * n)(*[s[1] for s in specs]), (C.c_int32 * n)(*[s[2] for s in specs]),
                                   *size, *halo, 1, ft, None)
    assert rc == 0
    torch.cuda.synchronize()
    for f, b, (xl, yl, sg), ls in zip(fields, before, specs, low_sum):
        want = np.concatenate([np.zeros_like(b[:, :Hy]), b], axis=1)
        oracle.fill_halo_regions(want, xl, yl, sg, (Nx, Hy + 1, Nz), halo)
        assert np.array_equal(f[:, top].cpu().numpy(), want[:, Hy:]), (xl, yl, sg)
        assert int(f.view(ity)[:, :Ny - 1, Hx:Hx + Nx].to(torch.int64).sum()) == ls
        assert torch.equal(f[:, :Ny - 1, :Hx], f[:, :Ny - 1, Nx:Nx + Hx])
        assert torch.equal(f[:, :Ny - 1, Nx + Hx:], f[:, :Ny - 1, Hx:2 * Hx])
        assert not bool((f[Hz:Hz + Nz, Hy:Ny - 1, :Hx] == 12345.0).any()) and not bool((f[Hz:Hz + Nz, Hy:Ny - 1, Nx + Hx:] == 12345.0).any())
        # z-halo levels: no fold there (k = 1..Nz only), but the periodic pass covers them like every other level
        assert bool((f[:Hz, Ny + Hy:, Hx:Hx + Nx] == 12345.0).all()) and bool((f[Hz + Nz:, Ny + Hy:, Hx:Hx + Nx] == 12345.0).all())


def test_randomised_geometries_against_the_oracle(osg, oracle, gpu):
    """80 random (Nx, Ny, Nz, halo, location, sign, dtype, level range) folds straight through the C ABI"""
    lib = osg._lib.lib()
    rng = np.random.default_rng(777)
    for trial in range(80):
        Nx = int(rng.choice([2, 4, 6, 8, 10, 14, 16, 30, 64, 66, 128, 130, 256, 258]))
        Ny = int(rng.integers(1, 25))
        Nz = int(rng.integers(1, 5))
        Hx = int(rng.integers(0, min(Nx, 6) + 1))
        Hy = int(rng.integers(0, min(Ny, 9) + 1))
        Hz = int(rng.integers(0, 3))
        xl, yl = int(rng.integers(0, 2)), int(rng.integers(0, 2))
        sgn = int(rng.choice([1, -1, 2, -3]))                       # bc.condition is any Int in the reference
        dt, tdt, ft = ((np.float64, torch.float64, 1), (np.float32, torch.float32, 0))[trial % 2]
        kstart = int(rng.integers(1 - Hz, Nz + 1))
        kcount = int(rng.integers(0, Nz + Hz - kstart + 2))
        h = rng.uniform(-1, 1, (Nz + 2 * Hz, Ny + 2 * Hy, Nx + 2 * Hx)).astype(dt)
        d = torch.from_numpy(h).to(gpu)
        rc = lib.tpg_zipper_fill(osg._lib.ptr_table([d]), 1, (C.c_int8 * 1)(xl), (C.c_int8 * 1)(yl), (C.c_int32 * 1)(sgn),
                                 Nx, Ny, Nz, Hx, Hy, Hz, kstart, kcount, ft, None)
        assert rc == 0, (trial, lib.tpg_last_error())
        torch.cuda.synchronize()
        if kcount > 0:                                    # Hy = 0 still substitutes row Ny of the y-Center folds (:102,:135)
            oracle.zipper_fill(h, xl, yl, sgn, (Nx, Ny, Nz), (Hx, Hy, Hz), kstart, kcount)
        assert np.array_equal(d.cpu().numpy(), h), (trial, Nx, Ny, Nz, Hx, Hy, Hz, xl, yl, sgn, kstart, kcount)


def test_fused_fill_randomised_against_the_oracle(osg, oracle, gpu, fused_knob):
    """120 random geometries through tpg_fill_halo_regions with the fused kernel forced on (chunk items), the merged kernel forced on,
    the automatic choice, and the fused kernel in its one-thread-per-cell form: whole padded array bit-identical to the oracle's
    zipper -> periodic x sequence"""
    lib = osg._lib.lib()
    rng = np.random.default_rng(4242)
    for trial in range(120):
        fused_knob["TPG_FILL_FUSED"] = "1"
        fused_knob["TPG_FILL_MERGED"] = "0"
        if trial % 4 == 2:
            fused_knob.pop("TPG_FILL_FUSED")
        if trial % 4 == 1:                                   # the merged large-field launch on the same random geometries
            fused_knob["TPG_FILL_FUSED"] = "0"
            fused_knob["TPG_FILL_MERGED"] = "1"
        if trial % 4 == 3:
            fused_knob["TPG_FILL_FUSED"] = "2"
        Nx = int(rng.choice([4, 6, 10, 12, 14, 16, 30, 64, 66, 128, 130, 258]))
        Ny = int(rng.integers(2, 30))
        Nz = int(rng.integers(1, 4))
        Hx = int(rng.integers(1, min(Nx, 6) + 1))
        Hy = int(rng.integers(1, min(Ny, 9) + 1))
        Hz = int(rng.integers(0, 3))
        nf = int(rng.integers(1, 5))
        dt, tdt, ft = ((np.float64, torch.float64, 1), (np.float32, torch.float32, 0))[(trial // 4) % 2]
        specs = [(int(rng.integers(0, 2)), int(rng.integers(0, 2)), int(rng.choice([1, -1, 2]))) for _ in range(nf)]
        hosts = [rng.uniform(-1, 1, (Nz + 2 * Hz, Ny + 2 * Hy, Nx + 2 * Hx)).astype(dt) for _ in specs]
        devs = [torch.from_numpy(h).to(gpu) for h in hosts]
        xl = (C.c_int8 * nf)(*[s[0] for s in specs]); yl = (C.c_int8 * nf)(*[s[1] for s in specs]); sg = (C.c_int32 * nf)(*[s[2] for s in specs])
        rc = lib.tpg_fill_halo_regions(osg._lib.ptr_table(devs), nf, xl, yl, sg, Nx, Ny, Nz, Hx, Hy, Hz, 1, ft, None)
        assert rc == 0, (trial, lib.tpg_last_error())
        torch.cuda.synchronize()
        for d, h, (x, y, sgn) in zip(devs, hosts, specs):
            oracle.fill_halo_regions(h, x, y, sgn, (Nx, Ny, Nz), (Hx, Hy, Hz))
            assert np.array_equal(d.cpu().numpy(), h), (trial, Nx, Ny, Nz, Hx, Hy, Hz, x, y, sgn)


def test_halo_fill_plan_equals_fill_halo_regions(osg, oracle, gpu):
    """a HaloFillPlan built once and called repeatedly gives what fill_halo_regions! gives each time
    (the self-mapped x-Face cell of row Ny flips sign on every fill of a -1 field, as in the reference)"""
    size, halo = (60, 30, 2), (4, 4, 2)
    grid = osg.TripolarGrid(size=size, halo=halo)
    rng = np.random.default_rng(21)
    u, c = osg.XFaceField(grid), osg.CenterField(grid)
    eta = osg.Field((osg.Center, osg.Center, None), grid)
    hosts = []
    for f in (u, c, eta):
        h = rng.uniform(-1, 1, tuple(f.data.shape)); f.data.copy_(torch.from_numpy(h)); hosts.append(h)
    plan = osg.halo_fill_plan((u, c, eta))
    for _ in range(3):
        plan()
        for f, h, (xl, yl, sg) in zip((u, c, eta), hosts, ((1, 0, -1), (0, 0, 1), (0, 0, 1))):
            oracle.fill_halo_regions(h, xl, yl, sg, (size[0], size[1], f.Nz), (halo[0], halo[1], f.Hz))
            assert np.array_equal(f.data.cpu().numpy(), h), f.loc


def test_halo_fill_plan_graph_replay(osg, oracle, gpu):
    """HaloFillPlan.graph(repeat): `repeat` fills captured into one HIP graph; a replay applies them all
    (two fills of a -1 x-Face field flip the self-mapped cell of row Ny twice)"""
    size, halo = (64, 32, 1), (4, 4, 1)
    grid = osg.TripolarGrid(size=size, halo=halo)
    rng = np.random.default_rng(31)
    U = osg.Field((osg.Face, osg.Center, None), grid)
    eta = osg.Field((osg.Center, osg.Center, None), grid)
    hosts = []
    for f in (U, eta):
        h = rng.uniform(-1, 1, tuple(f.data.shape)); hosts.append(h)
    g = osg.halo_fill_plan((U, eta)).graph(repeat=2)
    for f, h in zip((U, eta), hosts):                       # (re)load the inputs after the capture's warm-up runs
        f.data.copy_(torch.from_numpy(h))
    g.replay()
    torch.cuda.synchronize()
    for f, h, (xl, yl, sg) in zip((U, eta), hosts, ((1, 0, -1), (0, 0, 1))):
        for _ in range(2):
            oracle.fill_halo_regions(h, xl, yl, sg, (size[0], size[1], 1), (halo[0], halo[1], 0))
        assert np.array_equal(f.data.cpu().numpy(), h), f.loc


def test_z_windowed_fields(osg, oracle, gpu):
    """Field(loc, grid; indices = (:, :, k1:k2)) (validate_indices, src/tripolar_grid_extensions.jl:58): the parent of a field windowed
    in z holds exactly those levels and no z halo, so the fill runs with Nz = k2 - k1 + 1, Hz = 0 -- e.g. the surface slice of u.
    Windows in x or y and levels outside the grid are refused."""
    size, halo = (64, 24, 6), (4, 4, 3)
    grid = osg.TripolarGrid(osg.GPU(0), torch.float64, size=size, halo=halo)
    full = slice(None)
    u_top = osg.Field((osg.Face, osg.Center, osg.Center), grid, indices=(full, full, 6))                 # k = Nz: one level
    c_win = osg.Field((osg.Center, osg.Center, osg.Center), grid, indices=(full, full, range(2, 5)))      # k = 2:4
    w_top = osg.Field((osg.Center, osg.Center, osg.Face), grid, indices=(full, full, 7))                  # z-Face: Nz + 1 levels exist
    assert (u_top.Nz, u_top.Hz, tuple(u_top.data.shape)) == (1, 0, (1, 32, 72))
    assert (c_win.Nz, c_win.Hz, tuple(c_win.data.shape)) == (3, 0, (3, 32, 72)) and w_top.Nz == 1
    rng = np.random.default_rng(8)
    hosts = []
    for f in (u_top, c_win, w_top):
        h = rng.uniform(-1, 1, tuple(f.data.shape))
        f.data.copy_(torch.from_numpy(h)); hosts.append(h)
    osg.fill_halo_regions([u_top, c_win, w_top])
    torch.cuda.synchronize()
    for f, h, (xl, yl, sg) in zip((u_top, c_win, w_top), hosts, ((1, 0, -1), (0, 0, 1), (0, 0, 1))):
        oracle.fill_halo_regions(h, xl, yl, sg, (64, 24, f.Nz), (4, 4, 0))
        assert np.array_equal(f.data.cpu().numpy(), h), f.loc
    c_win.set_(lambda x, y, z: z)                                      # z nodes of the window: levels 2..4 of the grid's centres
    assert torch.equal(c_win.interior()[:, 0, 0], grid.z_centers[3 + 1:3 + 4].to(torch.float64))
    with pytest.raises(NotImplementedError):
        osg.Field((osg.Center, osg.Center, osg.Center), grid, indices=(range(1, 9), full, full))
    with pytest.raises(ValueError):
        osg.Field((osg.Center, osg.Center, osg.Center), grid, indices=(full, full, range(5, 8)))
    with pytest.raises(ValueError):
        osg.Field((osg.Center, osg.Center, None), grid, indices=(full, full, 1))
