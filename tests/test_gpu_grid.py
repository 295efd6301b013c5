"""GPU parity of the TripolarGrid metric precompute (tpg_build_grid) against the oracle.
Tolerance stated by BASELINE.json north_star: <= 1e-12 relative on Float64 metrics.  By
construction (tests/../csrc/tpg_math.hpp) the device arithmetic is the same IEEE operation
sequence as the oracle's, so the expected outcome is bit-identity; both are asserted."""
import ctypes as C

import numpy as np
import pytest
import torch

from helpers import max_rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-12     # relative, Float64 metrics (north_star)
TOL32 = 0.0     # Float32 grids: Float64 pipeline rounded once at the end -> identical roundings


def compare(grid, ref, tol=TOL):
    worst, ndiff = 0.0, 0
    for name, r in ref.items():
        got = getattr(grid, name).cpu().numpy()
        assert got.shape == r.shape and got.dtype == r.dtype, name
        w, n = max_rel_err(got, r)
        worst, ndiff = max(worst, w), ndiff + n
    assert worst <= tol, f"max relative error {worst:.3e} over {ndiff} differing elements"
    return ndiff


CASES = [
    dict(size=(60, 30, 1)),                                                            # config 1 (README)
    dict(size=(4, 5, 1), first_pole_longitude=75, north_poles_latitude=35),            # runtests.jl:10-14 (fold reads row 1)
    dict(size=(10, 10, 1)),                                                            # zipper test grid (Nx % 4 != 0)
    dict(size=(360, 180, 1), first_pole_longitude=75, north_poles_latitude=35),        # test_tripolar_grid.jl:59
    dict(size=(62, 31, 2), halo=(3, 2, 1)),
    dict(size=(64, 33, 1), halo=(1, 1, 1), southernmost_latitude=-75.5, radius=1.0),   # non-integer south: dd range path
    dict(size=(128, 64, 1), halo=(7, 5, 2), north_poles_latitude=60, first_pole_longitude=-35.25),
    dict(size=(8, 4, 1), halo=(4, 4, 1)),                                              # Ny == Hy: fold reaches the zero south halo
    dict(size=(126, 40, 1), first_pole_longitude=1000.5),                              # |fpl+90| > 360: every step takes the general (coord) path
    dict(size=(62, 24, 1), halo=(2, 2, 1)),                                            # exactly one 62-column wave window
    dict(size=(250, 36, 1), north_poles_latitude=80.25, southernmost_latitude=-89),    # points within 1 degree of the south pole / poles near 90
]


@pytest.mark.parametrize("kw", CASES, ids=[str(c["size"]) for c in CASES])
def test_parity_f64(osg, oracle, gpu, kw):
    grid = osg.TripolarGrid(osg.GPU(0), torch.float64, **kw)
    ndiff = compare(grid, oracle.build_grid(dtype=np.float64, **kw))
    assert ndiff == 0, "Float64 arrays expected bit-identical to the oracle"


@pytest.mark.parametrize("kw", CASES[:5], ids=[str(c["size"]) for c in CASES[:5]])
def test_parity_f32(osg, oracle, gpu, kw):
    grid = osg.TripolarGrid(osg.GPU(0), torch.float32, **kw)
    assert grid.lambda_cc.dtype == torch.float32                                       # eltype(grid) == FT
    compare(grid, oracle.build_grid(dtype=np.float32, **kw), TOL32)


def test_config2_quarter_degree(osg, oracle, gpu):
    """BASELINE config 2: 1/4 degree (1440x720x1) Float64 metric precompute"""
    kw = dict(size=(1440, 720, 1))
    oracle.set_threads(min(16, oracle.max_threads()))
    ref = oracle.build_grid(**kw)
    oracle.set_threads(1)
    assert compare(osg.TripolarGrid(osg.GPU(0), **kw), ref) == 0


def test_tenth_degree_full_parity_and_properties(osg, oracle, gpu):
    """1/10 degree (3600x1800): full-array parity plus the size-independent properties"""
    size = (3600, 1800, 75)
    grid = osg.TripolarGrid(osg.GPU(0), size=size)
    oracle.set_threads(min(16, oracle.max_threads()))
    ref = oracle.build_grid(size)
    oracle.set_threads(1)
    assert compare(grid, ref) == 0
    Nx, Ny, _ = size
    for n in ("lambda_cc", "lambda_fc", "lambda_cf", "lambda_ff"):
        a = grid.interior(n)
        assert float(a.min()) >= 0 and float(a.max()) < 360
    assert bool((grid.interior("lambda_ff")[:, 0] == 250.0).all()) and bool((grid.interior("lambda_ff")[:, Nx // 2] == 70.0).all())
    assert float(grid.interior("phi_fc")[Ny - 1, 0]) == 55.0 and float(grid.interior("phi_fc")[Ny - 1, Nx // 2]) == 55.0
    for n in ("phi_cc", "dx_cc", "dy_cc", "az_cc"):                                    # fold symmetry of row Ny
        row = grid.interior(n)[Ny - 1]
        assert torch.equal(row, row.flip(0))
    for n in ("dx_cc", "dy_ff", "az_fc"):                                              # periodic x halos
        a = getattr(grid, n)
        assert torch.equal(a[:, :4], a[:, Nx:Nx + 4]) and torch.equal(a[:, Nx + 4:], a[:, 4:8])
    assert osg.x_domain(grid) == (0, 360)
    assert osg.y_domain(grid) == (float(ref["phi_ff"].min()), 90)


def test_twentyfourth_degree_parity(osg, oracle, gpu):
    """1/24 degree (8640x4320, BASELINE config 5's grid): 37 M cells, 6 GB of Float64 arrays"""
    size = (8640, 4320, 1)
    grid = osg.TripolarGrid(osg.GPU(0), size=size)
    oracle.set_threads(min(16, oracle.max_threads()))
    ref = oracle.build_grid(size)
    oracle.set_threads(1)
    assert compare(grid, ref) == 0


def test_reference_readme_numbers_on_gpu(osg, gpu, kats):
    """the reference's only numeric known answers, reproduced by the HIP path itself"""
    k = kats["readme_60x30"]
    size = tuple(k["size"])
    g = osg.TripolarGrid(size=size)
    R = osg.R_Earth
    sig = lambda x: float(f"{float(x):.6g}")
    Nx, Ny, _ = size
    assert float(g.interior("lambda_ff")[Ny // 2, Nx // 2]) == k["center_lambda_phi"][0]
    assert round(float(g.interior("phi_ff")[Ny // 2, Nx // 2]), 4) == k["center_lambda_phi"][1]
    assert sig(np.rad2deg(float(g.interior("dx_cf")[14].sum())) / R) == k["longitude_extent_deg"]
    assert sig(np.rad2deg(float(g.interior("dx_ff").min())) / R) == k["min_dlambda"]
    assert sig(np.rad2deg(float(g.interior("dx_ff").max())) / R) == k["max_dlambda"]
    assert sig(np.rad2deg(float(g.interior("dy_fc")[:, 15].sum())) / R) == k["latitude_extent_deg"]
    assert sig(np.rad2deg(float(g.interior("dy_ff").min())) / R) == k["min_dphi"]
    assert sig(np.rad2deg(float(g.interior("dy_ff").max())) / R) == k["max_dphi"]


def test_golden_restatement_vectors(osg, gpu):
    import os
    from conftest import GOLDEN
    gold = np.load(os.path.join(GOLDEN, "restatement_60x30_f64.npz"))
    g = osg.TripolarGrid(size=(60, 30, 1))
    for name in gold.files:
        assert np.array_equal(getattr(g, name).cpu().numpy(), gold[name], equal_nan=True), name


@pytest.mark.parametrize("R", [2, 3, 8])
def test_latitude_bands_equal_slices_of_the_global_grid(osg, oracle, gpu, R):
    """config 4 partitioning (distributed_tripolar_grid.jl:41-49,112-120): every rank's band,
    built independently on the device, equals rows jstart-Hy:jend+Hy of the global grid."""
    size, halo = (96, 48, 1), (4, 4, 4)
    glob = oracle.build_grid(size, halo=halo)
    for r in range(R):
        arch = osg.Distributed(osg.GPU(0), osg.Partition(y=R), local_rank=r)
        g = osg.TripolarGrid(arch, size=size, halo=halo)
        jstart, jend = osg.local_row_range(size[1], arch)
        assert g.Ny == jend - jstart + 1 and g.jrange == (jstart, jend)
        assert g.topology[1] is (osg.RightConnected if r == 0 else osg.FullyConnected)     # :75
        for name, ref in glob.items():
            got = getattr(g, name).cpu().numpy()
            assert np.array_equal(got, ref[jstart - 1:jend + 2 * halo[1]], equal_nan=True), (r, name)


@pytest.mark.parametrize("size,halo,R", [((106, 12, 1), (4, 6, 1), 9), ((276, 4, 1), (3, 2, 1), 4), ((150, 8, 1), (4, 6, 1), 2)],
                         ids=["12rows-9ranks-Hy6", "4rows-4ranks-Hy2", "8rows-2ranks-Hy6"])
def test_bands_thinner_than_the_halo(osg, oracle, gpu, size, halo, R):
    """A band whose halo reaches past its neighbour sees row Ny (with its substitution) and north fold rows even
    though it is not the north rank: the reference slices these out of the global padded arrays
    (distributed_tripolar_grid.jl:47-49,112-120); found by tests/soak/soak_grid.py."""
    glob = oracle.build_grid(size, halo=halo)
    for r in range(R):
        arch = osg.Distributed(osg.GPU(0), osg.Partition(y=R), local_rank=r)
        g = osg.TripolarGrid(arch, size=size, halo=halo)
        jstart, jend = g.jrange
        for name, ref in glob.items():
            assert np.array_equal(getattr(g, name).cpu().numpy(), ref[jstart - 1:jend + 2 * halo[1]], equal_nan=True), (r, name)


def test_with_halo_and_reconstruct(osg, oracle, gpu):
    g = osg.TripolarGrid(size=(60, 30, 1))
    g2 = osg.with_halo((2, 3, 1), g)                                                   # with_halo.jl:5-23
    assert g2.halo_size == (2, 3, 1) and g2.size == g.size
    ref = oracle.build_grid((60, 30, 1), halo=(2, 3, 1))
    assert compare(g2, ref) == 0
    arch = osg.Distributed(osg.GPU(0), osg.Partition(y=2), local_rank=1)
    band = osg.TripolarGrid(arch, size=(60, 30, 1))
    full = osg.reconstruct_global_grid(band)                                           # distributed_tripolar_grid.jl:201-226
    assert full.size == (60, 30, 1) and compare(full, oracle.build_grid((60, 30, 1))) == 0


def test_conformal_mapping_kept_verbatim(osg, gpu):
    g = osg.TripolarGrid(size=(4, 5, 1), z=(0, 1), first_pole_longitude=75, north_poles_latitude=35,
                         southernmost_latitude=-80)
    assert g.Nx == 4 and g.Ny == 5 and g.Nz == 1                                       # runtests.jl:19-21
    cm = g.conformal_mapping
    assert cm.first_pole_longitude == 75 and cm.north_poles_latitude == 35 and cm.southernmost_latitude == -80
    assert isinstance(cm.first_pole_longitude, int)


def test_workspace_too_small_is_an_error(osg, gpu):
    lib = osg._lib.lib()
    p = osg._lib.TpgParams(60, 30, 1, 4, 4, 4, -80.0, 55.0, 70.0, 6371e3, 1, 1, 30, 0)
    arrs = [torch.empty((38, 68), dtype=torch.float64, device=gpu) for _ in range(20)]
    ws = torch.empty(64, dtype=torch.uint8, device=gpu)
    rc = lib.tpg_build_grid(C.byref(p), osg._lib.ptr_table(arrs), ws.data_ptr(), ws.numel(), None)
    assert rc == -4


def test_randomised_parameters_against_the_oracle(osg, oracle, gpu):
    """60 random (size, halo, poles, south, radius, dtype, band) configurations, bit-exact vs the oracle:
    exercises the index maps on shapes no fixed case covers (Nx = 2, halo == size, bands of one row ...)."""
    rng = np.random.default_rng(20261004)
    for trial in range(60):
        Nx = int(rng.choice([2, 4, 6, 8, 10, 12, 16, 30, 62, 64, 66, 124, 126, 130, 200]))
        Ny = int(rng.integers(2, 41))
        Hx = int(rng.integers(1, min(Nx, 6) + 1))
        Hy = int(rng.integers(1, min(Ny, 6) + 1))
        kw = dict(size=(Nx, Ny, 1), halo=(Hx, Hy, 1),
                  north_poles_latitude=float(rng.choice([35, 55, 60.5, 75, 85])),
                  first_pole_longitude=float(rng.choice([-180, -35.25, 0, 70, 90, 200, 359.5])),
                  southernmost_latitude=float(rng.choice([-85, -80, -77.7, -60, 0, 20])),
                  radius=float(rng.choice([1.0, 6371e3])))
        dtype = np.float64 if trial % 4 else np.float32
        tdt = torch.float64 if dtype == np.float64 else torch.float32
        ref = oracle.build_grid(dtype=dtype, **kw)
        g = osg.TripolarGrid(osg.GPU(0), tdt, **kw)
        for name, r in ref.items():
            assert np.array_equal(getattr(g, name).cpu().numpy(), r, equal_nan=True), (trial, kw, name)
        if Ny >= 4 and trial % 3 == 0:              # a random latitude band of the same grid
            R = int(rng.integers(2, min(Ny, 5) + 1))
            r = int(rng.integers(0, R))
            arch = osg.Distributed(osg.GPU(0), osg.Partition(y=R), local_rank=r)
            band = osg.TripolarGrid(arch, tdt, **kw)
            j0, j1 = band.jrange
            for name, rr in ref.items():
                assert np.array_equal(getattr(band, name).cpu().numpy(), rr[j0 - 1:j1 + 2 * Hy], equal_nan=True), (trial, kw, r, R, name)


def test_orthogonality_of_the_device_built_grid(osg, gpu):
    """test/test_tripolar_grid.jl:36-76 on the HIP-built 1 degree grid (poles 75E / 35N): the chord
    angle at every FF node (5-degree pole boxes and phi < -78 masked, as the reference's immersed
    mask does) deviates from 90 degrees by less than 2 degrees.  The reference bounds it by a
    cubed-sphere panel's range; that grid generator is Oceananigans-internal (see test_oracle_kat)."""
    Nx, Ny, H = 360, 180, 4
    g = osg.TripolarGrid(size=(Nx, Ny, 1), first_pole_longitude=75, north_poles_latitude=35)
    lam, phi = torch.deg2rad(g.lambda_ff), torch.deg2rad(g.phi_ff)
    xyz = [torch.cos(lam) * torch.cos(phi), torch.sin(lam) * torch.cos(phi), torch.sin(phi)]
    sl = lambda a, di, dj: a[H + dj:H + Ny - 1 + dj, H + di:H + Nx - 1 + di]
    v1 = torch.stack([sl(a, 1, 0) - sl(a, 0, 0) for a in xyz])
    v2 = torch.stack([sl(a, 0, 1) - sl(a, 0, 0) for a in xyz])
    cos = (v1 * v2).sum(0) / torch.sqrt((v1 * v1).sum(0) * (v2 * v2).sum(0))
    ang = torch.rad2deg(torch.acos(cos)) - 90
    L, P = sl(g.lambda_ff, 0, 0), sl(g.phi_ff, 0, 0)
    mask = ((abs(L - 75) < 5) & (abs(35 - P) < 5)) | ((abs(L - 255) < 5) & (abs(35 - P) < 5)) | (P < -78)
    ang = torch.where(mask, torch.zeros_like(ang), ang)
    assert float(ang.max()) < 2.0 and float(ang.min()) > -2.0


def test_arrays_beyond_4_GiB_take_the_64_bit_offsets(osg, gpu):
    """The tile kernel addresses its stores with 32-bit byte offsets (saddr form), valid while ONE array is below 4 GiB; the launcher
    must send larger arrays to the thread-per-cell kernel with 64-bit offsets.  16384 x 32768 Float64: 4.30 GB per array, 86 GB for
    the 20.  Checked against latitude BANDS of the same grid built by the tile kernel (their arrays are small; band builds are
    bit-exact against the oracle in test_latitude_bands_equal_slices_of_the_global_grid): rows at the south end, across the 4 GiB
    mark, and at the north fold."""
    size, halo = (16384, 32768, 1), (4, 4, 1)
    Nx, Ny = size[0], size[1]
    assert (Nx + 8) * (Ny + 8) * 8 >= 1 << 32
    free, _ = torch.cuda.mem_get_info(gpu)
    if free < 100e9:
        pytest.skip("needs 100 GB of free HBM")
    big = osg.TripolarGrid(osg.GPU(0), torch.float64, size=size, halo=halo)
    R = 512                                                            # 64 rows per band
    row_4gib = (1 << 32) // ((Nx + 8) * 8) - 4 + 1                     # the global row whose cells straddle byte offset 2^32: 32749
    assert 64 * (R - 1) < row_4gib <= Ny                               # ... lies in the northernmost band, below the fold rows
    for r in (0, R // 2 - 1, R - 1):
        band = osg.TripolarGrid(osg.Distributed(osg.GPU(0), osg.Partition(y=R), local_rank=r), torch.float64, size=size, halo=halo)
        j0, j1 = band.jrange
        assert (j0, j1) == (1 + 64 * r, 64 * (r + 1))
        for name in osg._lib.ARRAY_NAMES:
            assert torch.equal(getattr(band, name), getattr(big, name)[j0 - 1:j1 + 2 * halo[1]]), (r, name)
    del big
    torch.cuda.empty_cache()


def test_tables_valid_flag_skips_the_table_kernel_and_changes_nothing(osg, gpu):
    """tpg_params.reserved = TPG_BUILD_TABLES_VALID: the workspace holds the 1-D tables of an earlier build of the same (Nx, Ny, Hy, ft,
    latitudes, radius); bands, Hx and first_pole_longitude may differ.  Every array must equal the build without the flag bit for bit --
    here 3 bands of a 360 x 180 grid built after ONE table pass, against their own full builds, f64 and f32; a poisoned workspace shows the
    flag really skips the kernel (garbage tables -> garbage grid)."""
    import ctypes as C
    lib, L = osg._lib.lib(), osg._lib
    Nx, Ny, H = 360, 180, 4
    for ft, tdt in ((L.TPG_F64, torch.float64), (L.TPG_F32, torch.float32)):
        def params(jstart, jend, flags, Hx=H, fpl=70.0):
            return L.TpgParams(Nx, Ny, 3, Hx, H, H, -80.0, 55.0, fpl, osg.R_Earth, ft, jstart, jend, flags)
        p0 = params(1, Ny, 0)
        ws = torch.zeros(int(lib.tpg_build_grid_workspace_bytes(C.byref(p0))), dtype=torch.uint8, device=gpu)
        def build(p, Hx=H):
            rows = p.jend - p.jstart + 1 + 2 * H
            out = [torch.full((rows, Nx + 2 * Hx), float("nan"), dtype=tdt, device=gpu) for _ in L.ARRAY_NAMES]
            L.check(lib.tpg_build_grid(C.byref(p), L.ptr_table(out), ws.data_ptr(), ws.numel(), None))
            torch.cuda.synchronize()
            return out
        for (j0, j1, Hx, fpl) in ((1, 60, 4, 70.0), (61, 120, 2, 75.0), (121, 180, 4, 70.0)):
            want = build(params(j0, j1, 0, Hx, fpl), Hx)                    # fills the tables
            got = build(params(j0, j1, 1, Hx, fpl), Hx)                     # reuses them
            for name, a, b in zip(L.ARRAY_NAMES, got, want):
                assert torch.equal(a.view(torch.int32), b.view(torch.int32)), (name, j0, ft)
        ws.fill_(0xFF)                                                       # NaN tables: the flag must not recompute them
        bad = build(params(1, 60, 1))
        assert bool(torch.isnan(bad[L.ARRAY_NAMES.index("lambda_cc")][H:-H, H:-H]).any())
        good = build(params(1, 60, 0))
        assert not bool(torch.isnan(good[L.ARRAY_NAMES.index("lambda_cc")][H:-H, H:-H]).any())
    p = params(1, Ny, 2)
    assert lib.tpg_build_grid(C.byref(p), L.ptr_table(good), ws.data_ptr(), ws.numel(), None) == -1 and b"unknown flag" in lib.tpg_last_error()


def test_grid_hosts_reuse_the_tables_and_a_changed_key_recomputes(osg, oracle, gpu):
    """The callers of TPG_BUILD_TABLES_VALID (include/tripolar_hip.h): the grid keeps its table workspace, and with_halo (same size, new
    Hx / Hz: src/with_halo.jl:5-44), reconstruct_global_grid after a band build (src/distributed_tripolar_grid.jl:201-226) and consecutive
    band builds of one geometry build with the flag -- bit-identical to fresh builds (the oracle); a changed Hy, size, latitude, radius or
    element type changes the key and recomputes.  A poisoned workspace proves which builds really skipped the table kernel."""
    import gc
    from orthogonalsphericalshellgrids.jl_amd import grids
    size = (120, 60, 2)                                                       # a geometry no other test holds alive
    g = osg.TripolarGrid(size=size)
    assert not g.tables_reused and g.workspace.key == grids.table_key(120, 60, 4, torch.float64, -80, 55, osg.R_Earth, g.device)
    g2 = osg.with_halo((2, 4, 1), g)                                          # new Hx, Hz: same tables
    assert g2.tables_reused and g2.workspace is g.workspace and compare(g2, oracle.build_grid(size, halo=(2, 4, 1))) == 0
    g3 = osg.with_halo((4, 3, 4), g)                                          # new Hy: the south-row table has another length -> recomputed
    assert not g3.tables_reused and g3.workspace is not g.workspace and compare(g3, oracle.build_grid(size, halo=(4, 3, 4))) == 0
    for kw in (dict(southernmost_latitude=-75), dict(north_poles_latitude=60), dict(radius=1.0)):
        gk = osg.TripolarGrid(size=size, **kw)
        assert not gk.tables_reused and compare(gk, oracle.build_grid(size, **kw)) == 0, kw
    g32 = osg.TripolarGrid(None, torch.float32, size=size)
    assert not g32.tables_reused
    # a plain second build of the same geometry does NOT look at which other grids are alive: sharing is explicit
    again = osg.TripolarGrid(size=size)
    assert not again.tables_reused and again.workspace is not g.workspace
    del again
    glob = oracle.build_grid(size)
    bands = []
    with osg.share_tables():                                                  # ... inside this scope any live grid's tables of the key are taken
        gfpl = osg.TripolarGrid(size=size, first_pole_longitude=75)          # first_pole_longitude does not enter the tables
        assert gfpl.tables_reused and compare(gfpl, oracle.build_grid(size, first_pole_longitude=75)) == 0
        # consecutive band builds of one geometry (an emulated chain in one process)
        for r in range(3):
            arch = osg.Distributed(osg.GPU(0), osg.Partition(y=3), local_rank=r)
            b = osg.TripolarGrid(arch, size=size)
            assert b.tables_reused                                            # a live grid of this geometry exists: every band finds tables
            j0, j1 = b.jrange
            for name, ref in glob.items():
                assert np.array_equal(getattr(b, name).cpu().numpy(), ref[j0 - 1:j1 + 8], equal_nan=True), (r, name)
            bands.append(b)
    lone = osg.TripolarGrid(osg.Distributed(osg.GPU(0), osg.Partition(y=3), local_rank=1), size=size)
    assert not lone.tables_reused                                             # outside the scope: its own tables
    full = osg.reconstruct_global_grid(bands[-1])                             # explicit hand-over of the band's workspace: always reused
    assert full.tables_reused and compare(full, glob) == 0
    del lone
    # the flag really skips the kernel: poison the shared tables, a reusing build goes wrong, a build with another key does not
    torch.cuda.synchronize()
    g.workspace.tensor.fill_(0xFF)
    bad = osg.with_halo((3, 4, 2), g)
    assert bad.tables_reused and bool(torch.isnan(bad.interior("lambda_cc")).any())
    ok = osg.with_halo((3, 5, 2), g)
    assert not ok.tables_reused and compare(ok, oracle.build_grid(size, halo=(3, 5, 2))) == 0
    # a workspace lives as long as a grid holds it: drop them all and the next build computes its own tables again
    del g, g2, bad, bands, b, full, gfpl
    gc.collect()
    with osg.share_tables():                                                  # (even where sharing is allowed: nothing of this key is alive)
        fresh = osg.TripolarGrid(size=size)
    assert not fresh.tables_reused and compare(fresh, glob) == 0
    # the tables may have been written on another stream: the reusing build waits for them
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        ga = osg.TripolarGrid(size=(240, 120, 1))
    gb = osg.with_halo((2, 4, 1), ga)                                         # current stream != side
    assert gb.tables_reused and compare(gb, oracle.build_grid((240, 120, 1), halo=(2, 4, 1))) == 0
