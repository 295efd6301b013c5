"""BASELINE config 5 -- 1/24 degree (8640 x 4320 x 100): the halo fills of one baroclinic step of the kind of
model test/runtests.jl:46-77 and examples/bickley_jet.jl:44-55 build (HydrostaticFreeSurfaceModel with a
SplitExplicitFreeSurface on a TripolarGrid), through the product API and against the oracle.

Per step such a model fills
  * its 3-D prognostic fields in one tupled fill_halo_regions!((u, v, T, S, c)) -- 5 x 32.3 GB of Float64 here;
  * every barotropic sub-step the 2-D fields eta, U, V, which live on a grid whose NORTH halo is extended
    (test/runtests.jl:61-71: Hy == length(averaging_weights) + 1; kernels run over 1:Ny+Hy-1).

Checking a 32 GB field against the oracle element by element would need 32 GB of host memory per field, so the 3-D test
follows test_config3: bit-exact oracle parity on every row the fold or a corner can touch (rows Ny-Hy .. Ny+Hy, all
levels incl. the z halos, x halos included), run by the oracle as a short field; below that slab the periodic pass
is checked on the device (x halos == the wrapped interior columns) and the interior by a checksum of the raw bits.
The 2-D fields are compared whole.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

@pytest.fixture(autouse=True)
def _room_for_160_gb():
    """these tests hold five 32 GB fields: whatever earlier tests left in torch's caching allocator (or in uncollected cycles) goes first"""
    import gc
    gc.collect()
    torch.cuda.empty_cache()
    yield
    gc.collect()
    torch.cuda.empty_cache()


SIZE = (8640, 4320, 100)
HALO = (4, 4, 4)
SUBSTEPS = 30


def _synthetic(osg, field, seed):
    from tools import testlib
    lib = testlib.lib()
    rc = lib.tpg_fill_synthetic(field.data.data_ptr(), seed, 12345.0, field.Nx, field.Ny, field.Nz, field.Hx, field.Hy, field.Hz,
                                osg._lib.ft_of(field.data.dtype), None)
    assert rc == 0


def _check_big_field(oracle, f, before_top, low_sum, xl, yl, sg):
    Nx, Ny, Nz, Hx, Hy, Hz = f.Nx, f.Ny, f.Nz, f.Hx, f.Hy, f.Hz
    d = f.data
    top = slice(Ny - 1, Ny + 2 * Hy)                     # parent rows of logical rows Ny-Hy .. Ny+Hy
    # (1) the slab as a short (Ny' = Hy+1) field: zipper on k = 1..Nz, then periodic x on every level
    want = np.concatenate([np.zeros_like(before_top[:, :Hy]), before_top], axis=1)
    oracle.fill_halo_regions(want, xl, yl, sg, (Nx, Hy + 1, Nz), (Hx, Hy, Hz))
    got = d[:, top].cpu().numpy()
    assert np.array_equal(got, want[:, Hy:]), (xl, yl, sg)
    # (2) below the slab: interior bits untouched, x halos are the wrapped interior columns
    ints = d.view(torch.int64)
    assert int(ints[:, :Ny - 1, Hx:Hx + Nx].sum()) == low_sum
    assert torch.equal(d[:, :Ny - 1, :Hx], d[:, :Ny - 1, Nx:Nx + Hx])
    assert torch.equal(d[:, :Ny - 1, Nx + Hx:], d[:, :Ny - 1, Hx:2 * Hx])
    assert not bool((d[Hz:Hz + Nz, Hy:Ny - 1, :Hx] == 12345.0).any())   # sentinel gone from the x halos of interior rows and levels


@pytest.mark.parametrize("HALO", [(4, 4, 4), (5, 5, 5)], ids=["halo4", "halo5"])
def test_config5_tupled_3d_fill(osg, oracle, gpu, HALO):
    """fill_halo_regions!((u, v, T, S, c)) at 8640 x 4320 x 100, Float64: ONE merged launch over 162 GB of fields on one MI355X --
    at halo 4 and at halo (5, 5, 5), the halo the reference's own model examples run (examples/bickley_jet.jl:21,
    examples/distributed_bickley_jet.jl:23; 165 GB)"""
    grid = osg.TripolarGrid(size=SIZE, halo=HALO)
    Nx, Ny, Nz = SIZE
    Hx, Hy, Hz = HALO
    u, v = osg.XFaceField(grid), osg.YFaceField(grid)
    T, S, c = osg.CenterField(grid), osg.CenterField(grid), osg.CenterField(grid)
    fields = (u, v, T, S, c)
    specs = ((1, 0, -1), (0, 1, -1), (0, 0, 1), (0, 0, 1), (0, 0, 1))
    for k, (f, (xl, yl, sg)) in enumerate(zip(fields, specs)):
        assert f.boundary_conditions.north.condition == sg          # default zipper sign by location (tripolar_grid_extensions.jl:49-53)
        _synthetic(osg, f, 0xC5 + k)
    top = slice(Ny - 1, Ny + 2 * Hy)
    before = [f.data[:, top].cpu().numpy() for f in fields]
    sums = [int(f.data.view(torch.int64)[:, :Ny - 1, Hx:Hx + Nx].sum()) for f in fields]
    osg.fill_halo_regions(fields)
    torch.cuda.synchronize()
    for f, b, s, (xl, yl, sg) in zip(fields, before, sums, specs):
        _check_big_field(oracle, f, b, s, xl, yl, sg)
    # a second fill through a reusable plan: the folded halos are a fixed point except for the self-mapped
    # x-Face cell of row Ny, which flips with every fill of a -1 field (App. C-5); run the oracle again on the slab
    before = [f.data[:, top].cpu().numpy() for f in fields]
    plan = osg.halo_fill_plan(fields)
    plan()
    torch.cuda.synchronize()
    for f, b, s, (xl, yl, sg) in zip(fields, before, sums, specs):
        _check_big_field(oracle, f, b, s, xl, yl, sg)


def test_config5_w_field_has_one_more_level(osg, oracle, gpu):
    """w lives at (Center, Center, Face): Nz + 1 = 101 levels are folded (the per-field level count comes from the
    field's location, fields.py), checked on a thin-in-y grid of the same 8640-wide rows"""
    size = (SIZE[0], 24, SIZE[2])
    grid = osg.TripolarGrid(size=size, halo=HALO)
    w = osg.ZFaceField(grid)
    assert w.Nz == SIZE[2] + 1
    h = np.random.default_rng(55).uniform(-1, 1, tuple(w.data.shape))
    w.data.copy_(torch.from_numpy(h))
    osg.fill_halo_regions(w)
    oracle.fill_halo_regions(h, 0, 0, 1, (size[0], size[1], w.Nz), HALO)
    assert np.array_equal(w.data.cpu().numpy(), h)


@pytest.mark.parametrize("HALO", [(4, 4, 4), (5, 5, 5)], ids=["halo4", "halo5"])
@pytest.mark.parametrize("mode", ["eager", "plan", "graph"])
def test_config5_barotropic_substep_fills(osg, oracle, gpu, mode, HALO):
    """eta, U, V of the split-explicit free surface at 8640 x 4320 with the extended north halo
    (Hy = substeps + 1 = 31 for the 30 sub-steps of examples/bickley_jet.jl:44): SUBSTEPS consecutive fills
    issued eagerly, through a reusable plan, and replayed from one captured HIP graph, whole arrays against the oracle."""
    Hy_ext = SUBSTEPS + 1
    grid = osg.TripolarGrid(size=(SIZE[0], SIZE[1], 1), halo=HALO)
    ext = osg.with_halo((HALO[0], Hy_ext, HALO[2]), grid)             # with_halo: src/with_halo.jl:5-23
    assert ext.halo_size == (HALO[0], Hy_ext, HALO[2]) and ext.size == grid.size
    eta = osg.Field((osg.Center, osg.Center, None), ext)
    U = osg.Field((osg.Face, osg.Center, None), ext)
    V = osg.Field((osg.Center, osg.Face, None), ext)
    fields = (eta, U, V)
    specs = ((0, 0, 1), (1, 0, -1), (0, 1, -1))
    rng = np.random.default_rng(2024)
    hosts = [rng.uniform(-1, 1, tuple(f.data.shape)) for f in fields]
    nfills = 3 if mode != "graph" else SUBSTEPS
    if mode == "graph":
        g = osg.halo_fill_plan(fields).graph(repeat=SUBSTEPS)
    for f, h in zip(fields, hosts):
        f.data.copy_(torch.from_numpy(h))
    if mode == "eager":
        for _ in range(nfills):
            osg.fill_halo_regions(fields)
    elif mode == "plan":
        plan = osg.halo_fill_plan(fields)
        for _ in range(nfills):
            plan()
    else:
        g.replay()
    torch.cuda.synchronize()
    size2, halo2 = (SIZE[0], SIZE[1], 1), (HALO[0], Hy_ext, 0)
    for f, h, (xl, yl, sg) in zip(fields, hosts, specs):
        for _ in range(nfills):
            oracle.fill_halo_regions(h, xl, yl, sg, size2, halo2)
        assert np.array_equal(f.data.cpu().numpy(), h), (mode, f.loc)
