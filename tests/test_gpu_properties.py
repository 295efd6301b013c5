"""Size-independent properties of fill_halo_regions! at BASELINE's FULL sizes, with no oracle in the loop: what the fill IS -- a signed
permutation copy of interior cells into halo cells (src/zipper_boundary_condition.jl:70-138, then periodic x) -- implies, bit for bit,

  * linearity     fill(X + Y) == fill(X) + fill(Y) on every cell of the parent (written cells: s (x + y) = s x + s y exactly; others: untouched);
  * idempotence   a second fill changes nothing (every source is an interior cell the fill does not write -- except the one x-Face pivot
                  cell i = Nx/2 + 1 of row Ny, which maps to itself with the field's sign: zeroed here, as a velocity on the fold is);
  * antisymmetry  the same data filled with the opposite sign differs by exactly a factor -1 on the cells the fold writes (except where the
                  reference takes |sign|: the wrap i = 1 of x-Face fields) and nowhere else;
  * permutation   every north halo row holds the same multiset of magnitudes as its source row.

  * fixed point   the 20 arrays tpg_build_grid writes are halo-filled by ITS kernels (k_halos: a4 / a11 of SURVEY.md 8) -- and must therefore be
                  left bit-identical, on the whole parent, by the fill kernels applied to them as fields of their location with sign +1:
                  two kernel families with separately written index maps agreeing on every halo cell.

  * halo window   what a logical cell (i, j) of a grid array holds does not depend on the halo width: the (4, 4, 4) grid is, bit for bit, the
                  inner window of the (5, 5, 5) grid, and the (2, 3, 1) grid the inner window of that -- interior, fold rows, periodic
                  columns and the continuation rows alike (three launches with three table layouts and three tile grids).

The kernels under test are the ones the bench step and config 5 launch (k_fill_merged, plain and GEN); parity against the oracle at these
sizes is tests/test_gpu_zipper.py / test_gpu_config5.py -- this file is the independent cross-check that needs no second implementation."""
import ctypes as C
import gc

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
SPECS = [(0, 0, 1), (1, 0, -1), (0, 1, -1), (1, 1, 1)]          # c, u, v, zeta: (xloc, yloc, sign)


def _fill(osg, fields, xl, yl, sg, size, halo, ft):
    n = len(fields)
    rc = osg._lib.lib().tpg_fill_halo_regions(osg._lib.ptr_table(fields), n, (C.c_int8 * n)(*[xl] * n), (C.c_int8 * n)(*[yl] * n),
                                              (C.c_int32 * n)(*sg), *size, *halo, 1, ft, None)
    assert rc == 0, osg._lib.lib().tpg_last_error()
    torch.cuda.synchronize()


def _synthetic(tlib, shape, tdt, ft, seed, sentinel, size, halo, gpu, xl, yl):
    d = torch.empty(shape, dtype=tdt, device=gpu)
    assert tlib.tpg_fill_synthetic(d.data_ptr(), seed, sentinel, *size, *halo, ft, None) == 0
    (Nx, Ny, _), (Hx, Hy, _) = size, halo
    if xl == 1 and yl == 0:
        d[:, Hy + Ny - 1, Hx + Nx // 2] = 0            # the pivot cell (i = Nx/2 + 1, j = Ny): maps to itself with the field's sign
    return d


@pytest.mark.parametrize("h", [4, 5], ids=["halo4", "halo5"])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_config3_fill_properties(osg, gpu, tlib, dtype, h):
    size, halo = (3600, 1800, 75), (h, h, h)
    (Nx, Ny, Nz), (Hx, Hy, Hz) = size, halo
    tdt, ft = (torch.float64, 1) if dtype == np.float64 else (torch.float32, 0)
    shape = (Nz + 2 * Hz, Ny + 2 * Hy, Nx + 2 * Hx)
    lev, north = slice(Hz, Hz + Nz), slice(Hy + Ny, None)
    for fid, (xl, yl, sg) in enumerate(SPECS):
        X = _synthetic(tlib, shape, tdt, ft, 0xA11CE + fid, 12345.0, size, halo, gpu, xl, yl)
        Y = _synthetic(tlib, shape, tdt, ft, 0xB0B + fid, -321.5, size, halo, gpu, xl, yl)
        X0 = X.clone()
        Z = X + Y
        _fill(osg, [X, Y, Z], xl, yl, [sg] * 3, size, halo, ft)                       # one launch, three fields of one location
        assert torch.equal(Z, X + Y), ("linearity", xl, yl)
        rows = slice(Hy, Hy + Ny)
        assert not bool((X[lev, north] == 12345.0).any()) and not bool((X[lev, rows, :Hx] == 12345.0).any()) \
            and not bool((X[lev, rows, Hx + Nx:] == 12345.0).any())                                          # the halos were written
        _fill(osg, [Z], xl, yl, [sg], size, halo, ft)
        assert torch.equal(Z, X + Y), ("idempotence", xl, yl)
        del Y, Z
        # the opposite sign: -1 times the result on the fold's cells, identical elsewhere
        W = X0
        _fill(osg, [W], xl, yl, [-sg], size, halo, ft)
        flipped = W == -X
        same = W == X
        assert bool((flipped | same).all()), ("antisymmetry: some cell is neither", xl, yl)
        keep = torch.ones(shape[2], dtype=torch.bool, device=gpu)
        if xl == 1:
            keep[Hx] = False; keep[Hx + Nx] = False                                    # i = 1 (and its periodic image): |sign| there
        assert bool(flipped[lev, north][:, :, keep].all()), ("antisymmetry: north halo rows", xl, yl)
        assert torch.equal(W[:, :Hy + Ny - 1], X[:, :Hy + Ny - 1]) and torch.equal(W[:Hz, north], X[:Hz, north])   # below row Ny, z-halo levels: no fold
        if yl == 0:                                                                   # row Ny: the substituted half flips, the other half does not
            row = Hy + Ny - 1
            assert torch.equal(W[lev, row, Hx:Hx + Nx // 2], X[lev, row, Hx:Hx + Nx // 2])
            assert bool(flipped[lev, row, Hx + Nx // 2 + 1:Hx + Nx].all())
        # permutation: halo row Ny + j holds the magnitudes of its source row (y-Center: Ny - j, y-Face: Ny - j + 1)
        for j in range(1, Hy + 1):
            dst = X[lev, Hy + Ny - 1 + j, Hx:Hx + Nx].abs().sort(dim=-1).values
            src = X[lev, Hy + Ny - 1 - j + (1 if yl == 1 else 0), Hx:Hx + Nx].abs().sort(dim=-1).values
            assert torch.equal(dst, src), ("permutation", xl, yl, j)
        del X, X0, W, flipped, same
        gc.collect(); torch.cuda.empty_cache()


def test_config5_fill_is_linear_and_idempotent_at_the_callers_halo(osg, gpu, tlib):
    """BASELINE config 5's caller geometry, 8640 x 4320 x 100 at the reference's model halo (5, 5, 5) (examples/bickley_jet.jl:21): three 32 GB
    fields X, Y, X + Y of the u location (x-Face, y-Center, sign -1: the wrap, the pivot and the row-Ny substitution all live there) through ONE
    launch of the GEN merged kernel; linearity and idempotence on all 3 x 4.1e9 cells, compared on the device."""
    gc.collect(); torch.cuda.empty_cache()
    size, halo = (8640, 4320, 100), (5, 5, 5)
    (Nx, Ny, Nz), (Hx, Hy, Hz) = size, halo
    shape = (Nz + 2 * Hz, Ny + 2 * Hy, Nx + 2 * Hx)
    xl, yl, sg = 1, 0, -1
    X = _synthetic(tlib, shape, torch.float64, 1, 0xC5 + 1, 12345.0, size, halo, gpu, xl, yl)
    Y = _synthetic(tlib, shape, torch.float64, 1, 0xC5 + 2, -321.5, size, halo, gpu, xl, yl)
    Z = X + Y
    _fill(osg, [X, Y, Z], xl, yl, [sg] * 3, size, halo, 1)
    assert torch.equal(Z, X + Y)
    lev, rows = slice(Hz, Hz + Nz), slice(Hy, Hy + Ny)
    assert not bool((X[lev, Hy + Ny:] == 12345.0).any()) and not bool((X[lev, rows, :Hx] == 12345.0).any()) and not bool((X[lev, rows, Hx + Nx:] == 12345.0).any())
    _fill(osg, [X, Y, Z], xl, yl, [sg] * 3, size, halo, 1)
    assert torch.equal(Z, X + Y)
    del X, Y, Z
    gc.collect(); torch.cuda.empty_cache()


@pytest.mark.parametrize("size,halo,tdt", [((3600, 1800, 1), (4, 4, 4), torch.float64), ((3600, 1800, 1), (5, 5, 5), torch.float64),
                                           ((3600, 1800, 1), (5, 5, 5), torch.float32), ((8640, 4320, 1), (4, 4, 4), torch.float64)],
                         ids=["tenth-halo4", "tenth-halo5", "tenth-halo5-f32", "twentyfourth-halo4"])
def test_a_built_grid_is_a_fixed_point_of_the_halo_fill(osg, gpu, size, halo, tdt):
    """src/tripolar_grid.jl:178-186 (coordinates) and :240-269 (metrics) fill the halos of the grid arrays with the very fill_halo_regions! of
    the fields, sign +1.  tpg_build_grid writes those halo cells with its own kernel; the fill kernels must find nothing to change."""
    gc.collect(); torch.cuda.empty_cache()
    lib = osg._lib.lib()
    (Nx, Ny, _), (Hx, Hy, _) = size, halo
    g = osg.TripolarGrid(None, tdt, size=size, halo=halo)
    ft = osg._lib.ft_of(tdt)
    for name in osg._lib.ARRAY_NAMES:
        a = getattr(g, name)
        loc = name.split("_")[1]
        xl, yl = int(loc[0] == "f"), int(loc[1] == "f")
        b = a.clone().view(1, *a.shape)                                        # a 2-D field: Nz = 1, Hz = 0
        assert lib.tpg_fill_halo_regions(osg._lib.ptr_table([b]), 1, (C.c_int8 * 1)(xl), (C.c_int8 * 1)(yl), (C.c_int32 * 1)(1),
                                         Nx, Ny, 1, Hx, Hy, 0, 1, ft, None) == 0
        torch.cuda.synchronize()
        same = (b[0] == a) | (torch.isnan(b[0]) & torch.isnan(a))
        assert bool(same.all()), (name, int((~same).sum()))


@pytest.mark.parametrize("tdt", [torch.float64, torch.float32], ids=["f64", "f32"])
def test_grid_values_do_not_depend_on_the_halo_width(osg, gpu, tdt):
    size = (3600, 1800, 1)
    wide, mid, narrow = (osg.TripolarGrid(None, tdt, size=size, halo=h) for h in ((5, 5, 5), (4, 4, 4), (2, 3, 1)))
    for name in osg._lib.ARRAY_NAMES:
        a5, a4, a2 = getattr(wide, name), getattr(mid, name), getattr(narrow, name)
        for outer, inner, dy, dx in ((a5, a4, 1, 1), (a4, a2, 1, 2)):
            w = outer[dy:outer.shape[0] - dy, dx:outer.shape[1] - dx]
            same = (w == inner) | (torch.isnan(w) & torch.isnan(inner))
            assert w.shape == inner.shape and bool(same.all()), (name, int((~same).sum()))


@pytest.mark.parametrize("size,halo", [((3600, 1800, 1), (4, 4, 4)), ((8640, 4320, 1), (5, 5, 5))], ids=["tenth", "twentyfourth-halo5"])
def test_face_areas_are_the_products_of_their_edge_lengths(osg, gpu, size, halo):
    """src/tripolar_grid_utils.jl:34-35: Az_fc = Dy_fc * Dx_fc and Az_cf = Dy_cf * Dx_cf -- one Float64 multiply, so the stored arrays must
    satisfy it exactly on every row the metric kernel owns (j >= 2; rows j <= 1 are the lat-lon continuation, src/tripolar_grid.jl:277-300)
    and, the halo fills being copies with sign +1 through one index map, on the north fold rows and the periodic columns as well.  It also
    pins WHICH Dy array sits under which name: with Dy_fc and Dy_cf swapped (the positional-order question of src/tripolar_grid.jl:321-324)
    the identity fails."""
    gc.collect(); torch.cuda.empty_cache()
    g = osg.TripolarGrid(size=size, halo=halo)
    Hy = halo[1]
    rows = slice(Hy + 1, None)
    assert torch.equal(g.az_fc[rows], g.dy_fc[rows] * g.dx_fc[rows])
    assert torch.equal(g.az_cf[rows], g.dy_cf[rows] * g.dx_cf[rows])
    assert not torch.equal(g.az_fc[rows], g.dy_cf[rows] * g.dx_fc[rows])
