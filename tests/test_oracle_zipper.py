"""Pins the oracle's zipper index/sign map against the reference's own test
(test/test_zipper_boundary_conditions.jl) -- this part of the path is fully pinned."""
import numpy as np
import pytest

SIZE, HALO = (10, 10, 1), (4, 4, 4)
Nx, Ny, Hx, Hy, Hz = 10, 10, 4, 4, 4


def field(value=0.0):
    return np.full((1 + 2 * Hz, Ny + 2 * Hy, Nx + 2 * Hx), value, dtype=np.float64)


def set_interior(f, v):
    f[Hz:Hz + 1, Hy:Hy + Ny, Hx:Hx + Nx] = v


def at(f, i, j, k=1):
    return f[k + Hz - 1, j + Hy - 1, i + Hx - 1]


def test_constant_field_fold(oracle, kats):
    k = kats["zipper_10x10"]["constant_one"]
    c, u, v = field(), field(), field()
    for f in (c, u, v):
        set_interior(f, 1.0)
    sg = kats["zipper_10x10"]["default_sign"]
    oracle.fill_halo_regions(c, 0, 0, sg["c"], SIZE, HALO)
    oracle.fill_halo_regions(u, 1, 0, sg["u"], SIZE, HALO)
    oracle.fill_halo_regions(v, 0, 1, sg["v"], SIZE, HALO)
    north = lambda f: f[Hz, Ny + Hy:Ny + 2 * Hy, :]           # view(c.data, :, Ny+1:Ny+Hy, 1)
    assert np.all(north(c) == k["c_north_halo"])                                         # :35
    assert np.all(north(v) == k["v_north_halo"])                                         # :36
    assert np.all(north(u)[:, Hx + 1:Hx + Nx - 1] == k["u_north_halo_i_2_to_Nx_minus_1"])   # :39-40
    assert np.all(north(u)[:, Hx] == k["u_north_halo_i_1"])                              # :42,44
    assert np.all(north(u)[:, Hx + Nx] == k["u_north_halo_i_Nx_plus_1"])                 # :43,45


def test_row_ny_symmetry(oracle):
    x = np.arange(1, Nx + 1, dtype=np.float64) * 36.0           # stand-in for the x node coordinate
    c, u = field(), field()
    set_interior(c, x[None, None, :]); set_interior(u, x[None, None, :])
    oracle.fill_halo_regions(c, 0, 0, +1, SIZE, HALO)
    oracle.fill_halo_regions(u, 1, 0, -1, SIZE, HALO)
    crow = c[Hz, Hy + Ny - 1, Hx:Hx + Nx]
    urow = u[Hz, Hy + Ny - 1, Hx:Hx + Nx]
    assert np.array_equal(crow, crow[::-1])                     # :65
    assert np.array_equal(urow[1:5], -urow[6:10][::-1])         # :68-72


def test_zipper_leaves_x_halos_to_the_periodic_pass(oracle):
    f = field(12345.0)
    set_interior(f, 1.0)
    oracle.zipper_fill(f, 0, 0, 1, SIZE, HALO)
    assert np.all(f[Hz, Ny + Hy:, :Hx] == 12345.0) and np.all(f[Hz, Ny + Hy:, Hx + Nx:] == 12345.0)
    assert np.all(f[:Hz] == 12345.0) and np.all(f[Hz + 1:] == 12345.0)     # k outside 1..Nz untouched


def test_face_center_self_map_flips_sign_each_fill(oracle):
    """at i = Nx/2+1 the FC row-Ny substitution maps onto itself, so with
    sign -1 the value flips on every halo fill (reference behaviour, kept)."""
    u = field()
    set_interior(u, 3.0)
    oracle.zipper_fill(u, 1, 0, -1, SIZE, HALO)
    assert at(u, Nx // 2 + 1, Ny) == -3.0
    oracle.zipper_fill(u, 1, 0, -1, SIZE, HALO)
    assert at(u, Nx // 2 + 1, Ny) == 3.0


@pytest.mark.parametrize("xloc,yloc", [(0, 0), (1, 0), (0, 1), (1, 1)])
@pytest.mark.parametrize("sgn", [1, -1])
def test_fold_index_map_against_closed_form(oracle, xloc, yloc, sgn):
    """independent python restatement of zipper_boundary_condition.jl:70-138 on a ragged geometry"""
    size, halo = (14, 9, 3), (3, 2, 1)
    (nx, ny, nz), (hx, hy, hz) = size, halo
    rng = np.random.default_rng(7)
    f = rng.uniform(-1, 1, (nz + 2 * hz, ny + 2 * hy, nx + 2 * hx))
    want = f.copy()
    g = lambda i, j, k: (k + hz - 1, j + hy - 1, i + hx - 1)
    for k in range(1, nz + 1):
        for i in range(1, nx + 1):
            ip = nx - i + 2 if xloc == 1 else nx - i + 1
            s = sgn
            if ip > nx:
                s, ip = abs(s), ip - nx
            for j in range(1, hy + 1):
                want[g(i, ny + j, k)] = s * f[g(ip, ny - j + 1 if yloc == 1 else ny - j, k)]
            if yloc == 0 and i > nx // 2:
                want[g(i, ny, k)] = s * f[g(ip, ny, k)]
    oracle.zipper_fill(f, xloc, yloc, sgn, size, halo)
    assert np.array_equal(f, want)
