"""Pins the oracle (CPU restatement) against every known answer the reference itself holds for the
metric-precompute path: the README transcript (6 digits), the unit-test bounds of
test/runtests.jl:8-41, and the structural identities derivable from the source."""
import os

import numpy as np
import pytest

from helpers import A, interior
from conftest import GOLDEN

R = 6371.0e3


def _sig(x, n=6):
    return float(f"{x:.{n}g}")


def test_readme_transcript(oracle, kats):
    k = kats["readme_60x30"]
    size = tuple(k["size"])
    g = oracle.build_grid(size)
    Nx, Ny, _ = size
    lam0, phi0 = A(g, "lambda_ff", Nx // 2 + 1, Ny // 2 + 1), A(g, "phi_ff", Nx // 2 + 1, Ny // 2 + 1)
    assert lam0 == k["center_lambda_phi"][0]
    assert round(phi0, 4) == k["center_lambda_phi"][1]
    dxcf, dxff = interior(g, "dx_cf", size), interior(g, "dx_ff", size)
    dyfc, dyff = interior(g, "dy_fc", size), interior(g, "dy_ff", size)
    assert _sig(np.rad2deg(dxcf[15 - 1, :].sum()) / R) == k["longitude_extent_deg"]
    assert _sig(np.rad2deg(dxff.min()) / R) == k["min_dlambda"]
    assert _sig(np.rad2deg(dxff.max()) / R) == k["max_dlambda"]
    assert _sig(np.rad2deg(dyfc[:, 16 - 1].sum()) / R) == k["latitude_extent_deg"]
    assert _sig(np.rad2deg(dyff.min()) / R) == k["min_dphi"]
    assert _sig(np.rad2deg(dyff.max()) / R) == k["max_dphi"]


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_unit_tests_of_runtests_jl(oracle, kats, dtype):
    k = kats["unit_4x5"]
    size = tuple(k["size"])
    g = oracle.build_grid(size, dtype=dtype, first_pole_longitude=k["first_pole_longitude"],
                          north_poles_latitude=k["north_poles_latitude"],
                          southernmost_latitude=k["southernmost_latitude"])
    assert g["lambda_cc"].dtype == dtype                       # eltype(grid) == FT  (runtests.jl:16)
    lam, phi = interior(g, "lambda_cc", size), interior(g, "phi_cc", size)
    min_dphi = (phi[1, :] - phi[0, :]).min()
    assert lam.min() >= 0 and lam.max() <= 360                 # :32-33
    assert phi.max() <= 90                                     # :34
    assert (phi + min_dphi / 10).min() >= k["southernmost_latitude"]   # :36-39


def test_odd_longitude_is_an_argument_error(oracle):
    with pytest.raises(ValueError, match="should be even"):
        oracle.build_grid((61, 30, 1))                         # tripolar_grid.jl:81-83


def test_structural_identities(oracle):
    size, fpl, npl = (60, 30, 1), 70, 55
    g = oracle.build_grid(size, first_pole_longitude=fpl, north_poles_latitude=npl)
    Nx, Ny, _ = size
    for name in ("lambda_fc", "lambda_ff"):                    # pole meridians after the shift (B-2)
        assert np.all(interior(g, name, size)[:, 0] == (fpl + 180) % 360)
        assert np.all(interior(g, name, size)[:, Nx // 2] == fpl)
    assert A(g, "phi_fc", 1, Ny) == npl and A(g, "phi_fc", Nx // 2 + 1, Ny) == npl
    for name in ("lambda_cc", "lambda_fc", "lambda_cf", "lambda_ff"):
        a = interior(g, name, size)
        assert a.min() >= 0 and a.max() < 360
    for name in ("phi_cc", "dx_cc", "dy_cc", "az_cc"):         # row Ny mirror symmetry (C x-location)
        row = interior(g, name, size)[Ny - 1]
        assert np.array_equal(row, row[::-1])
    row = interior(g, "phi_fc", size)[Ny - 1]                  # F x-location: i <-> Nx-i+2
    assert np.array_equal(row[1:], row[1:][::-1])
    # south halos of the coordinates stay zero (tripolar_grid.jl:148)
    for name in ("lambda_cc", "phi_ff"):
        assert np.all(g[name][:4, :] == 0)


def test_spherical_cap_area_identity(oracle):
    """B-3: unit-sphere sum of Az_cc (row Ny counted half) ~ area north of phi_f[1]."""
    size = (60, 30, 1)
    g = oracle.build_grid(size, radius=1.0)
    az = interior(g, "az_cc", size).astype(np.float64)
    # row 1 is the lat-lon continuation (continue_south! overwrites it): rebuild the cap from rows >= 2
    _, _, pf, _ = oracle.tables(size)
    total = az[1:-1].sum() + 0.5 * az[-1].sum()
    cap = 2 * np.pi * (1 - np.sin(np.deg2rad(pf[1])))
    assert abs(total - cap) / cap < 5e-5


def _polygon_area_left(lams, phis):
    """unit-sphere area of the region to the LEFT of the closed geodesic polygon through the (lambda, phi) vertices [degrees],
    by Gauss-Bonnet (2 pi minus the sum of the turning angles), in 40-digit arithmetic"""
    import mpmath as mp
    mp.mp.dps = 40
    V = []
    for l, p in zip(lams, phis):
        l, p = mp.radians(mp.mpf(float(l))), mp.radians(mp.mpf(float(p)))
        V.append((mp.cos(l) * mp.cos(p), mp.sin(l) * mp.cos(p), mp.sin(p)))
    cross = lambda a, b: (a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0])
    dot = lambda a, b: a[0] * b[0] + a[1] * b[1] + a[2] * b[2]
    unit = lambda a: tuple(x / mp.sqrt(dot(a, a)) for x in a)
    turning = mp.mpf(0)
    for k in range(len(V)):
        a, b, c = V[k - 1], V[k], V[(k + 1) % len(V)]
        tin, tout = unit(cross(cross(a, b), b)), unit(cross(cross(b, c), b))       # arrival / departure directions at b
        turning += mp.atan2(dot(b, cross(tin, tout)), dot(tin, tout))              # left turn positive
    return 2 * mp.pi - turning


@pytest.mark.parametrize("size,kw", [((60, 30, 1), {}),
                                     ((120, 60, 1), dict(north_poles_latitude=65, first_pole_longitude=10, southernmost_latitude=-70))])
def test_cell_areas_tile_the_sphere_exactly(oracle, size, kw):
    """Pins Az_cc / Az_ff (no reference-held value does) through an exact identity: the spherical quadrilaterals of
    src/tripolar_grid_utils.jl:23-28,38-43 have great-circle edges shared with their neighbours, so they TILE --
      * sum over rows 2..Ny-1 of Az_cc + half of row Ny (its cells straddle the fold: cell i and cell Nx-i+1 are one region,
        vertices FF[., Ny] and their fold images) = area north of the closed polygon through the FF nodes of row 2;
      * sum over rows 2..Ny of Az_ff = area north of the polygon through the CC nodes of row 1 (the CC nodes of row Ny lie on
        the fold line, out and back: they enclose nothing)
    (row 1 is excluded: continue_south! overwrites it with lat-lon values).  The polygon areas come from Gauss-Bonnet in
    40 digits on the stored Float64 coordinates.  Measured agreement: 7e-18 / 1e-17 relative."""
    import mpmath as mp
    g = oracle.build_grid(size, radius=1.0, **kw)
    azcc, azff = interior(g, "az_cc", size), interior(g, "az_ff", size)
    total_cc = mp.fsum(mp.mpf(float(x)) for x in azcc[1:-1].ravel()) + mp.fsum(mp.mpf(float(x)) for x in azcc[-1]) / 2
    exact_cc = _polygon_area_left(interior(g, "lambda_ff", size)[1], interior(g, "phi_ff", size)[1])
    total_ff = mp.fsum(mp.mpf(float(x)) for x in azff[1:].ravel())
    exact_ff = _polygon_area_left(interior(g, "lambda_cc", size)[0], interior(g, "phi_cc", size)[0])
    assert abs(total_cc - exact_cc) / exact_cc < 1e-14, float(abs(total_cc - exact_cc) / exact_cc)
    assert abs(total_ff - exact_ff) / exact_ff < 1e-14, float(abs(total_ff - exact_ff) / exact_ff)
    assert 2 * mp.pi < exact_cc < 4 * mp.pi                               # the region is the sphere minus a southern cap


def test_continue_south_rows(oracle):
    """rows j = 1-Hy..1 (interior row 1 included) hold the lat-lon metrics (tripolar_grid.jl:336-357)"""
    size = (60, 30, 1)
    g = oracle.build_grid(size)
    dy = R * np.deg2rad(170 / 30)
    for name in ("dy_cc", "dy_cf", "dy_fc", "dy_ff"):
        assert np.allclose(g[name][:5, :], dy, rtol=1e-15)
    for name in ("dx_cc", "az_ff"):
        assert np.all(g[name][:5, :] == g[name][:5, :1])       # 1-D in j: constant along i, halos included


def test_band_equals_slice_of_global(oracle):
    size, halo = (20, 12, 1), (3, 2, 1)
    full = oracle.build_grid(size, halo=halo)
    band = oracle.build_grid(size, halo=halo, jstart=5, jend=8)
    for name in full:
        assert np.array_equal(band[name], full[name][4:4 + 4 + 2 * 2], equal_nan=True)   # rows jstart-Hy..jend+Hy


@pytest.mark.parametrize("fname", sorted(f for f in os.listdir(GOLDEN) if f.endswith(".npz")))
def test_restatement_golden_vectors(oracle, fname):
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(GOLDEN, "make_golden.py"))
    mg = importlib.util.module_from_spec(spec); spec.loader.exec_module(mg)
    gold = np.load(os.path.join(GOLDEN, fname))
    g = oracle.build_grid(**mg.CASES[fname[:-4]])
    for name in gold.files:
        assert np.array_equal(g[name], gold[name], equal_nan=True), name


def test_orthogonality_property(oracle):
    """test/test_tripolar_grid.jl:36-76 checks the FF-node non-orthogonality of a 1 degree grid
    (poles 75E/35N, 5-degree pole boxes and phi < -78 masked) against the range of a 90x90
    cubed-sphere panel.  The cubed-sphere generator is Oceananigans-internal and absent here, so
    the bound is stated directly: the chord-angle deviation stays within +-2 degrees."""
    size, H = (360, 180, 1), 4
    oracle.set_threads(min(8, oracle.max_threads()))
    g = oracle.build_grid(size, north_poles_latitude=35, first_pole_longitude=75)
    oracle.set_threads(1)
    lam, phi = np.deg2rad(g["lambda_ff"]), np.deg2rad(g["phi_ff"])
    xyz = [np.cos(lam) * np.cos(phi), np.sin(lam) * np.cos(phi), np.sin(phi)]
    Nx, Ny = size[0], size[1]
    sl = lambda a, di, dj: a[H + dj:H + Ny - 1 + dj, H + di:H + Nx - 1 + di]
    v1 = np.stack([sl(a, 1, 0) - sl(a, 0, 0) for a in xyz])
    v2 = np.stack([sl(a, 0, 1) - sl(a, 0, 0) for a in xyz])
    cos = (v1 * v2).sum(0) / np.sqrt((v1 * v1).sum(0) * (v2 * v2).sum(0))
    ang = np.rad2deg(np.arccos(cos)) - 90
    L, P = sl(g["lambda_ff"], 0, 0), sl(g["phi_ff"], 0, 0)
    mask = ((abs(L - 75) < 5) & (abs(35 - P) < 5)) | ((abs(L - 255) < 5) & (abs(35 - P) < 5)) | (P < -78)
    ang = np.where(mask, 0.0, ang)
    assert ang.max() < 2.0 and ang.min() > -2.0
