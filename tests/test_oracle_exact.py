"""End-to-end accuracy of the restatement (and therefore of the bit-identical HIP path) against an
independent 60-digit evaluation of the reference's formulas with mpmath -- a third statement of
src/generate_tripolar_coordinates.jl:66-87 and src/tripolar_grid_utils.jl:13-43, written directly from
the Julia source rather than from oracle/tpg_oracle.c.

Two claims, tolerance as in BASELINE.json's north_star (1e-12 relative for Float64 metrics):
  (a) coordinates: |lambda, phi - exact| <= 2e-13 degrees, the 1-D tables taken as the reference's
      Float64 inputs (they are pinned exactly by tests/test_detmath.py::test_lambda_phi_tables);
  (b) metrics: every Dx, Dy (and the product areas Az_fc, Az_cf) within 1e-12 relative of the exact
      haversine value of the Float64 coordinates the formulas are given (the reference also evaluates them
      on its stored Float64 coordinate arrays) wherever the edge subtends at least ~0.1 degree, and within
      2 eps / (edge angle) everywhere: Distances.haversine differences two ROUNDED radian latitudes
      (deg2rad(phi2) - deg2rad(phi1)), so any Float64 evaluation -- the reference's included -- loses
      eps / angle; the refined cells around the two northern poles of a 1/10 degree grid reach 3e-12;
  (c) the spherical-excess areas Az_cc, Az_ff within eps / (solid angle of the cell): the triangle
      formula |a.(b x c)| / (1 + a.b + b.c + a.c) cancels ~(cell angle)^2 of its leading digits in ANY
      Float64 evaluation, the reference's included -- at 1/10 degree that is ~1e-10 relative, so these two
      arrays cannot agree with the Julia reference to 1e-12 unless every rounding matches ("parity
      unpinned" for Az in SURVEY.md 8c); what can be asserted is that the error stays at the formula's
      conditioning.
The cells are sampled away from the fold, the seam and the south edge so that only plain (i+-1, j+-1)
neighbours enter; those index maps are pinned bit-exactly by the zipper tests and the identities in
tests/test_oracle_kat.py.
"""
import mpmath as mp
import numpy as np
import pytest

from helpers import A

mp.mp.dps = 60
R = 6371.0e3
D2R = mp.pi / 180
EPS = 2.220446049250313e-16


def exact_point(lam1d, phi1d, i0, Nx, npl, fpl):
    """generate_tripolar_coordinates.jl:66-87 at one pre-shift index i0 (1-based), exact arithmetic"""
    a = mp.tan((90 - mp.mpf(npl)) / 2 * D2R)                                  # tripolar_grid.jl:76
    psi = mp.asinh(mp.tan((90 - mp.mpf(phi1d)) / 2 * D2R) / a)                # :66
    x = a * mp.sin(mp.mpf(lam1d) * D2R) * mp.cosh(psi)                        # :67
    y = a * mp.cos(mp.mpf(lam1d) * D2R) * mp.sinh(psi)                        # :68
    lam = -180 / mp.pi * mp.atan(y / x)                                       # :77 (no pole among the samples)
    phi = 90 - 360 / mp.pi * mp.atan(mp.sqrt(y * y + x * x))                  # :78
    lam += -90 if i0 <= Nx // 2 else 90                                       # :82
    lam += mp.mpf(fpl) + 90                                                   # :86
    lam = mp.fmod(mp.fmod(lam, 360) + 360, 360)                               # :87
    return lam, phi


def hav(p, q):
    """Distances.haversine((lam1, phi1), (lam2, phi2), R)"""
    (l1, p1), (l2, p2) = p, q
    dl, dp = (mp.mpf(l2) - mp.mpf(l1)) * D2R, (mp.mpf(p2) - mp.mpf(p1)) * D2R
    h = mp.sin(dp / 2) ** 2 + mp.cos(mp.mpf(p1) * D2R) * mp.cos(mp.mpf(p2) * D2R) * mp.sin(dl / 2) ** 2
    return 2 * R * mp.asin(mp.sqrt(h))


def cart(p):
    """Oceananigans lat_lon_to_cartesian(phi, lambda, 1)"""
    lam, phi = mp.mpf(p[0]) * D2R, mp.mpf(p[1]) * D2R
    return (mp.cos(lam) * mp.cos(phi), mp.sin(lam) * mp.cos(phi), mp.sin(phi))


def tri(a, b, c):
    dot = lambda u, v: u[0] * v[0] + u[1] * v[1] + u[2] * v[2]
    cross = (b[1] * c[2] - b[2] * c[1], b[2] * c[0] - b[0] * c[2], b[0] * c[1] - b[1] * c[0])
    return 2 * mp.atan(abs(dot(a, cross)) / (1 + dot(a, b) + dot(b, c) + dot(a, c)))


def quad(a, b, c, d):
    return (tri(a, b, c) + tri(a, b, d) + tri(a, c, d) + tri(b, c, d)) / 2


CASES = [((60, 30, 1), 24), ((360, 180, 1), 24), ((3600, 1800, 1), 10)]


@pytest.mark.parametrize("size,nsamples", CASES, ids=[f"{s[0]}x{s[1]}" for s, _ in CASES])
def test_metrics_and_coordinates_against_exact_arithmetic(oracle, size, nsamples):
    Nx, Ny, _ = size
    npl, fpl = 55, 70
    g = oracle.build_grid(size, north_poles_latitude=npl, first_pole_longitude=fpl)
    lf, lc, pf, pc = oracle.tables(size, north_poles_latitude=npl, first_pole_longitude=fpl)
    shift = Nx // 4
    rng = np.random.default_rng(Nx)
    cells = [(int(rng.integers(3, Nx - 2)), int(rng.integers(3, Ny - 2))) for _ in range(nsamples)]
    cells += [(Nx // 2 + 1, Ny // 2 + 1), (3, Ny - 3), (Nx - 3, 3)]
    worst_coord, worst_metric, worst_area, worst_edge = 0.0, 0.0, 0.0, 0.0
    for i, j in cells:
        # (a) coordinates of the four locations of cell (i, j): array index i <-> pre-shift index i0
        i0 = i - shift
        if i0 < 1:
            i0 += Nx
        for name, lam1d, phi1d in (("ff", lf, pf), ("fc", lf, pc), ("cf", lc, pf), ("cc", lc, pc)):
            if float(lam1d[i0 - 1]) in (-180.0, 0.0, 180.0):
                continue        # pole meridians: x = +-0 and y/x = +-Inf by the sign of zero (pinned by the KAT tests instead)
            lam, phi = exact_point(float(lam1d[i0 - 1]), float(phi1d[j - 1]), i0, Nx, npl, fpl)
            dlam = abs(mp.mpf(float(A(g, "lambda_" + name, i, j))) - lam)
            dlam = min(dlam, abs(dlam - 360))                                # a longitude that wraps at 0 / 360
            worst_coord = max(worst_coord, float(dlam), float(abs(mp.mpf(float(A(g, "phi_" + name, i, j))) - phi)))
        # (b) metrics from the stored Float64 coordinates (tripolar_grid_utils.jl:13-43)
        P = lambda name, ii, jj: (float(A(g, "lambda_" + name, ii, jj)), float(A(g, "phi_" + name, ii, jj)))
        want = {
            "dx_cc": hav(P("fc", i + 1, j), P("fc", i, j)), "dx_fc": hav(P("cc", i, j), P("cc", i - 1, j)),
            "dx_cf": hav(P("ff", i + 1, j), P("ff", i, j)), "dx_ff": hav(P("cf", i, j), P("cf", i - 1, j)),
            "dy_cc": hav(P("cf", i, j + 1), P("cf", i, j)), "dy_fc": hav(P("ff", i, j + 1), P("ff", i, j)),
            "dy_cf": hav(P("cc", i, j), P("cc", i, j - 1)), "dy_ff": hav(P("fc", i, j), P("fc", i, j - 1)),
            "az_cc": quad(cart(P("ff", i, j)), cart(P("ff", i + 1, j)), cart(P("ff", i + 1, j + 1)), cart(P("ff", i, j + 1))) * R * R,
            "az_ff": quad(cart(P("cc", i - 1, j - 1)), cart(P("cc", i, j - 1)), cart(P("cc", i, j)), cart(P("cc", i - 1, j))) * R * R,
        }
        want["az_fc"] = want["dy_fc"] * want["dx_fc"]
        want["az_cf"] = want["dy_cf"] * want["dx_cf"]
        for name, w in want.items():
            got = mp.mpf(float(A(g, name, i, j)))
            rel = float(abs(got - w) / abs(w))
            if name in ("az_cc", "az_ff"):
                # the spherical-excess formula takes a triple product of nearly parallel unit vectors: its
                # Float64 evaluation (the reference's too) is conditioned like eps / (solid angle of the cell)
                worst_area = max(worst_area, rel * float(w / (R * R)) / EPS)
            elif name in ("az_fc", "az_cf"):
                theta = min(float(want["dy" + name[2:]] / R), float(want["dx" + name[2:]] / R))
                worst_edge = max(worst_edge, rel * theta / EPS / 2)          # product of two edges
                worst_metric = max(worst_metric, rel if theta >= 1.7e-3 else 0.0)
            else:
                theta = float(w / R)                                         # angle the edge subtends
                worst_edge = max(worst_edge, rel * theta / EPS)
                worst_metric = max(worst_metric, rel if theta >= 1.7e-3 else 0.0)
    assert worst_coord <= 2e-13, worst_coord          # degrees (measured: 8e-14)
    assert worst_metric <= 1e-12, worst_metric        # north_star tolerance, edges of at least ~0.1 degree
    assert worst_edge <= 2.0, worst_edge              # Dx, Dy everywhere: error in units of eps / edge angle (measured: 0.9)
    assert worst_area <= 1.0, worst_area              # Az_cc, Az_ff: error in units of eps / solid angle (measured: 0.11)
