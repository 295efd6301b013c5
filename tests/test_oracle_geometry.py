"""CPU checks of the oracle's geometry utilities: the non-orthogonality diagnostic of
test/test_tripolar_grid.jl:8-34,49-75 and the frame rotation of examples/convert_to_latlong_frame.jl:12-55.

The reference bounds the tripolar grid's angle by the range found on a conformal cubed-sphere panel (:74-75); the cubed
sphere is Oceananigans' (absent here), so that bound is replaced by a stated one and is "parity unpinned": what IS
checked against reference-held facts is the test's set-up (1-degree grid, poles 35N / 75E, mask of :59-60) and the
properties the example relies on (the two conversions are inverse rotations; a zonal unit vector stays a unit vector)."""
import mpmath as mp
import numpy as np
import pytest


def _masked_setup(oracle):
    size, halo = (360, 180, 1), (4, 4, 4)
    g = oracle.build_grid(size, halo=halo, first_pole_longitude=75, north_poles_latitude=35)     # :52-57
    lam = g["lambda_cc"][4:-4, 4:-4]; phi = g["phi_cc"][4:-4, 4:-4]
    l1, pp = 75.0, 35.0
    l2 = l1 + 180
    mask = ((np.abs(lam - l1) < 5) & (np.abs(pp - phi) < 5)) | ((np.abs(lam - l2) < 5) & (np.abs(pp - phi) < 5)) | (phi < -78)   # :59-60
    return size, halo, g, mask


def test_acos_restatement_against_mpmath(oracle):
    mp.mp.dps = 40
    x = np.concatenate([np.linspace(-1, 1, 2001), np.random.default_rng(1).uniform(-1, 1, 4000), [1e-20, -1e-20, 0.5, -0.5, 0.4999999, 0.975]])
    y = oracle.math_probe("acos", x)
    worst = 0.0
    for a, b in zip(x, y):
        e = mp.acos(mp.mpf(float(a)))
        ulp = np.spacing(abs(float(e))) if e != 0 else 1.0
        worst = max(worst, abs(float((mp.mpf(float(b)) - e) / ulp)))
    assert worst < 1.0, worst                       # msun acos: < 1 ulp


def test_tripolar_grid_is_orthogonal_away_from_the_singularities(oracle):
    size, halo, g, mask = _masked_setup(oracle)
    ang = oracle.nonorthogonality_angle(g["lambda_ff"], g["phi_ff"], size, halo, immersed=mask)
    assert ang.shape == (180, 360)
    assert np.all(ang[-1] == 0) and np.all(ang[:, -1] == 0)          # outside the (Nx-1, Ny-1) launch (:70)
    assert np.all(ang[mask] == 0)                                     # immersed -> pi/2 - pi/2 (:29)
    # builder-stated bound (the reference's is the cubed-sphere panel range, :74-75: parity unpinned)
    assert ang.max() < 2.0 and ang.min() > -2.0, (ang.min(), ang.max())
    # unmasked, the three singular neighbourhoods are NOT orthogonal: the diagnostic must see them
    raw = oracle.nonorthogonality_angle(g["lambda_ff"], g["phi_ff"], size, halo)
    assert np.abs(raw).max() > 10.0


def test_nonorthogonality_against_mpmath_on_sample_nodes(oracle):
    """the angle formula re-evaluated with 40 digits from the stored Float64 coordinates"""
    mp.mp.dps = 40
    size, halo = (60, 30, 1), (4, 4, 4)
    g = oracle.build_grid(size, halo=halo)
    ang = oracle.nonorthogonality_angle(g["lambda_ff"], g["phi_ff"], size, halo)
    def P(i, j):
        l = mp.radians(mp.mpf(float(g["lambda_ff"][j + 3, i + 3]))); p = mp.radians(mp.mpf(float(g["phi_ff"][j + 3, i + 3])))
        return mp.matrix([mp.cos(l) * mp.cos(p), mp.sin(l) * mp.cos(p), mp.sin(p)])
    for i, j in ((5, 5), (30, 20), (17, 28), (59, 29), (1, 1), (31, 29)):
        p0, p1, p2 = P(i, j), P(i + 1, j), P(i, j + 1)
        v1, v2 = p1 - p0, p2 - p0
        c = (v1.T * v2)[0] / (mp.norm(v1) * mp.norm(v2))
        want = mp.degrees(mp.acos(c) - mp.pi / 2)
        assert abs(float(want) - ang[j - 1, i - 1]) < 1e-9, (i, j, float(want), ang[j - 1, i - 1])


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_frame_conversions_round_trip(oracle, dtype):
    size, halo = (180, 90, 2), (4, 4, 4)                                                  # the example's grid (:58)
    g = oracle.build_grid(size, dtype=dtype, halo=halo, north_poles_latitude=35)
    rng = np.random.default_rng(3)
    shape = (2 + 8, 90 + 8, 180 + 8)
    u = rng.uniform(-1, 1, shape).astype(dtype); v = rng.uniform(-1, 1, shape).astype(dtype)
    ul, vl = oracle.convert_frame(g, u, v, size, halo, to_native=False)
    ub, vb = oracle.convert_frame(g, ul, vl, size, halo, to_native=True)
    I = (slice(4, -4), slice(4, -4), slice(4, -4))
    tol = 1e-13 if dtype == np.float64 else 1e-5
    # the example's "native" conversion (:54) returns (u d1 + v d2, u d2 - v d1): the inverse rotation with the SECOND
    # component negated -- kept as written in the reference, so a round trip gives (u, -v)
    assert np.max(np.abs(ub[I] - u[I])) < tol and np.max(np.abs(vb[I] + v[I])) < tol
    speed2 = u[I].astype(np.float64) ** 2 + v[I].astype(np.float64) ** 2
    assert np.max(np.abs(ul[I].astype(np.float64) ** 2 + vl[I].astype(np.float64) ** 2 - speed2)) < 10 * tol   # norm kept
    assert np.all(ul[0] == 0) and np.all(ul[:, :4] == 0)                                   # only the interior is written
    # purely zonal unit flow (:64): far from the northern poles the grid lines are nearly parallels and meridians, so the
    # rotation is nearly the identity (d1 -> 1, d2 -> 0); it stays a unit vector everywhere
    one = np.ones(shape, dtype=dtype); zero = np.zeros(shape, dtype=dtype)
    uz, vz = oracle.convert_frame(g, one, zero, size, halo, to_native=True)
    assert np.max(np.abs(uz[4:-4, 4:10, 4:-4] - 1)) < 1e-4 and np.max(np.abs(vz[4:-4, 4:10, 4:-4])) < 5e-3
    assert np.max(np.abs(uz[I].astype(np.float64) ** 2 + vz[I].astype(np.float64) ** 2 - 1)) < 10 * tol
