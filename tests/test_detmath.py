"""The oracle's deterministic elementary functions (oracle/detmath.h) against mpmath.
The msun-family minimax coefficients are recalled, not derivable: an error in any of them shows
up here as an error >> 1 ulp."""
import mpmath as mp
import numpy as np
import pytest

mp.mp.prec = 200
RNG = np.random.default_rng(2024)


def worst_ulp(oracle, name, xs, f):
    ys = oracle.math_probe(name, xs)
    worst = 0.0
    for x, y in zip(xs, ys):
        t = f(mp.mpf(float(x)))
        if t == 0:
            assert y == 0
            continue
        u = np.spacing(abs(float(t)))
        worst = max(worst, float(abs((mp.mpf(float(y)) - t) / mp.mpf(float(u)))))
    return worst


CASES = [
    ("sin", lambda: RNG.uniform(-4, 4, 600), mp.sin, 1.0),
    ("sin", lambda: np.pi + RNG.uniform(-1e-3, 1e-3, 300), mp.sin, 1.0),       # seam-crossing haversines
    ("cos", lambda: RNG.uniform(-4, 4, 600), mp.cos, 1.0),
    ("sind", lambda: RNG.uniform(-360, 360, 600), lambda x: mp.sin(x * mp.pi / 180), 1.0),
    ("cosd", lambda: RNG.uniform(-360, 360, 600), lambda x: mp.cos(x * mp.pi / 180), 1.0),
    ("tand", lambda: RNG.uniform(0, 85, 300), lambda x: mp.tan(x * mp.pi / 180), 2.0),
    ("atan", lambda: np.concatenate([RNG.uniform(-5, 5, 600), 10.0 ** RNG.uniform(-30, 30, 200)]), mp.atan, 1.0),
    ("asin", lambda: np.concatenate([RNG.uniform(-1, 1, 600), 10.0 ** RNG.uniform(-8, 0, 200)]), mp.asin, 1.0),
    ("asinh", lambda: np.concatenate([RNG.uniform(0, 40, 200), 10.0 ** RNG.uniform(-8, 0, 100)]), mp.asinh, 0.501),
    ("sinh", lambda: np.concatenate([RNG.uniform(0, 5, 200), 10.0 ** RNG.uniform(-8, 0, 100)]), mp.sinh, 0.501),
    ("cosh", lambda: np.concatenate([RNG.uniform(0, 5, 200), 10.0 ** RNG.uniform(-8, 0, 100)]), mp.cosh, 0.501),
]


@pytest.mark.parametrize("name,gen,f,bound", CASES, ids=[f"{c[0]}-{i}" for i, c in enumerate(CASES)])
def test_ulp_accuracy(oracle, name, gen, f, bound):
    assert worst_ulp(oracle, name, gen(), f) < bound


def test_degree_exact_values(oracle):
    """Julia Base semantics the pole detection relies on"""
    s = oracle.math_probe("sind", np.array([-180.0, 180.0, 0.0, 90.0, -90.0, 360.0]))
    assert list(s) == [0.0, 0.0, 0.0, 1.0, -1.0, 0.0]
    assert np.signbit(s[0]) and not np.signbit(s[1])           # sind(-180) = -0.0, sind(180) = +0.0
    c = oracle.math_probe("cosd", np.array([90.0, -90.0, 270.0, 0.0, 180.0]))
    assert list(c) == [0.0, 0.0, 0.0, 1.0, -1.0]
    assert not np.signbit(c[0]) and not np.signbit(c[1])
    a = oracle.math_probe("atan", np.array([np.inf, -np.inf, 0.0, -0.0]))
    assert a[0] == np.pi / 2 and a[1] == -np.pi / 2 and a[2] == 0 and np.signbit(a[3])
    assert oracle.math_probe("sinh", np.array([0.0]))[0] == 0.0
    assert oracle.math_probe("cosh", np.array([0.0]))[0] == 1.0
    assert oracle.math_probe("asinh", np.array([0.0]))[0] == 0.0


def test_lambda_phi_tables(oracle):
    """A-2/A-3: range elements are correctly rounded exact rationals"""
    from fractions import Fraction
    lf, lc, pf, pc = oracle.tables((3600, 1800, 1))
    for i in (1, 2, 7, 901, 1801, 3600):
        assert lf[i - 1] == float(Fraction(-180) + Fraction(360 * (i - 1), 3600))
        assert lc[i - 1] == float(Fraction(-180) + Fraction(360 * (2 * i - 1), 7200))
    assert lf[0] == -180.0 and lf[1800] == 0.0
    for j in (1, 2, 3, 900, 1800):
        assert pc[j - 1] == float(Fraction(-80) + Fraction(170 * (j - 1), 1799))
    assert pc[-1] == 90.0
    dphi = pc[1] - pc[0]
    assert np.array_equal(pf, pc - dphi / 2)
