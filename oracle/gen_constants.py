#!/usr/bin/env python3
"""Derive (with mpmath, 200 digits) every constant of oracle/detmath.h that has a closed form.

TEST INFRASTRUCTURE (oracle side).  Run:  python oracle/gen_constants.py
The minimax polynomial coefficients (sin/cos/atan/asin kernels, FreeBSD msun family) are not
derivable; they are validated instead by tests/test_detmath.py against mpmath to <1 ulp.
"""
import mpmath as mp
mp.mp.prec = 700

def d(x):            # round-to-nearest double
    return float(mp.mpf(x))
def hexs(x):
    return float(x).hex()
def trunc_bits(x, nbits):
    """x truncated to its leading nbits significant bits (as msun's pio2_1 etc.)."""
    x = mp.mpf(x); e = mp.floor(mp.log(abs(x), 2))
    s = mp.mpf(2) ** (e - nbits + 1)
    return mp.floor(x / s) * s

def show(name, v):
    print(f"{name:12s} = {hexs(v):26s} /* {float(v):.21e} */")

pi = mp.pi
show("PI", d(pi))
show("DEG2RAD", d(pi) / 180.0)                       # Julia: deg2rad(x) = x*(Float64(pi)/180)
m = d(pi / 180); show("D2R_HI", m); show("D2R_LO", d(pi / 180 - mp.mpf(m)))
show("C180_PI", 180.0 / d(pi)); show("C360_PI", 360.0 / d(pi))
show("INVPIO2", d(2 / pi))
p1 = trunc_bits(pi / 2, 33); show("PIO2_1", d(p1)); r = pi / 2 - p1; show("PIO2_1T", d(r))
p2 = trunc_bits(r, 33);      show("PIO2_2", d(p2)); r2 = r - p2;     show("PIO2_2T", d(r2))
p3 = trunc_bits(r2, 33);     show("PIO2_3", d(p3)); r3 = r2 - p3;    show("PIO2_3T", d(r3))
for nm, v in (("ATAN_0_5", mp.atan(mp.mpf(1) / 2)), ("ATAN_1", mp.atan(1)),
              ("ATAN_1_5", mp.atan(mp.mpf(3) / 2)), ("ATAN_INF", pi / 2)):
    hi = d(v); show(nm + "_HI", hi); show(nm + "_LO", d(v - mp.mpf(hi)))
hi = d(pi / 4); show("PIO4_HI", hi)
hi = d(mp.log(2)); show("LN2_HI", hi); show("LN2_LO", d(mp.log(2) - mp.mpf(hi)))
print("/* double-double 1/k!, k = 2..11 (Taylor coefficients of dm_dd_expm1_reduced) */")
for k in range(2, 12):
    v = mp.mpf(1) / mp.factorial(k)
    hi = d(v)
    print(f"    {{ {hexs(hi)}, {hexs(d(v - mp.mpf(hi)))} }},   /* 1/{k}! */")
