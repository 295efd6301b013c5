"""Shared helpers for the parity tests."""
import numpy as np

LOCS = {"cc": (0, 0), "fc": (1, 0), "cf": (0, 1), "ff": (1, 1)}


def A(g, name, i, j, halo=(4, 4, 4)):
    """reference-style 1-based access A[i, j] into a padded [row, col] numpy array"""
    return g[name][j + halo[1] - 1, i + halo[0] - 1]


def interior(g, name, size, halo=(4, 4, 4)):
    Nx, Ny = size[0], size[1]
    return g[name][halo[1]:halo[1] + Ny, halo[0]:halo[0] + Nx]


def max_rel_err(got, ref):
    """max |got-ref|/|ref| with exact matches (incl. NaN==NaN, 0==0) counted as 0"""
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    same = (got == ref) | (np.isnan(got) & np.isnan(ref))
    with np.errstate(divide="ignore", invalid="ignore"):
        rel = np.abs(got - ref) / np.abs(ref)
    rel[same] = 0.0
    return float(np.nanmax(rel)) if rel.size else 0.0, int((~same).sum())


def splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)).astype(np.uint64)
    x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return x ^ (x >> np.uint64(31))


def synthetic_field(seed, sentinel, size, halo, dtype=np.float64):
    """host twin of tpg_fill_synthetic"""
    (Nx, Ny, Nz), (Hx, Hy, Hz) = size, halo
    shape = (Nz + 2 * Hz, Ny + 2 * Hy, Nx + 2 * Hx)
    idx = np.arange(np.prod(shape), dtype=np.uint64)
    with np.errstate(over="ignore"):
        h = splitmix64(np.uint64(seed) ^ splitmix64(idx))
    v = ((h >> np.uint64(11)).astype(np.float64) + 0.5) * 2.0 ** -52 - 1.0
    out = np.full(shape, sentinel, dtype=np.float64)
    v = v.reshape(shape)
    out[Hz:Hz + Nz, Hy:Hy + Ny, Hx:Hx + Nx] = v[Hz:Hz + Nz, Hy:Hy + Ny, Hx:Hx + Nx]
    return out.astype(dtype)
