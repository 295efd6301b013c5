"""bench_common.py -- what bench.py (the N = 1 step and the contract line), bench_chain.py (the N > 1 latitude-band run) and
bench_aux.py (the auxiliary measurements behind the timed region) share: the workload constants, the algorithmic byte counts of
SURVEY.md 8(d), the precompute roofline object, the PMC traffic table.  No torch import, no device."""
import hashlib
import json
import os

ROOT = os.path.dirname(os.path.abspath(__file__))

NX, NY, NZ, H = 3600, 1800, 75, 4
SPECS = [("c", 0, 0, 1), ("u", 1, 0, -1), ("v", 0, 1, -1), ("zeta", 1, 1, 1)]   # name, xloc, yloc, sign
HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FP64_VALU_PEAK_TFLOPS = 78.6    # MI355X vector FP64 (datasheet)
FLOP_PER_CELL = 2285.0          # FP64 add/mul/fma (fma = 2) per evaluated cell, PMC-counted on the round-3 kernel (profiles/r03/cells_trims.txt)
LIB = os.path.join(ROOT, "orthogonalsphericalshellgrids.jl_amd", "libtripolar_hip.so")
TESTLIB = os.path.join(ROOT, "tools", "libtripolar_hip_test.so")
METRIC = "grid-cells/s metric precompute + zipper halo-fill GB/s, 1/10°×75z"


def libraries_built():
    return os.path.exists(LIB) and os.path.exists(TESTLIB)


def zipper_algorithmic_bytes(nx, nz, hy, specs=SPECS, s=8):
    """CF/FF fields Nx*Nz*Hy*2*s; CC/FC add the row-Ny substitution (Nx/2)*Nz*2*s"""
    per_field = {}
    for name, xl, yl, _ in specs:
        b = nx * nz * hy * 2 * s
        if yl == 0:
            b += (nx // 2) * nz * 2 * s
        per_field[name] = b
    return per_field


def periodic_algorithmic_bytes(ny, nz, h, nfields, s=8):
    """Oceananigans' periodic west/east fill: 2 Hx elements read + 2 Hx written per row, every row and level of the parent"""
    return nfields * (ny + 2 * h) * (nz + 2 * h) * 2 * h * 2 * s


def precompute_roofline(t_build_ms, evaluated_cells, band_cells, traffic, slowest_rank=False):
    """`roofline_precompute`: FP64 VALU issue is what bounds tpg_build_grid (VALU busy 91 %), so THAT is `bound` / `frac`; its store stream
    is secondary.  flops are per cell the kernel evaluates (band rows + the seam halo rows it computes), bytes per padded cell it stores."""
    tflops = FLOP_PER_CELL * evaluated_cells / (t_build_ms * 1e-3) / 1e12
    gbps = 160.0 * band_cells / (t_build_ms * 1e-3) / 1e9
    return {
        "kernel": "tpg_build_grid (k_tables + k_cells_tile + k_halos)" + (", slowest rank" if slowest_rank else ""), "bound": "fp64_valu",
        "achieved": tflops, "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tflops / FP64_VALU_PEAK_TFLOPS,
        "flop_per_cell": FLOP_PER_CELL, "evaluated_cells": evaluated_cells, "stored_cells": band_cells,
        "hbm_GBps": gbps, "hbm_frac": gbps / HBM_PEAK_GBPS,
        "algorithmic_bytes_per_launch": 160 * band_cells, "traffic": traffic.get("k_cells_tile"),
        "note": "FP64-issue bound (VALU busy 91 %%): %.1f of the %.1f TFLOP/s vector FP64 peak at 2.29 kflop/cell (PMC count); the 160 B/cell store "
                "stream is %.0f %%%% of HBM peak" % (tflops, FP64_VALU_PEAK_TFLOPS, 100 * gbps / HBM_PEAK_GBPS)}


def load_traffic():
    """PMC traffic per kernel (profiles/traffic.json), valid only for the build it was measured on: every kernel entry carries the
    file its kernel lives in and that file's hash at measurement time; an entry whose source has changed since is dropped."""
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.exists(tpath):
        return {}
    with open(tpath) as f:
        tj = json.load(f)
    out, hashes = {}, {}
    for k, v in tj.get("kernels", {}).items():
        srcs = tuple(v.get("sources") or ())
        if not srcs:
            continue
        if srcs not in hashes:
            hashes[srcs] = sources_sha16(srcs)
        if hashes[srcs] is not None and hashes[srcs] == v.get("sources_sha16"):
            out[k] = v.get("hbm_bytes_per_launch")
    return out


def sources_sha16(srcs):
    """one hash over the files (repo-relative) a kernel is compiled from; None if one is missing"""
    h = hashlib.sha256()
    for rel in srcs:
        path = os.path.join(ROOT, rel)
        if not os.path.exists(path):
            return None
        h.update(open(path, "rb").read())
    return h.hexdigest()[:16]
