#!/usr/bin/env python3
"""compare_reference_dump.py -- compare a dump of the REFERENCE package's own outputs (julia/dump_reference.jl, format
"tripolar-reference-dump-1") with libtripolar_hip (on an MI355X) and with the oracle (anywhere): the one command that turns the
"parity unpinned" lines of DESIGN.md 2 into measured figures at a site that has Julia.

    julia --project=<env with OrthogonalSphericalShellGrids> orthogonalsphericalshellgrids.jl_amd/julia/dump_reference.jl <dir>
    python tests/compare_reference_dump.py <dir> [--oracle-only] [--json report.json]

It lives under tests/ because it loads the oracle (test infrastructure: only tests/, smoke() and bench.py's cpu_baseline may).

What is compared, per grid case (TripolarGrid(CPU(), FT; kwargs...) of the reference):
  * lambda, phi (8 arrays): max ABSOLUTE difference in degrees (lambda modulo 360), interior + north / x halos, and the south halo rows
    (the reference leaves them 0.0: src/tripolar_grid.jl:148);
  * Dx, Dy, Az (12 arrays): max RELATIVE difference over rows j >= 2 (target <= 1e-12, north_star), rows j <= 1 (the lat-lon
    continuation of continue_south!, src/tripolar_grid.jl:277-300) reported SEPARATELY;
  * Dy_cf / Dy_fc additionally against each other's dump (src/tripolar_grid.jl:321-324 passes Dy positionally as cc, cf, fc, ff);
and per field case (test/test_zipper_boundary_conditions.jl:5-31,56-63): the reference's filled parent array against our
fill_halo_regions! of the reference's own pre-fill array, BIT-EXACT.
Every mismatch names the "[recalled]" reading of SURVEY.md Appendix A it implicates.  Exit status 0 = everything within tolerance.
A dump whose generator starts with "glue" (julia/dump_reference.jl --glue: the same cases through julia/TripolarHIP.jl on an MI355X, the first
execution of that binding) was computed by THIS library: there every array must be bit-identical, and a difference implicates the binding.
No file of the reference is read or copied here: a dump is data (arrays the reference computed)."""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FORMAT = "tripolar-reference-dump-1"
ARRAY_NAMES = ("lambda_cc", "lambda_fc", "lambda_cf", "lambda_ff", "phi_cc", "phi_fc", "phi_cf", "phi_ff",
               "dx_cc", "dx_fc", "dx_cf", "dx_ff", "dy_cc", "dy_cf", "dy_fc", "dy_ff", "az_cc", "az_fc", "az_cf", "az_ff")
ELTYPES = {"Float64": np.dtype("<f8"), "Float32": np.dtype("<f4")}
METRIC_RTOL = 1e-12          # north_star: Float64 metrics within 1e-12 relative of the CPU reference
COORD_ATOL_DEG = 1e-11       # ~200 ulp of a longitude: two correctly-rounded libms differ by a few ulp per transcendental

# which "[recalled]" reading of SURVEY.md Appendix A a mismatch in (array group, region) points at
IMPLICATES = {
    ("coord", "interior"): "A-2 lambda tables (generate_coordinate: exact-rational / twice-precision range), A-3 phi range, A-4 degree trig "
                           "(sind / cosd / tand reduction), A-5 expression order of generate_tripolar_coordinates.jl:66-87; a difference of "
                           "a few 1e-14 deg is two libms rounding differently (Julia Base ports msun; DESIGN.md 2)",
    ("coord", "south_halo"): "C-1: south halos of the 8 coordinate arrays stay 0.0 (src/tripolar_grid.jl:148)",
    ("dx", "rows>=2"): "A-7 haversine form (Distances 0.10: sin^2 of half differences, asin(min(sqrt h, 1)), deg2rad = x * (pi / 180)); near the "
                       "poles the formula's conditioning allows ~3e-12 between independently rounded libms (DESIGN.md 2)",
    ("dy", "rows>=2"): "A-7 haversine form, as Dx",
    ("az_cc_ff", "rows>=2"): "A-8 lat_lon_to_cartesian / spherical_area_triangle (Eriksson) / quadrilateral association; individual cell areas "
                             "are conditioned ~eps / solid angle (1e-10 at 1/10 degree): compare the SUMS too",
    ("az_fc_cf", "rows>=2"): "products Dy * Dx of src/tripolar_grid_utils.jl:34-35: follow from Dx, Dy",
    ("metric", "rows<=1"): "A-8 lat-lon continuation [recalled LatitudeLongitudeGrid metrics]: association R * deg2rad(dlambda) * cos vs "
                           "R * cos * deg2rad(dlambda), hack_cosd / hack_sind, the pairing Dy_ff <- Dy_fc (C-2); only Dy is pinned by the README",
    ("dy_order",): "src/tripolar_grid.jl:321-324 passes Dy as (cc, cf, fc, ff): if the dump's Dy_fc equals OUR Dy_cf, Oceananigans' positional "
                   "parameters are (cc, fc, cf, ff) and the grid's field NAMES are swapped relative to the arrays' content (enum tpg_array "
                   "follows the call's position, include/tripolar_hip.h:62-69)",
    ("glue",): "a dump made through julia/TripolarHIP.jl must be BIT-identical to the same library driven from Python: a difference is the "
               "binding's marshalling -- the TpgParams layout / keyword forwarding (build_band), the order of the 20 output pointers against "
               "enum tpg_array, the OffsetArray offsets, or the device-to-host copy of HIPArray",
    ("field",): "zipper index / sign map (zipper_boundary_condition.jl:70-155), fill order zipper -> periodic x, the sign policy of "
                "tripolar_grid_extensions.jl:49-53 -- all pinned by the reference's own tests: a mismatch here is a bug, not a reading",
}


# ---------------------------------------------------------------------------------------------------------------------
# the dump format
# ---------------------------------------------------------------------------------------------------------------------
def read_array(case_dir, entry):
    """raw little-endian, first Julia index fastest -> numpy array of shape dims[::-1] ([k,] j, i), C order"""
    dt = ELTYPES[entry["eltype"]]
    a = np.fromfile(os.path.join(case_dir, entry["file"]), dtype=dt)
    dims = tuple(int(d) for d in entry["dims"])
    if a.size != int(np.prod(dims)):
        raise ValueError(f"{entry['file']}: {a.size} elements on disk, manifest says {dims}")
    return a.reshape(dims[::-1]).astype(dt.newbyteorder("="), copy=False)


def read_dump(dump_dir):
    with open(os.path.join(dump_dir, "manifest.json")) as f:
        man = json.load(f)
    if man.get("format") != FORMAT or man.get("endianness") != "little":
        raise ValueError(f"not a {FORMAT} dump: {man.get('format')!r} / {man.get('endianness')!r}")
    return man


def write_dump(dump_dir, cases, generator):
    """cases: [{"name", "kwargs", "eltype", "arrays": {name: ndarray [j, i]}, "fields": {name: {"after": ndarray [k, j, i], "before": ndarray,
    "location": [...], "sign": int, "initial": str}}}] -> the format dump_reference.jl writes (used by the self-test and by anyone who wants to
    exchange arrays without Julia)"""
    os.makedirs(dump_dir, exist_ok=True)
    out = []
    for c in cases:
        cdir = os.path.join(dump_dir, c["name"])
        os.makedirs(cdir, exist_ok=True)

        def put(name, a):
            et = {np.dtype("float64"): "Float64", np.dtype("float32"): "Float32"}[a.dtype]
            np.ascontiguousarray(a).astype(ELTYPES[et]).tofile(os.path.join(cdir, name + ".bin"))
            return {"file": name + ".bin", "dims": list(a.shape[::-1]), "eltype": et}
        arrays = {n: put(n, a) for n, a in c.get("arrays", {}).items()}
        fields = {}
        for n, fd in c.get("fields", {}).items():
            e = put(n, fd["after"])
            e["before"] = put(n + "_before", fd["before"])["file"]
            e.update(location=fd["location"], sign=fd["sign"], initial=fd.get("initial", ""))
            fields[n] = e
        out.append({"name": c["name"], "kwargs": c["kwargs"], "eltype": c["eltype"], "arrays": arrays, "fields": fields})
    with open(os.path.join(dump_dir, "manifest.json"), "w") as f:
        json.dump({"format": FORMAT, "endianness": "little", "generator": generator, "cases": out}, f, indent=1)


# ---------------------------------------------------------------------------------------------------------------------
# our two sides
# ---------------------------------------------------------------------------------------------------------------------
class OracleSide:
    name = "oracle"

    def __init__(self):
        from oracle import oracle
        self.o = oracle

    def grid(self, kw, eltype):
        return self.o.build_grid(tuple(kw["size"]), dtype=np.float64 if eltype == "Float64" else np.float32, halo=tuple(kw["halo"]),
                                 southernmost_latitude=kw["southernmost_latitude"], north_poles_latitude=kw["north_poles_latitude"],
                                 first_pole_longitude=kw["first_pole_longitude"], radius=kw["radius"])

    def fill(self, before, xl, yl, sign, size, halo):
        a = np.array(before, copy=True)
        self.o.fill_halo_regions(a, xl, yl, sign, size, halo)
        return a


class HipSide:
    name = "hip"

    def __init__(self):
        import torch
        import orthogonalsphericalshellgrids.jl_amd as osg
        if not torch.cuda.is_available():
            raise RuntimeError("no HIP device: use --oracle-only")
        self.torch, self.osg = torch, osg

    def grid(self, kw, eltype):
        t, osg = self.torch, self.osg
        g = osg.TripolarGrid(osg.GPU(0), t.float64 if eltype == "Float64" else t.float32, size=tuple(kw["size"]), halo=tuple(kw["halo"]),
                             southernmost_latitude=kw["southernmost_latitude"], north_poles_latitude=kw["north_poles_latitude"],
                             first_pole_longitude=kw["first_pole_longitude"], radius=kw["radius"])
        return {n: getattr(g, n).cpu().numpy() for n in ARRAY_NAMES}

    def fill(self, before, xl, yl, sign, size, halo):
        import ctypes as C
        t, osg = self.torch, self.osg
        d = t.from_numpy(np.array(before, copy=True)).to("cuda:0")
        lib = osg._lib.lib()
        ft = 1 if before.dtype == np.float64 else 0
        rc = lib.tpg_fill_halo_regions(osg._lib.ptr_table([d]), 1, (C.c_int8 * 1)(xl), (C.c_int8 * 1)(yl), (C.c_int32 * 1)(sign),
                                       *size, *halo, 1, ft, None)
        osg._lib.check(rc)
        t.cuda.synchronize()
        return d.cpu().numpy()


# ---------------------------------------------------------------------------------------------------------------------
# the comparison
# ---------------------------------------------------------------------------------------------------------------------
def _max_rel(got, ref):
    got, ref = got.astype(np.float64), ref.astype(np.float64)
    same = (got == ref) | (np.isnan(got) & np.isnan(ref))
    with np.errstate(divide="ignore", invalid="ignore"):
        rel = np.abs(got - ref) / np.abs(ref)
    rel[same] = 0.0
    rel[np.isnan(rel)] = np.inf                       # NaN on one side only
    return (float(rel.max()) if rel.size else 0.0), int((~same).sum())


def _max_abs_deg(got, ref, wrap):
    d = np.abs(got.astype(np.float64) - ref.astype(np.float64))
    if wrap:
        d = np.minimum(d, np.abs(360.0 - d))
    d[np.isnan(d)] = np.inf
    return (float(d.max()) if d.size else 0.0), int((~((got == ref) | (np.isnan(got) & np.isnan(ref)))).sum())


def compare_grid(case, case_dir, side, exact=False):
    """exact: the dump was made by THIS library through another binding (julia/dump_reference.jl --glue): 0 differing elements or a finding"""
    kw, et = case["kwargs"], case["eltype"]
    Hy = int(kw["halo"][1])
    ours = side.grid(kw, et)
    f32 = et == "Float32"
    rtol = 2e-6 if f32 else METRIC_RTOL               # Float32 grids: the Float64 pipeline rounded once (SURVEY A-1): 1-2 ulp of Float32
    atol = 2e-5 if f32 else COORD_ATOL_DEG
    rows, findings = [], []
    ref = {n: read_array(case_dir, e) for n, e in case["arrays"].items()}
    for n in ARRAY_NAMES:
        if n not in ref:
            continue
        g, r = ours[n], ref[n]
        if g.shape != r.shape:
            findings.append({"array": n, "problem": f"shape {r.shape} in the dump, {g.shape} here", "implicates": "halo / size keywords of the case"})
            continue
        south = slice(0, Hy)                           # parent rows of j = 1-Hy .. 0
        if n.startswith(("lambda", "phi")):
            e_main, n_main = _max_abs_deg(g[Hy:], r[Hy:], wrap=n.startswith("lambda"))
            e_south, n_south = _max_abs_deg(g[south], r[south], wrap=False)
            rows.append({"array": n, "region": "interior + north / x halos", "max_abs_deg": e_main, "differing": n_main, "ok": e_main <= atol})
            rows.append({"array": n, "region": "south halo rows", "max_abs_deg": e_south, "differing": n_south, "ok": e_south == 0.0})
            if e_main > atol:
                findings.append({"array": n, "region": "interior", "max_abs_deg": e_main, "implicates": IMPLICATES[("coord", "interior")]})
            if e_south != 0.0:
                findings.append({"array": n, "region": "south halo", "max_abs": e_south, "implicates": IMPLICATES[("coord", "south_halo")]})
        else:
            lo = slice(0, Hy + 1)                      # j = 1-Hy .. 1: continue_south! rows
            hi = slice(Hy + 1, None)                   # j >= 2
            e_hi, n_hi = _max_rel(g[hi], r[hi])
            e_lo, n_lo = _max_rel(g[lo], r[lo])
            rows.append({"array": n, "region": "rows j >= 2", "max_rel": e_hi, "differing": n_hi, "ok": e_hi <= rtol})
            rows.append({"array": n, "region": "rows j <= 1 (lat-lon continuation)", "max_rel": e_lo, "differing": n_lo, "ok": e_lo <= rtol})
            grp = "dx" if n.startswith("dx") else "dy" if n.startswith("dy") else ("az_cc_ff" if n in ("az_cc", "az_ff") else "az_fc_cf")
            if e_hi > rtol:
                findings.append({"array": n, "region": "rows j >= 2", "max_rel": e_hi, "implicates": IMPLICATES[(grp, "rows>=2")]})
            if e_lo > rtol:
                findings.append({"array": n, "region": "rows j <= 1", "max_rel": e_lo, "implicates": IMPLICATES[("metric", "rows<=1")]})
    # the Dy positional-order question: does the dump's Dy_fc hold what we call Dy_cf?
    dy_order = None
    if "dy_fc" in ref and "dy_cf" in ref and ours["dy_cf"].shape == ref["dy_fc"].shape:
        hi = slice(Hy + 1, None)
        straight = max(_max_rel(ours["dy_cf"][hi], ref["dy_cf"][hi])[0], _max_rel(ours["dy_fc"][hi], ref["dy_fc"][hi])[0])
        swapped = max(_max_rel(ours["dy_cf"][hi], ref["dy_fc"][hi])[0], _max_rel(ours["dy_fc"][hi], ref["dy_cf"][hi])[0])
        dy_order = {"by_name_max_rel": straight, "swapped_max_rel": swapped,
                    "reading": "names agree with content" if straight <= swapped else "SWAPPED: the grid's Dy_fc FIELD holds the Dy_cf array"}
        if swapped < straight:
            findings.append({"array": "dy_cf / dy_fc", "by_name_max_rel": straight, "swapped_max_rel": swapped, "implicates": IMPLICATES[("dy_order",)]})
    # tiling identity carried over from the oracle's own KAT: sums of the cell areas (insensitive to the per-cell conditioning)
    sums = {}
    for n in ("az_cc", "az_ff"):
        if n in ref:
            a, b = float(ours[n][Hy + 1:].astype(np.float64).sum()), float(ref[n][Hy + 1:].astype(np.float64).sum())
            sums[n] = {"ours": a, "reference": b, "rel": abs(a - b) / abs(b) if b else 0.0}
    if exact:
        for r in rows:
            r["ok"] = r["differing"] == 0
            if not r["ok"]:
                findings.append({"array": r["array"], "region": r["region"], "differing": r["differing"], "implicates": IMPLICATES[("glue",)]})
    return {"case": case["name"], "side": side.name, "eltype": et, "kwargs": kw, "arrays": rows, "dy_order": dy_order, "area_sums_rows_ge_2": sums,
            "findings": findings, "ok": all(r["ok"] for r in rows) and not findings}


LOC = {"Center": 0, "Face": 1, "Nothing": 0}


def compare_fields(case, case_dir, side):
    kw = case["kwargs"]
    rows, findings = [], []
    for n, e in case["fields"].items():
        after = read_array(case_dir, e)
        before = read_array(case_dir, dict(e, file=e["before"]))
        nz = after.shape[0]
        Hz = int(kw["halo"][2]) if nz > 1 or int(kw["size"][2]) + 2 * int(kw["halo"][2]) == nz else 0
        size = (int(kw["size"][0]), int(kw["size"][1]), nz - 2 * Hz)
        halo = (int(kw["halo"][0]), int(kw["halo"][1]), Hz)
        xl, yl = LOC[e["location"][0]], LOC[e["location"][1]]
        got = side.fill(before, xl, yl, int(e["sign"]), size, halo)
        # the reference's fill also runs Oceananigans' bottom / top pass (none set here) and leaves south halos alone: compare whole parents
        same = np.array_equal(got, after)
        nbad = int((got != after).sum())
        rows.append({"field": n, "location": e["location"], "sign": e["sign"], "bit_exact": same, "differing_cells": nbad, "ok": same})
        if not same:
            bad = np.argwhere(got != after)
            findings.append({"field": n, "differing_cells": nbad, "first_k_j_i": bad[0].tolist(), "implicates": IMPLICATES[("field",)]})
    return {"case": case["name"], "side": side.name, "fields": rows, "findings": findings, "ok": all(r["ok"] for r in rows)}


def compare_dump(dump_dir, sides):
    man = read_dump(dump_dir)
    exact = str(man.get("generator", "")).startswith("glue")          # julia/dump_reference.jl --glue: this library through the Julia binding
    reports = []
    for case in man["cases"]:
        cdir = os.path.join(dump_dir, case["name"])
        for side in sides:
            if case.get("arrays"):
                reports.append(compare_grid(case, cdir, side, exact))
            if case.get("fields"):
                reports.append(compare_fields(case, cdir, side))
    return {"dump": os.path.abspath(dump_dir), "generator": man.get("generator"), "bit_exact_required": exact,
            "metric_rtol": METRIC_RTOL, "coord_atol_deg": COORD_ATOL_DEG,
            "reports": reports, "ok": all(r["ok"] for r in reports)}


def print_report(rep, out=sys.stdout):
    print(f"dump: {rep['dump']}  (generator: {rep['generator']})", file=out)
    if rep.get("bit_exact_required"):
        print("a dump of THIS library through the Julia binding: every array must be bit-identical", file=out)
    for r in rep["reports"]:
        print(f"\n== {r['case']}  vs {r['side']}: {'OK' if r['ok'] else 'MISMATCH'}", file=out)
        for a in r.get("arrays", []):
            key = "max_rel" if "max_rel" in a else "max_abs_deg"
            print(f"   {a['array']:10s} {a['region']:36s} {key} {a[key]:.3e}  differing {a['differing']:8d}  {'ok' if a['ok'] else 'MISMATCH'}", file=out)
        if r.get("dy_order"):
            d = r["dy_order"]
            print(f"   Dy order: by name {d['by_name_max_rel']:.3e}, swapped {d['swapped_max_rel']:.3e} -> {d['reading']}", file=out)
        for n, s in (r.get("area_sums_rows_ge_2") or {}).items():
            print(f"   sum {n} (rows j >= 2): rel {s['rel']:.3e}", file=out)
        for f in r.get("fields", []):
            print(f"   field {f['field']:8s} {'/'.join(f['location']):24s} sign {f['sign']:+d}  {'bit-exact' if f['bit_exact'] else 'MISMATCH in %d cells' % f['differing_cells']}", file=out)
        for f in r["findings"]:
            print(f"   -> {f.get('array', f.get('field'))} [{f.get('region', '')}]: implicates {f['implicates']}", file=out)
    print("\nRESULT:", "every compared quantity within tolerance" if rep["ok"] else "MISMATCHES (see the readings named above)", file=out)


def main():
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("dump_dir")
    ap.add_argument("--oracle-only", action="store_true", help="no GPU: compare the dump with the oracle only")
    ap.add_argument("--hip-only", action="store_true")
    ap.add_argument("--json", help="write the full report there")
    args = ap.parse_args()
    sides = []
    if not args.hip_only:
        sides.append(OracleSide())
    if not args.oracle_only:
        sides.append(HipSide())
    rep = compare_dump(args.dump_dir, sides)
    print_report(rep)
    if args.json:
        with open(args.json, "w") as f:
            json.dump(rep, f, indent=1)
    sys.exit(0 if rep["ok"] else 1)


if __name__ == "__main__":
    main()
