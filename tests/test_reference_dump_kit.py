"""The parity-pinning kit for a site with Julia: julia/dump_reference.jl runs the REFERENCE package and
writes its 20 grid arrays and filled fields as raw little-endian files + a JSON manifest; tests/compare_reference_dump.py compares such a dump
with the HIP path and with the oracle.  Julia cannot run here, so the kit is proven on what CAN run: the restatement-derived golden grids
(tests/golden/restatement_*.npz) and oracle-filled fields are written in the dump format and round-tripped through the comparator -- exact
agreement, the right regions and tolerances, the Dy-order question answered, and a planted mismatch named with the reading it implicates."""
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import compare_reference_dump as crd            # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
JL = os.path.join(ROOT, "orthogonalsphericalshellgrids.jl_amd", "julia", "dump_reference.jl")
KW = {"restatement_60x30_f64": dict(size=[60, 30, 1], halo=[4, 4, 4], north_poles_latitude=55.0, first_pole_longitude=70.0),
      "restatement_4x5_f32_poles75_35": dict(size=[4, 5, 1], halo=[4, 4, 4], north_poles_latitude=35.0, first_pole_longitude=75.0)}


def _cases(oracle):
    cases = []
    for name, kw in KW.items():
        z = np.load(os.path.join(GOLDEN, name + ".npz"))
        kw = dict(kw, southernmost_latitude=-80.0, radius=oracle.R_EARTH)
        cases.append({"name": name, "kwargs": kw, "eltype": "Float64" if z["dx_cc"].dtype == np.float64 else "Float32",
                      "arrays": {n: z[n] for n in crd.ARRAY_NAMES}})
    for h in (4, 5):                                # the field cases of test/test_zipper_boundary_conditions.jl, halo 4 and the model halo 5
        size, halo = (10, 10, 1), (h, h, h)
        rng = np.random.default_rng(h)
        fields = {}
        for fname, loc, sign in (("c_one", ["Center", "Center", "Center"], 1), ("u_x", ["Face", "Center", "Center"], -1), ("v_x", ["Center", "Face", "Center"], -1)):
            before = rng.uniform(-1, 1, (1 + 2 * h, 10 + 2 * h, 10 + 2 * h))
            after = before.copy()
            oracle.fill_halo_regions(after, crd.LOC[loc[0]], crd.LOC[loc[1]], sign, size, halo)
            fields[fname] = {"before": before, "after": after, "location": loc, "sign": sign, "initial": "random"}
        cases.append({"name": f"fields_10x10_halo{h}", "eltype": "Float64", "fields": fields,
                      "kwargs": dict(size=list(size), halo=list(halo), north_poles_latitude=55.0, first_pole_longitude=70.0, southernmost_latitude=-80.0, radius=oracle.R_EARTH)})
    return cases


def test_dump_format_round_trips_and_the_comparator_agrees_with_the_oracle(oracle, tmp_path):
    d = str(tmp_path / "dump")
    crd.write_dump(d, _cases(oracle), "self-test: tests/golden/restatement_*.npz + oracle-filled fields")
    man = crd.read_dump(d)
    assert man["format"] == "tripolar-reference-dump-1" and [c["name"] for c in man["cases"]][:2] == list(KW)
    # raw file = the parent array, first (i) index fastest: dims are Julia's (sx, rows)
    e = man["cases"][0]["arrays"]["lambda_cc"]
    assert e["dims"] == [68, 38] and e["eltype"] == "Float64" and os.path.getsize(os.path.join(d, man["cases"][0]["name"], e["file"])) == 68 * 38 * 8
    z = np.load(os.path.join(GOLDEN, "restatement_60x30_f64.npz"))
    assert np.array_equal(crd.read_array(os.path.join(d, "restatement_60x30_f64"), e), z["lambda_cc"])
    rep = crd.compare_dump(d, [crd.OracleSide()])
    assert rep["ok"], json.dumps([r for r in rep["reports"] if not r["ok"]])[:2000]
    g = [r for r in rep["reports"] if r["case"] == "restatement_60x30_f64"][0]
    assert len(g["arrays"]) == 40 and all(a["ok"] and a["differing"] == 0 for a in g["arrays"])      # golden == oracle, bit for bit
    assert {a["region"] for a in g["arrays"] if a["array"] == "dx_cc"} == {"rows j >= 2", "rows j <= 1 (lat-lon continuation)"}
    assert g["dy_order"]["reading"] == "names agree with content" and g["dy_order"]["by_name_max_rel"] == 0.0 and g["dy_order"]["swapped_max_rel"] > 1e-3
    assert all(s["rel"] == 0.0 for s in g["area_sums_rows_ge_2"].values())
    f = [r for r in rep["reports"] if r["case"] == "fields_10x10_halo5"][0]
    assert [x["bit_exact"] for x in f["fields"]] == [True, True, True]


def test_comparator_names_the_reading_a_mismatch_implicates(oracle, tmp_path):
    cases = _cases(oracle)
    g = cases[0]
    g["arrays"] = {n: a.copy() for n, a in g["arrays"].items()}
    g["arrays"]["dx_fc"][20, 30] *= 1 + 3e-12                                       # a haversine difference beyond 1e-12, row j = 17
    g["arrays"]["az_cc"][2, :] *= 1 + 1e-9                                           # a continuation row (j = -1)
    g["arrays"]["phi_ff"][0, 5] = 1.0                                                # a non-zero south halo
    g["arrays"]["dy_cf"], g["arrays"]["dy_fc"] = g["arrays"]["dy_fc"], g["arrays"]["dy_cf"]    # field names swapped relative to content
    cases[-1]["fields"]["u_x"]["after"][4, -1, 7] *= -1                               # one wrong sign in a north halo row
    d = str(tmp_path / "dump")
    crd.write_dump(d, cases, "self-test with planted mismatches")
    rep = crd.compare_dump(d, [crd.OracleSide()])
    assert not rep["ok"]
    g = [r for r in rep["reports"] if r["case"] == "restatement_60x30_f64"][0]
    by = {(f.get("array"), f.get("region")): f for f in g["findings"]}
    assert "haversine" in by[("dx_fc", "rows j >= 2")]["implicates"] and 2e-12 < by[("dx_fc", "rows j >= 2")]["max_rel"] < 4e-12
    assert "lat-lon continuation" in by[("az_cc", "rows j <= 1")]["implicates"]
    assert "stay 0.0" in by[("phi_ff", "south halo")]["implicates"]
    assert g["dy_order"]["reading"].startswith("SWAPPED") and "positional" in by[("dy_cf / dy_fc", None)]["implicates"]
    ok_elsewhere = [a for a in g["arrays"] if a["array"] in ("lambda_cc", "dx_cc", "az_ff")]
    assert all(a["ok"] for a in ok_elsewhere)
    f = [r for r in rep["reports"] if r["case"] == "fields_10x10_halo5"][0]
    assert [x["bit_exact"] for x in f["fields"]] == [True, False, True] and f["findings"][0]["differing_cells"] == 1
    # the command line: exit status 1, the readings in the text
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "compare_reference_dump.py"), d, "--oracle-only", "--json", str(tmp_path / "r.json")],
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 1 and "MISMATCH" in p.stdout and "haversine" in p.stdout and json.load(open(tmp_path / "r.json"))["ok"] is False


def test_dump_script_calls_only_the_exported_api_and_writes_this_format():
    """static (no Julia here): the script runs the reference package itself, names every one of the 20 struct fields, the reference's own test
    grids and the model halo (5, 5, 5), writes the format the comparator reads, and copies nothing from the reference's files"""
    src = open(JL).read()
    assert "using OrthogonalSphericalShellGrids" in src and "TripolarGrid(dump_arch(), case.FT; case.kwargs...)" in src and "fill_halo_regions!(f)" in src
    assert "dump_arch() = GLUE ? HIPGPU() : CPU()" in src
    assert '"format\\": \\"tripolar-reference-dump-1\\"' in src and "htol.(p)" in src and "parent(a)" in src
    names = re.findall(r'"([a-z]+_[cf][cf])" => :', src)
    assert sorted(names) == sorted(crd.ARRAY_NAMES)
    for needle in ("size = (4, 5, 1), first_pole_longitude = 75, north_poles_latitude = 35", "size = (10, 10, 1)", "size = (60, 30, 1)",
                   "size = (360, 180, 1)", "halo = (5, 5, 5)", "Field{LX, LY, Center}(grid)", '("u_one", Face, Center, 1', '("v_one", Center, Face, 1'):
        assert needle in src, needle
    code = re.sub(r"#.*", "", src)
    # the package's API only, no test file pulled in; the one include is this repository's own binding, in --glue mode
    assert re.findall(r"include\(([^\n]*)\)", code) == ['joinpath(@__DIR__, "TripolarHIP.jl")'] and "@testset" not in code and "ccall" not in code
    # --glue: the same cases through julia/TripolarHIP.jl on HIPGPU(); reduced fields (no bottom / top condition), pre-fill state written
    # into the parent, a generator string the comparator recognises (bit-exact mode)
    assert 'const GLUE = "--glue" in ARGS' in src and "Field{LX, LY, Nothing}(grid)" in src and 'GLUE ? "glue: julia' in src
    assert "GLUE || return set!(f, init)" in src
    # blocks balance (as for the glue: a coarse syntax check)
    openers = len(re.findall(r"(?m)^\s*(?:function|for|if|let|begin)\b(?!.*\bend\s*$)", code)) + len(re.findall(r"\bdo\s*(?:\w+\s*)?$", code, flags=re.M))
    assert openers == len(re.findall(r"(?m)^\s*end\b", code)), openers


def test_a_glue_dump_must_be_bit_identical(oracle, tmp_path):
    """A dump whose generator starts with "glue" was computed by this library through the Julia binding: the comparator then demands 0
    differing elements (a one-ulp difference, fine against the reference, is a finding that names the binding) -- and passes an exact dump."""
    cases = _cases(oracle)
    d = str(tmp_path / "exact")
    crd.write_dump(d, cases, "glue: self-test")
    rep = crd.compare_dump(d, [crd.OracleSide()])
    assert rep["bit_exact_required"] and rep["ok"], json.dumps([r for r in rep["reports"] if not r["ok"]])[:2000]
    g = cases[0]
    g["arrays"] = {n: a.copy() for n, a in g["arrays"].items()}
    g["arrays"]["dx_cc"][20, 30] = np.nextafter(g["arrays"]["dx_cc"][20, 30], np.inf)       # one ulp: within 1e-12, not bit-identical
    d2 = str(tmp_path / "one_ulp")
    crd.write_dump(d2, cases, "glue: self-test with one ulp planted")
    rep = crd.compare_dump(d2, [crd.OracleSide()])
    assert not rep["ok"]
    bad = [f for r in rep["reports"] for f in r["findings"]]
    assert len(bad) == 1 and bad[0]["array"] == "dx_cc" and bad[0]["differing"] == 1 and "marshalling" in bad[0]["implicates"]
    crd.write_dump(d2, cases, "julia 1.10, OrthogonalSphericalShellGrids 0.2.1")             # the same arrays as a REFERENCE dump: within tolerance
    assert crd.compare_dump(d2, [crd.OracleSide()])["ok"]


@pytest.mark.gpu
def test_comparator_against_the_hip_path(oracle, gpu, tmp_path):
    """the same round trip with the product library as the second side: the golden grids and the oracle-filled fields agree with the HIP path"""
    d = str(tmp_path / "dump")
    crd.write_dump(d, _cases(oracle), "self-test")
    rep = crd.compare_dump(d, [crd.HipSide()])
    assert rep["ok"], json.dumps([r for r in rep["reports"] if not r["ok"]])[:2000]
    assert all(a["differing"] == 0 for r in rep["reports"] for a in r.get("arrays", []))
