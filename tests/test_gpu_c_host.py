"""The C ABI from a plain C99 host (examples/c_host.c: HIP C API + include/tripolar_hip.h, no Python / C++ / torch in the
process): compiled with gcc here, run on the GPU, its printed numbers checked against the reference's README transcript
(README.md:52-60) and zipper test (test/test_zipper_boundary_conditions.jl:38-45)."""
import math
import os
import re
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _sig6(x):
    return float(f"{x:.6g}")


def test_c_host_reproduces_the_reference_known_answers(gpu, kats, tmp_path):
    pkg = os.path.join(ROOT, "orthogonalsphericalshellgrids.jl_amd")
    exe = str(tmp_path / "c_host")
    cmd = ["gcc", "-std=c99", "-Wall", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "examples", "c_host.c"), "-L", pkg, "-ltripolar_hip", "-L/opt/rocm/lib", "-lamdhip64",
           f"-Wl,-rpath,{pkg}", "-Wl,-rpath,/opt/rocm/lib", "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout, r.stderr)
    out = r.stdout
    k = kats["readme_60x30"]
    m = re.search(r"center node \(lambda, phi\) = \(([-\d.e+]+), ([-\d.e+]+)\); dx_ff in \[([-\d.e+]+), ([-\d.e+]+)\] m; dy_ff in \[([-\d.e+]+), ([-\d.e+]+)\]", out)
    assert m, out
    lam, phi, dxmin, dxmax, dymin, dymax = (float(x) for x in m.groups())
    R = 6371.0e3
    assert lam == k["center_lambda_phi"][0] and round(phi, 4) == k["center_lambda_phi"][1]
    # the transcript prints spacings in degrees: rad2deg(metric / R), 6 significant digits (the C host prints 6 digits of metres)
    for got, key in ((dxmin, "min_dlambda"), (dxmax, "max_dlambda"), (dymin, "min_dphi"), (dymax, "max_dphi")):
        assert abs(math.degrees(got / R) - k[key]) <= 2e-6 * k[key], (key, got)
    z = kats["zipper_10x10"]["constant_one"]
    m = re.search(r"u\[2, Ny\+1\] = ([-\d.]+), u\[1, Ny\+1\] = ([-\d.]+), u\[Nx\+1, Ny\+1\] = ([-\d.]+), u\[Nx/2\+1, Ny\+4\] = ([-\d.]+)", out)
    assert m, out
    inner, left, right, mid = (float(x) for x in m.groups())
    assert inner == z["u_north_halo_i_2_to_Nx_minus_1"] and left == z["u_north_halo_i_1"] and right == z["u_north_halo_i_Nx_plus_1"]
    assert mid == -1.0
    assert "status -2" in out and "even" in out
    # the distributed entry point on a one-band chain is the serial fill: only the self-mapped pole point (sign -1) changes again
    m = re.search(r"distributed entry point, one band: (\d+) cells differ", out)
    assert m and int(m.group(1)) == 1, out
    m = re.search(r"pipelined entry point, one band: (\d+) cells differ", out)      # a third fill: the pole point is back where the first fill left it
    assert m and int(m.group(1)) == 0, out
