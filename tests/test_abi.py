"""The C-ABI shared library loads without a GPU and exports every symbol include/tripolar_hip.h
declares; argument validation (which precedes any device work) mirrors the reference's
ArgumentErrors.  No compute call is made here."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols(header="tripolar_hip.h"):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tpg_[a-z0-9_]+)\s*\(", text)))


def exported_symbols(path):
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
    return sorted(l.split()[-1] for l in out.splitlines() if l.strip())


TEST_ONLY = ["tpg_fill_synthetic", "tpg_math_probe", "tpg_reload_config", "tpg_zipper_copy_probe"]


def test_header_symbols_are_all_exported_and_bound(osg):
    lib = osg._lib.lib()
    names = declared_symbols()
    assert len(names) >= 12
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/tripolar_hip.h but not exported"
        assert n in osg._lib.SIGNATURES, f"{n} has no ctypes signature"
    assert sorted(osg._lib.SIGNATURES) == names


PROFILING_API = ["tpg_event_create", "tpg_event_destroy", "tpg_event_elapsed_ms", "tpg_fill_halo_regions_timed", "tpg_zipper_fill_timed"]


def test_product_library_exports_exactly_the_header(osg):
    """libtripolar_hip.so exports the symbols of include/tripolar_hip.h and NOTHING else: no test / bench hook, no C++ symbol,
    no knob reload; the hooks live in tools/libtripolar_hip_test.so (include/tripolar_hip_test.h) = product symbols + 4.
    Of the header's symbols five are its "profiling API" (kernel-event timing, no reference counterpart) and say so."""
    product = exported_symbols(osg._lib.LIB_PATH)
    assert product == declared_symbols()
    header = open(os.path.join(ROOT, "include", "tripolar_hip.h")).read()
    assert "profiling API (NO reference counterpart" in header and all(n in product for n in PROFILING_API)
    assert not set(product) & set(TEST_ONLY)
    from tools import testlib
    assert declared_symbols("tripolar_hip_test.h") == TEST_ONLY
    assert exported_symbols(testlib.LIB_PATH) == sorted(product + TEST_ONLY)
    assert sorted(testlib.TEST_SIGNATURES) == TEST_ONLY
    # the product library never touches the environment: no getenv import
    import subprocess
    und = subprocess.run(["nm", "-D", "--undefined-only", osg._lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    assert "getenv" not in und
    und_t = subprocess.run(["nm", "-D", "--undefined-only", testlib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    assert "getenv" in und_t


def test_version_and_status_strings(osg):
    lib = osg._lib.lib()
    assert lib.tpg_version() == 600
    assert b"even" in lib.tpg_status_string(-2)
    assert lib.tpg_status_string(0) == b"ok"


def _params(osg, **kw):
    d = dict(Nx=60, Ny=30, Nz=1, Hx=4, Hy=4, Hz=4, south=-80.0, npl=55.0, fpl=70.0, R=6371e3, ft=1, jstart=1, jend=30)
    d.update(kw)
    return osg._lib.TpgParams(d["Nx"], d["Ny"], d["Nz"], d["Hx"], d["Hy"], d["Hz"], d["south"], d["npl"], d["fpl"],
                              d["R"], d["ft"], d["jstart"], d["jend"], 0)


def test_build_grid_argument_errors_without_device_work(osg):
    lib = osg._lib.lib()
    out = (C.c_void_p * 20)(*([1 << 20] * 20))          # never dereferenced: validation fails first
    p = _params(osg, Nx=61)
    assert lib.tpg_build_grid(C.byref(p), out, None, 0, None) == -2          # odd Nlambda (tripolar_grid.jl:81-83)
    assert b"even" in lib.tpg_last_error()
    p = _params(osg, jstart=5, jend=31)
    assert lib.tpg_build_grid(C.byref(p), out, None, 0, None) == -3          # band outside the grid
    p = _params(osg, ft=7)
    assert lib.tpg_build_grid(C.byref(p), out, None, 0, None) == -1
    p = _params(osg)
    assert lib.tpg_build_grid(C.byref(p), out, None, 0, None) == -4          # no workspace
    assert lib.tpg_build_grid_workspace_bytes(C.byref(p)) >= 8 * (4 * 60 + 4 * 30 + 5 * 5)
    p = _params(osg, Hy=40)
    assert lib.tpg_build_grid(C.byref(p), out, None, 0, None) == -5          # halo larger than the grid
    p = _params(osg)
    p.reserved = 6                                                            # only TPG_BUILD_TABLES_VALID (1) is a known flag
    assert lib.tpg_build_grid(C.byref(p), out, None, 0, None) == -1 and b"unknown flag" in lib.tpg_last_error()


def test_zipper_argument_errors_without_device_work(osg):
    lib = osg._lib.lib()
    fields = (C.c_void_p * 1)(1 << 20)
    xl, yl, sg = (C.c_int8 * 1)(0), (C.c_int8 * 1)(0), (C.c_int32 * 1)(1)
    assert lib.tpg_zipper_fill(fields, 1, xl, yl, sg, 11, 10, 1, 4, 4, 4, 1, 1, 1, None) == -2
    xl2 = (C.c_int8 * 1)(3)
    assert lib.tpg_zipper_fill(fields, 1, xl2, yl, sg, 10, 10, 1, 4, 4, 4, 1, 1, 1, None) == -1   # no method for this location
    assert lib.tpg_zipper_fill(fields, 1, xl, yl, sg, 10, 10, 1, 4, 4, 4, 1, 9, 1, None) == -1    # levels outside the parent
    assert lib.tpg_zipper_fill(None, 0, xl, yl, sg, 10, 10, 1, 4, 4, 4, 1, 1, 1, None) == -1
    assert lib.tpg_y_halo_buffer_elems(4, 3600, 75, 4, 4, 4) == 4 * 3608 * 4 * 83


def test_distributed_fill_argument_errors_without_device_work(osg):
    lib = osg._lib.lib()
    fields = (C.c_void_p * 1)(1 << 20)
    xl, yl, sg = (C.c_int8 * 1)(0), (C.c_int8 * 1)(0), (C.c_int32 * 1)(1)
    geom = (10, 10, 1, 4, 4, 4)
    assert lib.tpg_fill_halo_regions_distributed(None, 2, 2, fields, 1, xl, yl, sg, None, None, None, None, *geom, 1, None) == -3
    assert lib.tpg_fill_halo_regions_distributed(None, -1, 2, fields, 1, xl, yl, sg, None, None, None, None, *geom, 1, None) == -3
    # the north side is the zipper OR a seam
    assert lib.tpg_fill_halo_regions_distributed_peers(None, -1, 1, 1, fields, 1, xl, yl, sg, None, None, None, None, *geom, 1, None) == -1
    assert b"zipper or a seam" in lib.tpg_last_error()
    # the pipelined forms validate the same way, and additionally refuse a missing message buffer / a negative stage size
    assert lib.tpg_fill_halo_regions_distributed_pipelined(None, 2, 2, fields, 1, xl, yl, sg, None, None, None, None, *geom, 1, None, None, 1) == -3
    assert lib.tpg_fill_halo_regions_distributed_pipelined_peers(None, -1, 1, 1, fields, 1, xl, yl, sg, None, None, None, None, *geom, 1,
                                                                 None, None, 1) == -1
    assert lib.tpg_halo_exchange_y_pipelined(1 << 20, 0, 2, fields, 1, None, None, None, None, *geom, 1, None, None, 1) == -1   # pack-free: not offered
    assert b"message buffer" in lib.tpg_last_error()
    assert lib.tpg_halo_exchange_y_pipelined(1 << 20, 0, 2, fields, 1, 1 << 20, 1 << 20, 1 << 20, 1 << 20, *geom, 1, None, None, -2) == -1
    assert b"fields_per_stage" in lib.tpg_last_error()
    assert lib.tpg_halo_exchange_y_pipelined(None, 0, 2, fields, 1, None, None, None, None, *geom, 1, None, None, 1) == -1          # null communicator
    # argument validation of the local fill comes first (odd Nx), before any communicator is looked at
    assert lib.tpg_fill_halo_regions_distributed(None, 0, 2, fields, 1, xl, yl, sg, None, None, None, None, 11, 10, 1, 4, 4, 4, 1, None) == -2
    assert lib.tpg_comm_available() in (0, -7)


def test_product_has_no_cpu_path(osg):
    """without a HIP device the host API must fail loudly instead of computing on the CPU"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(RuntimeError, match="no HIP device"):
        osg.TripolarGrid(size=(60, 30, 1))
    with pytest.raises(RuntimeError, match="HIP-only"):
        osg.TripolarGrid(osg.CPU(), size=(60, 30, 1))


def test_missing_extension_fails_loudly(osg, monkeypatch, tmp_path):
    """no silent fallback: without libtripolar_hip.so the binding raises"""
    monkeypatch.setattr(osg._lib, "_lib", None)
    monkeypatch.setattr(osg._lib, "LIB_PATH", str(tmp_path / "libtripolar_hip.so"))
    with pytest.raises(ImportError, match="only backend"):
        osg._lib.lib()


def test_product_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "orthogonalsphericalshellgrids.jl_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".jl")):
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert "oracle" not in text.replace("oracle/ as test infrastructure", ""), os.path.join(dirpath, f)


def test_header_is_plain_c(tmp_path):
    """include/tripolar_hip.h (and the test header on top of it) must compile as C99 with no C++ or HIP types (the boundary a Julia ccall / cgo / JNI stub binds)"""
    import subprocess
    names = declared_symbols() + declared_symbols("tripolar_hip_test.h")
    src = tmp_path / "abi.c"
    src.write_text('#include "tripolar_hip_test.h"\n'
                   + "typedef void (*fn)(void);\nstatic fn table[] = {" + ", ".join(f"(fn){n}" for n in names) + "};\n"
                   + "int main(void) { tpg_params p = {0}; (void)p; return sizeof(table) == 0 || TPG_NUM_ARRAYS != 20 || TPG_COMM_ID_BYTES != 128; }\n")
    r = subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-c", str(src),
                        "-o", str(tmp_path / "abi.o")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_only_the_checker_sites_load_the_oracle():
    """oracle/ is test infrastructure: besides tests/ (incl. tests/soak/), only __graft_entry__.smoke() and bench.py's cpu_baseline()
    may import it -- tools/ and examples/ may not, and bench.py may not outside that one function"""
    import ast
    for sub in ("tools", "examples"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, sub)):
            for f in files:
                if f.endswith((".py", ".sh", ".c", ".hip")):
                    text = open(os.path.join(dirpath, f), errors="replace").read()
                    assert not re.search(r"^\s*(from oracle|import oracle)", text, flags=re.M), os.path.join(dirpath, f)
    for name, allowed in (("bench.py", {"cpu_baseline"}), ("__graft_entry__.py", {"smoke"})):
        tree = ast.parse(open(os.path.join(ROOT, name)).read())
        for node in ast.walk(tree):
            if isinstance(node, ast.FunctionDef):
                imports = [n for n in ast.walk(node) if isinstance(n, (ast.Import, ast.ImportFrom))
                           and ("oracle" in (getattr(n, "module", None) or "") or any("oracle" in a.name for a in n.names))]
                assert not imports or node.name in allowed, (name, node.name)
        top = [n for n in tree.body if isinstance(n, (ast.Import, ast.ImportFrom))
               and ("oracle" in (getattr(n, "module", None) or "") or any("oracle" in a.name for a in n.names))]
        assert not top, name


def test_product_library_does_not_know_the_test_double():
    """the knob that selects the double lives in the test library only"""
    import subprocess
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    prod = os.path.join(ROOT, "orthogonalsphericalshellgrids.jl_amd", "libtripolar_hip.so")
    strings = subprocess.run(["strings", prod], capture_output=True, text=True).stdout
    assert "TPG_RCCL_LIBRARY" not in strings and "nccl_shim" not in strings
    assert "TPG_RCCL_LIBRARY" in subprocess.run(["strings", os.path.join(ROOT, "tools", "libtripolar_hip_test.so")], capture_output=True, text=True).stdout


def test_cell_kernel_stores_are_all_streaming(tmp_path):
    """k_cells_tile writes 1 GB at 1/10 degree; every store of it must carry the non-temporal hint, or the arrays stay behind as dirty lines
    in L2 / Infinity Cache and the NEXT kernel on the stream pays for their eviction (the halo fill of a bench step: 67 instead of 47 us).
    A first form of round 6's halo-push variant lost the hint on 11 of the 21 stores of the ordinary path without any change to them: the
    compiler sinks the stores of two branches (edge tile / ordinary tile) into one instruction and keeps a hint only if BOTH had it.  The
    generated ISA is what counts, so it is checked: the streaming instantiations of both cell kernels have no plain global store."""
    import subprocess
    csrc = os.path.join(ROOT, "orthogonalsphericalshellgrids.jl_amd", "csrc")
    asm = tmp_path / "tpg_grid.s"
    # compiled as the TEST library compiles it (-DTPG_TEST_ABI): the product's kernel k_cells_tile AND the halo-push variant k_cells_tile_push
    r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-DTPG_TEST_ABI",
                        "-S", "--cuda-device-only", "-o", str(asm), os.path.join(csrc, "tpg_grid.hip")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    text = asm.read_text()
    found = 0
    for kernel, nmin in (("12k_cells_tile", 21), ("17k_cells_tile_push", 100)):
        for T in ("d", "f"):                                           # <double, true, 8>, <float, true, 8>: the streaming instantiations
            m = re.search(r"^(_ZN12_GLOBAL__N_1%sI%sLb1ELi8E\w*):.*?^\.Lfunc_end" % (kernel, T), text, flags=re.S | re.M)
            assert m, "%s<%s, true, 8> not found in the ISA" % (kernel, T)
            stores = [l for l in m.group(0).splitlines() if "global_store" in l]
            plain = [l.strip() for l in stores if not re.search(r"\bnt\b", l)]
            assert len(stores) >= nmin and not plain, (kernel, T, len(stores), plain[:5])
            found += 1
    assert found == 4
