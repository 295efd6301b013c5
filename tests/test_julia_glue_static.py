"""julia/TripolarHIP.jl cannot be run here (no Julia toolchain), so its `ccall`s are checked statically against the C header:
every called symbol is declared in include/tripolar_hip.h, the argument-type tuple has as many entries as the C prototype has
parameters, each Julia type is compatible with the C type in that position, the return type matches, and as many values are
passed as types are declared.  This catches the class of error nothing else here could: a ccall that drifted from the ABI."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
JL = os.path.join(ROOT, "orthogonalsphericalshellgrids.jl_amd", "julia", "TripolarHIP.jl")


def split_top(s):
    """split at top-level commas (parentheses, brackets and braces nest)"""
    parts, depth, cur = [], 0, []
    for ch in s:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            parts.append("".join(cur).strip()); cur = []
        else:
            cur.append(ch)
    if "".join(cur).strip():
        parts.append("".join(cur).strip())
    return parts


def balanced(text, start):
    """index just past the parenthesis group that opens at text[start] == '('"""
    depth = 0
    for i in range(start, len(text)):
        if text[i] == "(":
            depth += 1
        elif text[i] == ")":
            depth -= 1
            if depth == 0:
                return i + 1
    raise ValueError("unbalanced")


def c_prototypes():
    text = open(os.path.join(ROOT, "include", "tripolar_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    protos = {}
    for m in re.finditer(r"([A-Za-z_][\w \*]*?)\b(tpg_[a-z0-9_]+)\s*\(", text):
        end = balanced(text, m.end() - 1)
        params = text[m.end():end - 1].strip()
        plist = [] if params in ("void", "") else split_top(" ".join(params.split()))
        protos[m.group(2)] = (" ".join(m.group(1).split()), plist)
    return protos


def compatible(jl, c):
    c = c.replace("const ", "").strip()
    is_ptr = "*" in c or "[" in c
    if jl in ("Cint", "Int32"):
        return not is_ptr and re.match(r"(int|int32_t)\b", c) is not None
    if jl == "Csize_t":
        return not is_ptr and c.startswith("size_t")
    if jl == "Ptr{Ptr{Cvoid}}":
        return bool(re.match(r"void \*\s*\w+\[", c) or re.match(r"void \*\*", c) or "void *const" in c)
    if jl in ("Ptr{Cvoid}", "Ptr{UInt8}"):
        return is_ptr and (c.startswith("void") or c.startswith("uint8_t") or c.startswith("double"))
    if jl == "Ref{Ptr{Cvoid}}":
        return c.startswith("void **")
    if jl == "Ptr{Int8}":
        return is_ptr and c.startswith("int8_t")
    if jl == "Ptr{Int32}":
        return is_ptr and c.startswith("int32_t")
    if jl == "Ref{TpgParams}":
        return is_ptr and c.startswith("tpg_params")
    return False


def test_every_ccall_matches_the_c_prototype():
    src = open(JL).read()
    protos = c_prototypes()
    calls = []
    for m in re.finditer(r"ccall\(\(:(tpg_[a-z0-9_]+),\s*libtripolar\)", src):
        end = balanced(src, src.index("(", m.start()))
        args = split_top(src[src.index("(", m.start()) + 1:end - 1])
        calls.append((m.group(1), args))
    assert len(calls) >= 14
    seen = set()
    for name, args in calls:
        assert name in protos, f"{name}: ccall of a symbol include/tripolar_hip.h does not declare"
        ret_c, params = protos[name]
        ret_jl, types = args[1], args[2]
        assert types.startswith("(") and types.endswith(")")
        tlist = split_top(types[1:-1])
        tlist = [t for t in tlist if t]                    # "(Ptr{UInt8},)" -> one entry
        values = args[3:]
        assert len(tlist) == len(params), f"{name}: {len(tlist)} Julia argument types for {len(params)} C parameters"
        assert len(values) == len(tlist), f"{name}: {len(values)} values passed for {len(tlist)} declared types"
        for pos, (jt, ct) in enumerate(zip(tlist, params)):
            assert compatible(jt, ct), f"{name}: argument {pos + 1}: Julia {jt} vs C `{ct}`"
        want_ret = {"int": "Cint", "size_t": "Csize_t", "const char *": "Cstring"}[ret_c.replace("extern ", "")]
        assert ret_jl == want_ret, f"{name}: returns {ret_jl}, C says {ret_c}"
        seen.add(name)
    # the entry points a Julia host needs for the whole path are all bound
    for needed in ("tpg_build_grid", "tpg_build_grid_workspace_bytes", "tpg_zipper_fill", "tpg_periodic_x_fill", "tpg_fill_halo_regions",
                   "tpg_fill_halo_regions_distributed", "tpg_halo_exchange_y", "tpg_fill_halo_regions_distributed_pipelined",
                   "tpg_halo_exchange_y_pipelined", "tpg_comm_available", "tpg_comm_unique_id",
                   "tpg_comm_init_rank", "tpg_comm_destroy", "tpg_y_halo_buffer_elems", "tpg_nonorthogonality_angle",
                   "tpg_convert_frame", "tpg_last_error"):
        assert needed in seen, f"{needed} is never ccall'ed from TripolarHIP.jl"


def test_integer_arguments_of_every_ccall_sit_under_the_right_parameter_name():
    """Types cannot tell Ny from Nz or rank from nranks: for every C parameter with one of these NAMES the Julia call must pass the value of
    the same meaning at that position (a swapped pair of Cints would compile, run and fill the wrong rows)."""
    src = open(JL).read()
    protos = c_prototypes()
    expect = {"Nx": {"Nx", "size(grid, 1)"}, "Ny": {"Ny"}, "Nz": {"Nz"}, "Hx": {"Hx", "halo_size(grid)[1]"}, "Hy": {"Hy", "halo_size(grid)[2]"},
              "Hz": {"Hz"}, "rank": {"comm.rank", "rank"}, "nranks": {"comm.nranks", "nranks"}, "nfields": {"length(fs)", "nfields"},
              "kstart": {"1"}, "kend": {"Nz"}, "north_is_zipper": {"1"}, "fields_per_stage": {"fps"},
              "xloc": {"xloc"}, "yloc": {"yloc"}, "sign": {"sgn"}, "to_native": {"to_native ? 1 : 0"},
              "send_south": {"bp[1]"}, "send_north": {"bp[2]"}, "recv_south": {"bp[3]"}, "recv_north": {"bp[4]"}}
    checked = 0
    for m in re.finditer(r"ccall\(\(:(tpg_[a-z0-9_]+),\s*libtripolar\)", src):
        end = balanced(src, src.index("(", m.start()))
        args = split_top(src[src.index("(", m.start()) + 1:end - 1])
        values = [" ".join(v.split()) for v in args[3:]]
        _, params = protos[m.group(1)]
        for pos, (ct, val) in enumerate(zip(params, values)):
            pname = re.sub(r"\[\]", "", ct.split()[-1]).lstrip("*")
            if pname in expect:
                assert val in expect[pname], f"{m.group(1)}: parameter {pos + 1} `{pname}` receives `{val}`"
                checked += 1
    assert checked >= 100, checked


def test_struct_layout_matches_tpg_params():
    """the Julia mirror of struct tpg_params lists the same fields with the same widths, in the same order"""
    src = open(JL).read()
    body = src[src.index("struct TpgParams"):src.index("\nend", src.index("struct TpgParams"))]
    fields = re.findall(r"(\w+)::(Int32|Float64)", body)
    hdr = open(os.path.join(ROOT, "include", "tripolar_hip.h")).read()
    cbody = hdr[hdr.index("typedef struct tpg_params {"):hdr.index("} tpg_params;")]
    cbody = re.sub(r"/\*.*?\*/", "", cbody, flags=re.S)
    cfields = []
    for ctype, names in re.findall(r"(int32_t|double)\s+([\w, ]+);", cbody):
        for n in names.split(","):
            cfields.append((n.strip(), {"int32_t": "Int32", "double": "Float64"}[ctype]))
    assert fields == cfields, (fields, cfields)


def _call_arguments(src, start):
    """the top-level comma-separated arguments of the call whose opening parenthesis is at src[start]"""
    depth, args, cur = 0, [], ""
    for ch in src[start:]:
        if ch in "([{":
            depth += 1
            if depth == 1:
                continue
        elif ch in ")]}":
            depth -= 1
            if depth == 0:
                args.append(cur.strip())
                return args
        if ch == "," and depth == 1:
            args.append(cur.strip()); cur = ""
        else:
            cur += ch
    raise AssertionError("unbalanced call")


def test_every_tpg_params_constructor_call_is_positionally_right():
    """Julia's default struct constructor is positional: every `TpgParams(...)` in the glue must pass one value per field of the struct, the
    geometry and mapping parameters in the header's order (a swapped pair of Float64 latitudes would compile and build a wrong grid)."""
    src = open(JL).read()
    body = src[src.index("struct TpgParams"):src.index("\nend", src.index("struct TpgParams"))]
    nfields = len(re.findall(r"(\w+)::(Int32|Float64)", body))
    calls = [m.end() - 1 for m in re.finditer(r"(?<!struct )\bTpgParams\(", src)]
    assert len(calls) >= 2
    want = ["Nλ", "Nφ", "Nz", "Hλ", "Hφ", "Hz", "southernmost_latitude", "north_poles_latitude", "first_pole_longitude", "radius", "ft_code(FT)",
            "jstart", "jend"]
    for c in calls:
        args = _call_arguments(src, c)
        assert len(args) == nfields == 14, (len(args), args)
        assert args[:13] == want, args
        assert args[13] in ("0", "reuse ? TPG_BUILD_TABLES_VALID : Int32(0)"), args[13]


def test_the_twenty_arrays_are_taken_in_the_order_of_enum_tpg_array():
    """build_band hands tpg_build_grid 20 pointers and names them by POSITION: the destructuring order must be enum tpg_array's (header), and
    the grid constructor must receive them in that same order with z between the coordinates and the metrics (src/tripolar_grid.jl:304-330)."""
    src = open(JL).read()
    hdr = open(os.path.join(ROOT, "include", "tripolar_hip.h")).read()
    enum = hdr[hdr.index("enum tpg_array {"):hdr.index("TPG_NUM_ARRAYS")]
    names = re.findall(r"TPG_(LAMBDA|PHI|DX|DY|AZ)_([CF][CF])", enum)
    assert len(names) == 20
    greek = {"LAMBDA": "λ", "PHI": "φ", "DX": "Δx", "DY": "Δy", "AZ": "Az"}
    want = [greek[k] + loc.lower() for k, loc in names]
    lhs = src[src.index("λcc, λfc"):src.index("= off.(arrays)")]
    got = [t.strip() for t in lhs.replace("\n", " ").split(",") if t.strip()]
    assert got == want, (got, want)
    ctor = src[src.index("return OrthogonalSphericalShellGrid{Periodic, LY, Bounded}("):]
    args = _call_arguments(ctor, ctor.index("("))
    assert args[8:16] == want[:8] and args[16] == "on_architecture(arch, zc)" and args[17:29] == want[8:], args
    assert "arrays = [device_array(arch, FT, Nλ + 2Hλ, ny + 2Hφ) for _ in 1:20]" in src and "ptrs = Ptr{Cvoid}[device_pointer(a) for a in arrays]" in src


def test_sign_policy_of_the_glue_is_the_reference_table():
    """src/tripolar_grid_extensions.jl:49-53: +1 everywhere except the two velocity locations; and `regularize_field_boundary_conditions`
    signs a field by its NAME (:u, :v -> -1), src/tripolar_grid_extensions.jl:32 -- the same table the Python host's tests execute"""
    src = re.sub(r"#.*", "", open(JL).read())
    methods = dict(re.findall(r"(?m)^sign\(([^)]*)\)\s*=\s*(-?\s*1)\s*$", src))
    assert {k.replace(" ", ""): int(v.replace(" ", "")) for k, v in methods.items()} == \
        {"LX,LY": 1, "::Type{Face},::Type{Center}": -1, "::Type{Center},::Type{Face}": -1}
    assert "sgn = field_name == :u || field_name == :v ? -1 : 1" in src
    assert src.count("ZipperBoundaryCondition(sign(LX, LY))") == 2           # serial and distributed Field constructors


def test_blocks_are_balanced():
    """a coarse syntax check: every block opener of the file has its `end` (strings and comments stripped)"""
    src = open(JL).read()
    src = re.sub(r'"""(?:.|\n)*?"""', '""', src)
    src = re.sub(r'"(?:\\.|[^"\\\n])*"', '""', src)
    src = re.sub(r"#.*", "", src)
    openers = len(re.findall(r"(?m)^\s*(?:mutable struct|struct|function|module|if|for|let|begin|@static if)\b(?!.*\bend\s*$)", src))
    openers += len(re.findall(r"\bdo\s*(?:\w+\s*)?$", src, flags=re.M)) + len(re.findall(r"GC\.@preserve[^\n]*\bbegin\b", src))
    ends = len(re.findall(r"(?m)^\s*end\b", src))
    assert openers == ends, (openers, ends)
    assert src.count("(") == src.count(")") and src.count("[") == src.count("]") and src.count("{") == src.count("}")


HIP_HEADER = "/opt/rocm/include/hip/hip_runtime_api.h"


def hip_prototypes(names):
    text = open(HIP_HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//.*", "", text)
    protos = {}
    for n in names:
        m = re.search(r"(?m)^((?:const )?[\w\*]+[ \*]+)" + n + r"\s*\(", text)
        assert m, f"{n} not declared in {HIP_HEADER}"
        end = balanced(text, m.end() - 1)
        params = " ".join(text[m.end():end - 1].split())
        protos[n] = (" ".join(m.group(1).split()), [] if params in ("void", "") else split_top(params))
    return protos


def hip_compatible(jl, c):
    c = c.replace("const ", "").strip()
    if jl == "Ref{Ptr{Cvoid}}":                       # an out-parameter that receives a pointer / an opaque handle
        return c.startswith("void**") or c.startswith("void **") or c.startswith("hipStream_t*") or c.startswith("hipStream_t *")
    if jl == "Ptr{Cvoid}":
        return c.startswith("void*") or c.startswith("void *") or re.match(r"hipStream_t\b(?!\s*\*)", c) is not None
    if jl == "Csize_t":
        return c.startswith("size_t")
    if jl == "Cint":
        return re.match(r"(int|hipMemcpyKind|hipError_t)\b(?!\s*\*)", c) is not None      # C enums are int-sized
    if jl == "Cuint":
        return c.startswith("unsigned int")
    return False


@pytest.mark.skipif(not os.path.exists(HIP_HEADER), reason="ROCm headers not installed")
def test_hip_runtime_ccalls_match_hip_runtime_api_h():
    """HIPArray owns its memory through plain ccalls on libamdhip64 (hipMalloc / hipFree / hipMemcpy / hipMemset / streams): each one
    is compared with the prototype in /opt/rocm/include/hip/hip_runtime_api.h -- symbol, return type, parameter count and kinds, number
    of values passed -- and the hipMemcpyKind / stream-flag constants with the header's values."""
    src = open(JL).read()
    calls = []
    for m in re.finditer(r"ccall\(\(:(hip[A-Za-z]+),\s*libhip\)", src):
        end = balanced(src, src.index("(", m.start()))
        calls.append((m.group(1), split_top(src[src.index("(", m.start()) + 1:end - 1])))
    names = sorted({c[0] for c in calls})
    assert {"hipMalloc", "hipFree", "hipMemcpy", "hipMemset", "hipStreamCreateWithFlags", "hipStreamSynchronize", "hipDeviceSynchronize",
            "hipGetErrorString"} <= set(names)
    protos = hip_prototypes(names)
    for name, args in calls:
        ret_c, params = protos[name]
        ret_jl, types = args[1], args[2]
        tlist = [t for t in split_top(types[1:-1]) if t]
        values = args[3:]
        assert len(tlist) == len(params), f"{name}: {len(tlist)} Julia argument types for {len(params)} C parameters"
        assert len(values) == len(tlist), f"{name}: {len(values)} values for {len(tlist)} types"
        for pos, (jt, ct) in enumerate(zip(tlist, params)):
            assert hip_compatible(jt, ct), f"{name}: argument {pos + 1}: Julia {jt} vs C `{ct}`"
        assert ret_jl == {"hipError_t": "Cint", "const char*": "Cstring", "const char *": "Cstring"}[ret_c], (name, ret_jl, ret_c)
    kinds = dict(re.findall(r"(hipMemcpy\w+)\s*=\s*(\d+)", open("/opt/rocm/include/hip/driver_types.h").read()))
    m = re.search(r"const hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice = Cint\((\d)\), Cint\((\d)\), Cint\((\d)\)", src)
    assert m and [int(x) for x in m.groups()] == [int(kinds[k]) for k in ("hipMemcpyHostToDevice", "hipMemcpyDeviceToHost", "hipMemcpyDeviceToDevice")]
    flag = re.search(r"#define hipStreamNonBlocking (0x[0-9a-fA-F]+)", open(HIP_HEADER).read()).group(1)
    assert re.search(r"const hipStreamNonBlocking = Cuint\((\d+)\)", src).group(1) == str(int(flag, 16))


def test_no_host_pointer_or_cuarray_can_reach_the_c_abi():
    """VERDICT r3 missing #3: the glue owns its device memory.  Statically: HIPArray is a DenseArray with a finalizer that frees; every
    pointer handed to a tpg_* ccall comes from device_pointer / a HIPArray-backed buffer; device_pointer's fallback method throws; the
    grid arrays are allocated through device_array, which accepts HIPGPU() (or Distributed over it) and refuses everything else; no
    CUDA.jl / AMDGPU.jl / KernelAbstractions import, no array_type(child_architecture(...)) allocation."""
    src = open(JL).read()
    code = re.sub(r"#.*", "", re.sub(r'"""(?:.|\n)*?"""', '""', src))
    assert re.search(r"mutable struct HIPArray\{T, N\} <: DenseArray\{T, N\}", code) and "finalizer(unsafe_free!, a)" in code
    assert re.search(r"Base\.getindex\(::HIPArray, I\.\.\.\) = error\(", code) and re.search(r"Base\.setindex!\(::HIPArray, v, I\.\.\.\) = error\(", code)
    assert "struct HIPGPU <: AbstractArchitecture end" in code and "array_type(::HIPGPU) = HIPArray" in code
    assert re.search(r"device_pointer\(a\) = throw\(ArgumentError\(", code)                          # the catch-all refuses
    assert re.search(r"device_array\(arch, FT, dims\.\.\.\) = throw\(ArgumentError\(", code)        # and so does allocation on another architecture
    assert "device_array(::HIPGPU, FT, dims...) = HIPArray{FT}(undef, dims...)" in code
    assert "array_type(child_architecture" not in code and "zeros(" not in code.split("function comm_unique_id")[0].split("mutable struct HIPArray")[1]
    for banned in ("using CUDA", "import CUDA", "CuArray(", "using AMDGPU", "import AMDGPU", "ROCArray(", "using KernelAbstractions", "@kernel"):
        assert banned not in code, banned
    # every data pointer argument of a tpg_* ccall is produced by device_pointer (or is a message-buffer pointer derived from it: bp[...])
    for m in re.finditer(r"ccall\(\(:(tpg_[a-z0-9_]+),\s*libtripolar\)", src):
        end = balanced(src, src.index("(", m.start()))
        args = split_top(src[src.index("(", m.start()) + 1:end - 1])
        for v in args[3:]:
            assert "pointer(" not in v or "device_pointer(" in v, (m.group(1), v)


def test_fallback_invoke_names_the_generic_method():
    """`invoke(fill_halo_regions!, Tuple{Any, Any, ...})` matches no method -- the generic Oceananigans method is typed
    on its first argument.  The fallback must invoke with a signature that method is applicable to: the data's own type first, an abstract
    grid type (never TRG / DTRG) in the grid slot, and the extra `buffers` slot on the distributed path."""
    src = open(JL).read()
    sigs = re.findall(r"invoke\(fill_halo_regions!, (Tuple\{.*?\}), c, bcs", src)
    assert len(sigs) == 2
    # on HIPGPU() (no KernelAbstractions backend) the hand-over must end in an ArgumentError naming the condition, not in a
    # MethodError: every invoke -- and both of Oceananigans' side launchers -- sits behind needs_oceananigans_kernels
    assert len(re.findall(r"needs_oceananigans_kernels\(grid, why_not\([^\n]*\) &&\s+return invoke\(fill_halo_regions!", src)) == 2
    assert "has_ka_backend(::HIPGPU) = false" in src and re.search(r"needs_oceananigans_kernels\(grid, what\) = has_ka_backend\(architecture\(grid\)\) \|\|\s+throw\(ArgumentError\(", src)
    assert re.search(r'needs_oceananigans_kernels\(grid, "a bottom / top boundary condition"\)\s+hip_fill!', src)
    for sig in sigs:
        inner = split_top(sig[len("Tuple{"):-1])
        assert inner[0] == "typeof(c)" and inner[0] != "Any"
        assert inner[4] in ("AbstractGrid", "DistributedGrid") and inner[-1] == "Vararg{Any}"
    assert split_top(sigs[0][6:-1])[4] == "AbstractGrid" and len(split_top(sigs[0][6:-1])) == 6
    assert split_top(sigs[1][6:-1])[4] == "DistributedGrid" and len(split_top(sigs[1][6:-1])) == 7      # ..., grid, buffers, args...
    assert "Tuple{Any, Any, Any, Any, Any" not in src.replace("an all-`Any` signature", "")


def test_distributed_fill_batches_and_spares_the_seam_side():
    """(a) a group of more than TPG_MAX_FIELDS fields is split before the distributed ccall (fields.py batches the
    same way); (b) Oceananigans' south launcher is driven only by south conditions that are neither `nothing` nor the injected
    halo-communication condition of ranks > 0 (that side is a seam, filled by the RCCL exchange)."""
    src = open(JL).read()
    hdr = open(os.path.join(ROOT, "include", "tripolar_hip.h")).read()
    cmax = int(re.search(r"#define TPG_MAX_FIELDS (\d+)", hdr).group(1))
    assert int(re.search(r"const TPG_MAX_FIELDS = (\d+)", src).group(1)) == cmax
    assert "Iterators.partition(group, TPG_MAX_FIELDS)" in src
    assert "fills_south(bc) = !isnothing(bc) && !is_communication(bc)" in src
    assert re.search(r"any\(b -> fills_south\(b\.south\), bs\) && needs_oceananigans_kernels\(grid, [^)]*\) &&\s+south_only!", src)
    assert "any(b -> !isnothing(b.south), bs)" not in src


EXT = os.path.join(ROOT, "orthogonalsphericalshellgrids.jl_amd", "julia", "ext", "TripolarHIPBackendExt.jl")


def _code(path):
    src = open(path).read()
    return re.sub(r"#.*", "", re.sub(r'"""(?:.|\n)*?"""', '""', src))


def test_backend_extension_extracts_pointers_and_streams_only():
    """the opposite hook of HIPArray -- a grid living in the host backend's own device arrays reaches tpg_* through
    three methods (device_pointer, device_array, stream_for) defined in a weak-dependency extension.  Statically: the core file still
    imports no backend and names none; the extension imports the backend, extends exactly the core's hooks, and holds no kernel, no
    launch and no ccall of its own."""
    core, ext = _code(JL), _code(EXT)
    bare = re.sub(r'"(?:\\.|[^"\\\n])*"', '""', core)                       # message strings may name CUDA.jl / the extension; code may not
    for banned in ("AMDGPU", "ROCArray", "ROCBackend", "CUDA", "KernelAbstractions", "TripolarHIPBackendExt"):
        assert banned not in bare, banned                                  # comments and messages aside, the core does not know the backend exists
    assert "stream_for(arch) = current_stream()" in core and "has_ka_backend(arch) = true" in core
    assert "using AMDGPU: AMDGPU, ROCArray, ROCBackend" in ext and "using TripolarHIP" in ext
    assert re.search(r"import TripolarHIP: device_pointer, device_array, stream_for, has_ka_backend", ext)
    defs = re.findall(r"(?m)^(\w+)\((?:[^()]|\([^()]*\))*\) = ", ext)
    assert sorted(defs) == ["device_array", "device_pointer", "has_ka_backend", "stream_for"], defs
    assert "device_pointer(a::ROCArray) = Ptr{Cvoid}(pointer(a))" in ext
    for banned in ("@kernel", "@roc", "launch!", "@index", "ccall", "@cuda", "function "):
        assert banned not in ext, banned
    # every stream argument of a tpg_* ccall in the core comes from stream_for(...) (or is the pipelined exchange's own second stream)
    src = open(JL).read()
    for m in re.finditer(r"ccall\(\(:(tpg_[a-z0-9_]+),\s*libtripolar\)", src):
        end = balanced(src, src.index("(", m.start()))
        args = split_top(src[src.index("(", m.start()) + 1:end - 1])
        assert not any(v.strip() == "current_stream()" for v in args[3:]), m.group(1)


def test_streams_and_tables_are_task_local_or_locked():
    """no unguarded process-wide dictionary keyed by the current task -- streams and seam buffers live in task_local_storage()
    (freed with their task: the stream handle has a finalizer calling hipStreamDestroy), the two remaining process-wide tables (seam
    communicators, table workspaces) are only touched under STATE_LOCK."""
    code = _code(JL)
    assert "TASK_STREAMS" not in code and "COMM_STREAMS" not in code and "SEAM_BUFFERS" not in code and "objectid(current_task())" not in code
    assert "get!(StreamHandle, task_local_storage(), key)" in code and "finalizer(destroy_stream!, h)" in code
    assert re.search(r"ccall\(\(:hipStreamDestroy, libhip\), Cint, \(Ptr\{Cvoid\},\), h\.ptr\)", code)
    assert "get!(task_local_storage(), (:tripolar_hip_seam_buffers" in code
    assert "const STATE_LOCK = ReentrantLock()" in code
    for table in ("SEAM_COMMS", "GRID_WORKSPACES"):
        uses = [m.start() for m in re.finditer(table, code)]
        for u in uses[1:]:                                                # every use after the declaration sits inside a lock(...) expression
            line_start = code.rfind("\n", 0, u)
            ctx = code[max(0, code.rfind("lock(", 0, u)):u]
            assert "lock(" in ctx and (ctx.count("\n") <= 3), (table, code[line_start:u + 40])


def test_build_band_sets_the_tables_valid_flag_for_a_live_workspace():
    """VERDICT r4 next #4 on the Julia side: build_band looks for a live workspace of the same table key, builds with
    TPG_BUILD_TABLES_VALID when it finds one and ties the workspace to the grid's lambda_cc parent (weak key)."""
    src = open(JL).read()
    hdr = open(os.path.join(ROOT, "include", "tripolar_hip.h")).read()
    flag = int(re.search(r"#define TPG_BUILD_TABLES_VALID (\d+)", hdr).group(1))
    assert int(re.search(r"const TPG_BUILD_TABLES_VALID = Int32\((\d+)\)", src).group(1)) == flag
    assert "const GRID_WORKSPACES = WorkspaceOwner[]" in src and "keep_workspace!(arrays[1], ws)" in src
    assert "reuse ? TPG_BUILD_TABLES_VALID : Int32(0)" in src
    # reuse is explicit: the old grid's workspace handed over by with_halo / reconstruct_global_grid, or -- inside
    # share_tables() only -- any live grid's tables of the key
    assert "(sharing_tables() ? live_workspace(key, nbytes) : nothing)" in src and "tables_from.key == key" in src
    assert src.count("_tables_from = table_workspace(") == 3 and "share_tables(f) = task_local_storage(f, :tpg_share_tables, true)" in src
    # the key holds exactly what the header says the tables depend on
    key = re.search(r"table_key\(arch, FT, Nλ, Nφ, Hφ, south, npl, radius\) = \((.*)\)", src).group(1)
    for part in ("serial_arch(arch)", "FT", "Int(Nλ)", "Int(Nφ)", "Int(Hφ)", "Float64(south)", "Float64(npl)", "Float64(radius)"):
        assert part in key
    assert "first_pole" not in key and "jstart" not in key and "Hλ" not in key


def test_no_table_is_keyed_by_a_device_array():
    """a Dict / WeakKeyDict keyed by a device array hashes and compares the array's ELEMENTS (Base.hash(::AbstractArray)
    indexes them); HIPArray refuses scalar indexing, so `GRID_WORKSPACES[arrays[1]] = ws` threw at the end of every build.  The table that
    ties a workspace to its grid is a Vector of (WeakRef(owner), workspace) entries matched with === and pruned under STATE_LOCK; the only
    remaining dictionaries are identity-keyed (IdDict on the architecture value, task_local_storage())."""
    code = _code(JL)
    assert "WeakKeyDict" not in code
    assert re.findall(r"(?<![A-Za-z])Dict\{", code) == [], "a hashed Dict in the glue: key it by identity (IdDict) or use a Vector"
    assert re.search(r"struct WorkspaceOwner\s+owner::WeakRef", code) and "e.owner.value === a && return e.workspace" in code
    assert "filter!(e -> e.owner.value !== nothing, table)" in code
    # nothing is indexed or looked up BY an array: no `[arrays[1]]` / `get(..., parent(...)` on a table
    assert "[arrays[1]]" not in code and not re.search(r"get!?\([A-Z_]+, parent\(", code)
    assert "const SEAM_COMMS = IdDict{Any, SeamComm}()" in code


def test_group_order_is_a_function_of_the_argument_list():
    """group(k) of a rank's seam exchange pairs with group(k) of its neighbour (include/tripolar_hip.h), so the order
    in which a tupled fill's geometry groups issue their RCCL groups must be the same on every rank.  fill_groups returns a Vector of
    `key => indices` in first-appearance order (as fields.py's insertion-ordered dict does); no function that issues tpg_halo_exchange_y* /
    tpg_fill_halo_regions_distributed* iterates a Dict."""
    src = open(JL).read()
    code = _code(JL)
    body = code[code.index("function fill_groups("):code.index("as_tuple(x::Tuple)")]
    assert "groups = Pair{Tuple{DataType, Int, Int}, Vector{Int}}[]" in body and "Dict" not in body
    assert "findfirst(g -> first(g) == key, groups)" in body and "push!(groups, key => [n])" in body
    # every function whose body reaches an exchange entry point: iterates fill_groups(...) (a Vector) and no Dict / keys() / values() / pairs()
    starts = [m.start() for m in re.finditer(r"(?m)^function \w+!?\(", code)] + [len(code)]
    issuing = []
    for a, b in zip(starts, starts[1:]):
        fn = code[a:b]
        if re.search(r":tpg_(halo_exchange_y|fill_halo_regions_distributed)", fn):
            issuing.append(fn)
            assert "in fill_groups(" in fn, fn[:80]
            for banned in ("Dict", "keys(", "values(", "pairs(", "Set("):
                assert banned not in fn, (banned, fn[:80])
    assert len(issuing) == 2                                             # hip_fill! and halo_exchange_y!
    fpy = open(os.path.join(ROOT, "orthogonalsphericalshellgrids.jl_amd", "fields.py")).read()
    assert "groups.setdefault(key, []).append(f)" in fpy and "return groups.values()" in fpy    # insertion order on the Python side
