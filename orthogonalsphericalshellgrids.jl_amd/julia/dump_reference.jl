# dump_reference.jl -- run the REFERENCE package itself (CliMA/OrthogonalSphericalShellGrids.jl on Oceananigans, CPU) and write what it
# produces as raw little-endian files + a JSON manifest, for tests/compare_reference_dump.py to compare with libtripolar_hip and with the
# repository's CPU restatement (its checker).  This is the parity-pinning kit of SURVEY.md 8(c): the build container has no Julia toolchain, so every Float64 value of the
# metric precompute beyond the 6 digits of the README transcript is "parity unpinned" (DESIGN.md 2) -- until this script has run once
# at a site that has Julia and the package:
#
#     julia --project=<environment with OrthogonalSphericalShellGrids v0.2.x> dump_reference.jl <outdir>
#     python tests/compare_reference_dump.py <outdir>            # on a box with the built library (MI355X); without a GPU: its CPU-only switch
#
# NOT exercised in the build container (no Julia).  It only calls the package's EXPORTED API -- TripolarGrid (src/tripolar_grid.jl:59),
# Field{LX, LY, LZ}(grid), set!, fill_halo_regions! -- exactly as the package's own tests do
# (test/test_zipper_boundary_conditions.jl:5-31, test/runtests.jl:8-41); no file of the reference is copied.
#
# Format "tripolar-reference-dump-1" (tests/compare_reference_dump.py writes the same format from an .npz for its self-test):
#   <outdir>/manifest.json          {"format", "endianness": "little", "generator", "cases": [ case ... ]}
#   case = {"name", "kwargs": {size, halo, north_poles_latitude, first_pole_longitude, southernmost_latitude, radius}, "eltype",
#           "arrays": {name: {"file", "dims": [n1, n2(, n3)], "eltype"}}, "fields": {name: {..., "location", "sign", "initial"}}}
#   <outdir>/<case>/<name>.bin      the PARENT array (halos included), column-major as Julia holds it: first index (i) fastest --
#                                   which is the layout of include/tripolar_hip.h, so numpy reads it as shape dims[::-1], C order.
# Grid arrays are dumped by FIELD NAME of the OrthogonalSphericalShellGrid struct (getproperty(grid, :Δyᶠᶜᵃ) -> "dy_fc"), not by the
# position they had in the constructor call: src/tripolar_grid.jl:321-324 passes Δy as (cc, cf, fc, ff), and whether Oceananigans'
# positional parameters are named in that order is exactly one of the things the comparison settles.
#
# SECOND MODE, `--glue`: the same cases through THIS repository's Julia binding instead of the reference package --
#
#     julia --project=<environment with Oceananigans> dump_reference.jl --glue <outdir>      # on an MI355X host, LIBTRIPOLAR_HIP set
#     python tests/compare_reference_dump.py <outdir>                                        # expects every array BIT-identical
#
# -- i.e. the first execution of julia/TripolarHIP.jl: TripolarGrid(HIPGPU(), FT; ...), the Field constructors and the intercepted
# fill_halo_regions!.  What the library computes is already checked through Python; this checks what the glue adds (argument marshalling,
# array order and layout, the field-location / sign policy, the fill interception): the comparator demands 0 differing elements.
using Oceananigans
using Oceananigans.BoundaryConditions: fill_halo_regions!
using Oceananigans.Grids: halo_size

const GLUE = "--glue" in ARGS
if GLUE
    include(joinpath(@__DIR__, "TripolarHIP.jl"))
    using .TripolarHIP
else
    using OrthogonalSphericalShellGrids
end
dump_arch() = GLUE ? HIPGPU() : CPU()

const GRID_ARRAYS = (
    "lambda_cc" => :λᶜᶜᵃ, "lambda_fc" => :λᶠᶜᵃ, "lambda_cf" => :λᶜᶠᵃ, "lambda_ff" => :λᶠᶠᵃ,
    "phi_cc" => :φᶜᶜᵃ, "phi_fc" => :φᶠᶜᵃ, "phi_cf" => :φᶜᶠᵃ, "phi_ff" => :φᶠᶠᵃ,
    "dx_cc" => :Δxᶜᶜᵃ, "dx_fc" => :Δxᶠᶜᵃ, "dx_cf" => :Δxᶜᶠᵃ, "dx_ff" => :Δxᶠᶠᵃ,
    "dy_cc" => :Δyᶜᶜᵃ, "dy_fc" => :Δyᶠᶜᵃ, "dy_cf" => :Δyᶜᶠᵃ, "dy_ff" => :Δyᶠᶠᵃ,
    "az_cc" => :Azᶜᶜᵃ, "az_fc" => :Azᶠᶜᵃ, "az_cf" => :Azᶜᶠᵃ, "az_ff" => :Azᶠᶠᵃ)

# the grids of the reference's own tests and README, plus the 1-degree grid of its orthogonality test
const GRID_CASES = (
    (name = "grid_4x5_f32_poles75_35", FT = Float32, kwargs = (size = (4, 5, 1), first_pole_longitude = 75, north_poles_latitude = 35)),   # test/runtests.jl:10-25
    (name = "grid_10x10_f64",          FT = Float64, kwargs = (size = (10, 10, 1),)),                                                    # test/test_zipper_boundary_conditions.jl:6
    (name = "grid_60x30_f64",          FT = Float64, kwargs = (size = (60, 30, 1),)),                                                    # README.md:54
    (name = "grid_60x30_f64_halo5",    FT = Float64, kwargs = (size = (60, 30, 1), halo = (5, 5, 5))),                                    # examples/bickley_jet.jl:21
    (name = "grid_360x180_f64_poles75_35", FT = Float64, kwargs = (size = (360, 180, 1), first_pole_longitude = 75, north_poles_latitude = 35)))   # test/test_tripolar_grid.jl:52-57

json_str(s::AbstractString) = "\"" * replace(s, "\\" => "\\\\", "\"" => "\\\"") * "\""
json(x::AbstractString) = json_str(x)
json(x::Symbol) = json_str(String(x))
json(x::Bool) = x ? "true" : "false"
json(x::Integer) = string(x)
json(x::AbstractFloat) = isfinite(x) ? repr(Float64(x)) : "null"
json(x::Nothing) = "null"
json(x::Union{Tuple, AbstractVector}) = "[" * join((json(v) for v in x), ", ") * "]"
json(x::NamedTuple) = "{" * join((json_str(String(k)) * ": " * json(v) for (k, v) in pairs(x)), ", ") * "}"
json(x::AbstractDict) = "{" * join((json_str(String(k)) * ": " * json(v) for (k, v) in x), ", ") * "}"

eltype_name(::Type{Float64}) = "Float64"
eltype_name(::Type{Float32}) = "Float32"

"write the parent of `a` (an OffsetArray / Array) raw, little-endian, column-major; returns the manifest entry"
function dump_array(dir, name, a)
    p = Array(parent(a))
    open(joinpath(dir, name * ".bin"), "w") do io
        write(io, htol.(p))
    end
    return (file = name * ".bin", dims = collect(size(p)), eltype = eltype_name(eltype(p)))
end

function dump_grid(outdir, case)
    dir = joinpath(outdir, case.name)
    mkpath(dir)
    grid = TripolarGrid(dump_arch(), case.FT; case.kwargs...)
    arrays = Dict{String, Any}()
    for (name, sym) in GRID_ARRAYS
        arrays[name] = dump_array(dir, name, getproperty(grid, sym))
    end
    cm = grid.conformal_mapping
    kw = (size = collect(size(grid)), halo = collect(halo_size(grid)), north_poles_latitude = Float64(cm.north_poles_latitude),
          first_pole_longitude = Float64(cm.first_pole_longitude), southernmost_latitude = Float64(cm.southernmost_latitude),
          radius = Float64(grid.radius))
    return (name = case.name, kwargs = kw, eltype = eltype_name(case.FT), arrays = arrays, fields = Dict{String, Any}())
end

# test/test_zipper_boundary_conditions.jl:5-31 and :56-63: c, u, v filled with 1, and c, u filled with x, then fill_halo_regions!
function dump_fields(outdir, halo)
    name = "fields_10x10_halo$(halo[1])"
    dir = joinpath(outdir, name)
    mkpath(dir)
    grid = TripolarGrid(dump_arch(); size = (10, 10, 1), halo)
    fields = Dict{String, Any}()
    loc(f) = collect(String(nameof(L)) for L in Oceananigans.Fields.location(f))
    # glue mode: HIPGPU() has no Oceananigans kernels, so `set!` cannot run there; the pre-fill state is written into the parent array
    # directly (reproducible values everywhere, halos included -- the fill must overwrite them) and the comparator fills the same state
    function initialise!(f, init, k)
        GLUE || return set!(f, init)
        p = parent(f.data)
        copyto!(p, reshape([sin(0.37 * (n + 1000k)) for n in 1:length(p)], size(p)))
        return f
    end
    # glue mode builds z-reduced fields (the reduced-field case of test/test_zipper_boundary_conditions.jl:47-54): a default 3-D field carries
    # Oceananigans' no-flux bottom / top conditions, whose halo kernels HIPGPU() cannot launch (the binding says so with an ArgumentError)
    make_field(LX, LY) = GLUE ? Field{LX, LY, Nothing}(grid) : Field{LX, LY, Center}(grid)
    for (k, (fname, LX, LY, init, what)) in enumerate((("c_one", Center, Center, 1, "set!(c, 1)"), ("u_one", Face, Center, 1, "set!(u, 1)"),
                                                       ("v_one", Center, Face, 1, "set!(v, 1)"), ("c_x", Center, Center, (x, y, z) -> x, "set!(c, (x, y, z) -> x)"),
                                                       ("u_x", Face, Center, (x, y, z) -> x, "set!(u, (x, y, z) -> x)"),
                                                       ("v_x", Center, Face, (x, y, z) -> x, "set!(v, (x, y, z) -> x)")))
        f = make_field(LX, LY)
        initialise!(f, init, k)
        before = dump_array(dir, fname * "_before", f.data)
        fill_halo_regions!(f)
        after = dump_array(dir, fname, f.data)
        fields[fname] = (file = after.file, before = before.file, dims = after.dims, eltype = after.eltype, location = loc(f),
                         sign = Int(f.boundary_conditions.north.condition), initial = GLUE ? "sin(0.37 (n + 1000 k)) over the whole parent" : what)
    end
    kw = (size = collect(size(grid)), halo = collect(halo_size(grid)), north_poles_latitude = 55.0, first_pole_longitude = 70.0,
          southernmost_latitude = -80.0, radius = Float64(grid.radius))
    return (name = name, kwargs = kw, eltype = "Float64", arrays = Dict{String, Any}(), fields = fields)
end

function main(outdir)
    mkpath(outdir)
    cases = Any[]
    for case in GRID_CASES
        push!(cases, dump_grid(outdir, case))
        println("dumped ", case.name)
    end
    for halo in ((4, 4, 4), (5, 5, 5))
        push!(cases, dump_fields(outdir, halo))
    end
    generator = GLUE ? "glue: julia $(VERSION), TripolarHIP.jl over $(TripolarHIP.libtripolar), Oceananigans $(pkgversion(Oceananigans))" :
                       "julia $(VERSION), OrthogonalSphericalShellGrids $(pkgversion(OrthogonalSphericalShellGrids)), Oceananigans $(pkgversion(Oceananigans))"
    open(joinpath(outdir, "manifest.json"), "w") do io
        print(io, "{\"format\": \"tripolar-reference-dump-1\", \"endianness\": \"little\", \"generator\": ", json_str(generator), ", \"cases\": [\n")
        print(io, join((json(c) for c in cases), ",\n"))
        print(io, "\n]}\n")
    end
    println("wrote ", joinpath(outdir, "manifest.json"))
end

let dirs = filter(a -> !startswith(a, "--"), ARGS)
    main(isempty(dirs) ? (GLUE ? "glue_dump" : "reference_dump") : dirs[1])
end
