# TripolarHIP.jl -- Julia glue over libtripolar_hip.so (include/tripolar_hip.h).
#
# NOT exercised in the build container (no Julia toolchain, SURVEY.md 8c): this file is the reference-side binding a
# maintainer adds so that Oceananigans keeps seeing TripolarGrid() / ZipperBoundaryCondition / fill_halo_regions!
# while the numerics run in the hand-written HIP kernels.  It is written as a replacement for the package's `src/`:
# it defines the package's own names (`Tripolar`, `Zipper`, `TripolarGrid`, `TRG`, `DTRG`, `ZBC`) and extends the same
# Oceananigans generics, method for method, in the order of the reference; every method names the reference method it
# replaces (file:line in CliMA/OrthogonalSphericalShellGrids.jl v0.2.1).  Names of Oceananigans internals that the
# reference itself does not spell out (the south/north halo launcher) are marked [recalled].
#
# No CUDA.jl, no KernelAbstractions / AMDGPU.jl code generation: device memory is reached through raw pointers.
# Three hooks adapt it to the array backend in use: `device_pointer`, `current_stream`, `device_zeros`.
module TripolarHIP

export TripolarGrid, ZipperBoundaryCondition           # src/OrthogonalSphericalShellGrids.jl:4

using Oceananigans
using Oceananigans.Architectures: AbstractArchitecture, architecture, child_architecture, on_architecture
using Oceananigans.Grids: R_Earth, Center, Face, Periodic, Bounded, RightConnected, FullyConnected,
                          OrthogonalSphericalShellGrid, generate_coordinate, halo_size, topology, cpu_face_constructor_z
using Oceananigans.ImmersedBoundaries: ImmersedBoundaryGrid
using Oceananigans.BoundaryConditions: AbstractBoundaryConditionClassification, BoundaryCondition, FieldBoundaryConditions,
                                       assumed_field_location, regularize_boundary_condition,
                                       regularize_immersed_boundary_condition, LeftBoundary, RightBoundary
using Oceananigans.Fields: validate_indices, validate_boundary_conditions, validate_field_data, FieldBoundaryBuffers
using Oceananigans.DistributedComputations: Distributed, local_size, ranks, concatenate_local_sizes,
                                            inject_halo_communication_boundary_conditions
using OffsetArrays
using Adapt

import Oceananigans.BoundaryConditions: bc_str, apply_y_north_bc!, regularize_field_boundary_conditions
import Oceananigans.Fields: Field, validate_boundary_condition_location
import Oceananigans.Grids: x_domain, y_domain, with_halo
import Oceananigans.DistributedComputations: reconstruct_global_grid

const libtripolar = get(ENV, "LIBTRIPOLAR_HIP", "libtripolar_hip.so")

# ---------------------------------------------------------------------------------------------------------------------
# 1. C structs / status handling (include/tripolar_hip.h)
# ---------------------------------------------------------------------------------------------------------------------
struct TpgParams
    Nx::Int32; Ny::Int32; Nz::Int32
    Hx::Int32; Hy::Int32; Hz::Int32
    southernmost_latitude::Float64
    north_poles_latitude::Float64
    first_pole_longitude::Float64
    radius::Float64
    ft::Int32; jstart::Int32; jend::Int32; reserved::Int32
end

const TPG_F32, TPG_F64 = Int32(0), Int32(1)
ft_code(::Type{Float32}) = TPG_F32
ft_code(::Type{Float64}) = TPG_F64

struct TripolarHIPError <: Exception
    status::Cint
    msg::String
end

function check(status::Cint)
    status == 0 && return nothing
    msg = unsafe_string(ccall((:tpg_last_error, libtripolar), Cstring, ()))
    # the reference's exception types: ArgumentError for odd Nlambda (tripolar_grid.jl:81-83), for a non-y partition
    # (distributed_tripolar_grid.jl:28-31) and for a zipper on a non-north side (zipper_boundary_condition.jl:58-62)
    (status == -2 || status == -3 || status == -6) && throw(ArgumentError(msg))
    throw(TripolarHIPError(status, msg))
end

# backend hooks -------------------------------------------------------------------------------------------------------
device_pointer(a) = Ptr{Cvoid}(UInt(pointer(parent(a))))         # raw HBM address of the parent array
current_stream() = C_NULL                                         # hipStream_t of the task; NULL = default stream
device_zeros(arch, FT, dims...) = on_architecture(arch, zeros(FT, dims...))   # a device array (HBM) of that shape

# ---------------------------------------------------------------------------------------------------------------------
# 2. Tripolar mapping record, grid aliases            src/tripolar_grid.jl:6-17,371; distributed_tripolar_grid.jl:12-15
# ---------------------------------------------------------------------------------------------------------------------
struct Tripolar{N, F, S}
    north_poles_latitude::N
    first_pole_longitude::F
    southernmost_latitude::S
end

Adapt.adapt_structure(to, t::Tripolar) = Tripolar(Adapt.adapt(to, t.north_poles_latitude),
                                                  Adapt.adapt(to, t.first_pole_longitude),
                                                  Adapt.adapt(to, t.southernmost_latitude))

const TripolarGrid{FT, TX, TY, TZ, CZ, A, Arch} = OrthogonalSphericalShellGrid{FT, TX, TY, TZ, CZ, A, <:Tripolar, Arch}
const DistributedTripolarGrid{FT, TX, TY, TZ, CZ, A, Arch} =
    OrthogonalSphericalShellGrid{FT, TX, TY, TZ, CZ, A, <:Tripolar, <:Distributed}
const TRG  = Union{TripolarGrid, ImmersedBoundaryGrid{<:Any, <:Any, <:Any, <:Any, <:TripolarGrid}}
const DTRG = Union{DistributedTripolarGrid, ImmersedBoundaryGrid{<:Any, <:Any, <:Any, <:Any, <:DistributedTripolarGrid}}

# ---------------------------------------------------------------------------------------------------------------------
# 3. TripolarGrid constructors                         src/tripolar_grid.jl:59-333; distributed_tripolar_grid.jl:24-110
# ---------------------------------------------------------------------------------------------------------------------
# ONE tpg_build_grid call fills the 20 padded arrays of the latitude band jstart:jend in HBM (no host passes, no H2D).
function build_band(arch, FT, size, halo, southernmost_latitude, radius, z, north_poles_latitude, first_pole_longitude,
                    jstart, jend, LY)
    Nλ, Nφ, Nz = size
    Hλ, Hφ, Hz = halo
    isodd(Nλ) && throw(ArgumentError("The number of cells in the longitude dimension should be even!"))   # :81-83
    ny = jend - jstart + 1
    p = Ref(TpgParams(Nλ, Nφ, Nz, Hλ, Hφ, Hz, southernmost_latitude, north_poles_latitude, first_pole_longitude,
                      radius, ft_code(FT), jstart, jend, 0))
    arrays = [device_zeros(arch, FT, Nλ + 2Hλ, ny + 2Hφ) for _ in 1:20]
    nbytes = ccall((:tpg_build_grid_workspace_bytes, libtripolar), Csize_t, (Ref{TpgParams},), p)
    workspace = device_zeros(arch, UInt8, Int(nbytes))
    ptrs = Ptr{Cvoid}[device_pointer(a) for a in arrays]
    GC.@preserve arrays workspace begin
        check(ccall((:tpg_build_grid, libtripolar), Cint,
                    (Ref{TpgParams}, Ptr{Ptr{Cvoid}}, Ptr{Cvoid}, Csize_t, Ptr{Cvoid}),
                    p, ptrs, device_pointer(workspace), nbytes, current_stream()))
    end
    # enum tpg_array == positional order of src/tripolar_grid.jl:308-328 (note dy: cc, cf, fc, ff)
    off(a) = OffsetArray(a, -Hλ, -Hφ)
    λcc, λfc, λcf, λff, φcc, φfc, φcf, φff,
    Δxcc, Δxfc, Δxcf, Δxff, Δycc, Δycf, Δyfc, Δyff, Azcc, Azfc, Azcf, Azff = off.(arrays)
    Lz, zc = generate_coordinate(FT, (Periodic, RightConnected, Bounded), size, halo, z, :z, 3, CPU())   # :91 (z stays with Oceananigans)
    return OrthogonalSphericalShellGrid{Periodic, LY, Bounded}(arch, Nλ, ny, Nz, Hλ, Hφ, Hz, convert(FT, Lz),
               λcc, λfc, λcf, λff, φcc, φfc, φcf, φff, on_architecture(arch, zc),
               Δxcc, Δxfc, Δxcf, Δxff, Δycc, Δycf, Δyfc, Δyff, Azcc, Azfc, Azcf, Azff,
               convert(FT, radius), Tripolar(north_poles_latitude, first_pole_longitude, southernmost_latitude))
end

"""
    TripolarGrid(arch = CPU(), FT = Float64; size, southernmost_latitude = -80, halo = (4, 4, 4),
                 radius = R_Earth, z = (0, 1), north_poles_latitude = 55, first_pole_longitude = 70)

Same keywords, defaults, return type and `ArgumentError` as src/tripolar_grid.jl:59-66,81-83,304-330.
"""
function TripolarGrid(arch::AbstractArchitecture = CPU(), FT::DataType = Float64; size, southernmost_latitude = -80,
                      halo = (4, 4, 4), radius = R_Earth, z = (0, 1), north_poles_latitude = 55, first_pole_longitude = 70)
    return build_band(arch, FT, size, halo, southernmost_latitude, radius, z, north_poles_latitude,
                      first_pole_longitude, 1, size[2], RightConnected)
end

"""
    TripolarGrid(arch::Distributed, FT = Float64; halo = (4, 4, 4), kwargs...)

src/distributed_tripolar_grid.jl:24-110: y-partitioning only; the rank's band is evaluated directly on its device (the
reference builds the whole globe on every rank's CPU and slices it).
"""
function TripolarGrid(arch::Distributed, FT::DataType = Float64; halo = (4, 4, 4), size, southernmost_latitude = -80,
                      radius = R_Earth, z = (0, 1), north_poles_latitude = 55, first_pole_longitude = 70)
    workers = ranks(arch.partition)
    workers[1] != 1 &&
        throw(ArgumentError("The tripolar grid is supported only on a Y-partitioning configuration"))      # :28-31
    lsize  = local_size(arch, size)                                                                          # :41
    nlocal = concatenate_local_sizes(lsize, arch, 2)                                                         # :44
    rank   = arch.local_rank
    jstart = 1 + sum(nlocal[1:rank])                                                                         # :47
    jend   = rank == workers[2] - 1 ? size[2] : sum(nlocal[1:rank+1])                                        # :48
    LY     = rank == 0 ? RightConnected : FullyConnected                                                     # :75
    return build_band(arch, FT, size, halo, southernmost_latitude, radius, z, north_poles_latitude,
                      first_pole_longitude, jstart, jend, LY)
end

# src/tripolar_grid_extensions.jl:20-21
x_domain(grid::TRG) = 0, 360
y_domain(grid::TRG) = minimum(parent(grid.φᶠᶠᵃ)), 90

# src/with_halo.jl:5-23 (serial) and :25-44 (distributed: `radius` is not forwarded there -- kept)
function with_halo(new_halo, old_grid::TripolarGrid)
    cm = old_grid.conformal_mapping
    return TripolarGrid(architecture(old_grid), eltype(old_grid); size = (old_grid.Nx, old_grid.Ny, old_grid.Nz),
                        z = cpu_face_constructor_z(old_grid), halo = new_halo, radius = old_grid.radius,
                        north_poles_latitude = cm.north_poles_latitude, first_pole_longitude = cm.first_pole_longitude,
                        southernmost_latitude = cm.southernmost_latitude)
end

function with_halo(new_halo, old_grid::DistributedTripolarGrid)
    arch = old_grid.architecture
    N  = map(sum, concatenate_local_sizes(size(old_grid), arch))
    cm = old_grid.conformal_mapping
    return TripolarGrid(arch, eltype(old_grid); halo = new_halo, size = N, z = cpu_face_constructor_z(old_grid),
                        north_poles_latitude = cm.north_poles_latitude, first_pole_longitude = cm.first_pole_longitude,
                        southernmost_latitude = cm.southernmost_latitude)
end

# src/distributed_tripolar_grid.jl:201-226
function reconstruct_global_grid(grid::DistributedTripolarGrid)
    arch = grid.architecture
    cm = grid.conformal_mapping
    return TripolarGrid(child_architecture(arch), eltype(grid); halo = halo_size(grid),
                        size = map(sum, concatenate_local_sizes(size(grid), arch)), z = cpu_face_constructor_z(grid),
                        north_poles_latitude = cm.north_poles_latitude, first_pole_longitude = cm.first_pole_longitude,
                        southernmost_latitude = cm.southernmost_latitude)
end

# ---------------------------------------------------------------------------------------------------------------------
# 4. Zipper boundary condition: metadata                                  src/zipper_boundary_condition.jl:8,52-64
# ---------------------------------------------------------------------------------------------------------------------
struct Zipper <: AbstractBoundaryConditionClassification end
ZipperBoundaryCondition(sign = 1) = BoundaryCondition(Zipper(), sign)
const ZBC = BoundaryCondition{<:Zipper}
bc_str(::ZBC) = "Zipper"

north_only(bc, loc, side) = side == :north ? nothing :
    throw(ArgumentError("Cannot specify $side boundary condition $bc on a field at $(loc) (north only)!"))
validate_boundary_condition_location(bc::Zipper, loc::Center, side) = north_only(bc, loc, side)   # :58-59
validate_boundary_condition_location(bc::Zipper, loc::Face,   side) = north_only(bc, loc, side)   # :61-62
@inline apply_y_north_bc!(Gc, loc, ::ZBC, args...) = nothing                                       # :64

# location -> sign (src/tripolar_grid_extensions.jl:49-53): edges carry signed vectors, nodes and centres scalars
sign(LX, LY) = 1
sign(::Type{Face},   ::Type{Center}) = -1
sign(::Type{Center}, ::Type{Face})   = -1

# src/tripolar_grid_extensions.jl:25-44
function regularize_field_boundary_conditions(bcs::FieldBoundaryConditions, grid::TRG, field_name::Symbol,
                                              prognostic_names = nothing)
    loc = assumed_field_location(field_name)
    sgn = field_name == :u || field_name == :v ? -1 : 1
    reg(bc, dim, side) = regularize_boundary_condition(bc, grid, loc, dim, side, prognostic_names)
    return FieldBoundaryConditions(reg(bcs.west, 1, LeftBoundary), reg(bcs.east, 1, RightBoundary),
                                   reg(bcs.south, 2, LeftBoundary), ZipperBoundaryCondition(sgn),
                                   reg(bcs.bottom, 3, LeftBoundary), reg(bcs.top, 3, RightBoundary),
                                   regularize_immersed_boundary_condition(bcs.immersed, grid, loc, field_name, prognostic_names))
end

# src/distributed_tripolar_grid.jl:129-155: the zipper only on the last rank (the other ranks' north side is regularised
# from `bcs.south`, as written in the reference, :147)
function regularize_field_boundary_conditions(bcs::FieldBoundaryConditions, grid::DTRG, field_name::Symbol,
                                              prognostic_names = nothing)
    arch = architecture(grid)
    loc  = assumed_field_location(field_name)
    sgn  = field_name == :u || field_name == :v ? -1 : 1
    reg(bc, dim, side) = regularize_boundary_condition(bc, grid, loc, dim, side, prognostic_names)
    north = arch.local_rank == ranks(arch.partition)[2] - 1 ? ZipperBoundaryCondition(sgn) : reg(bcs.south, 2, RightBoundary)
    return FieldBoundaryConditions(reg(bcs.west, 1, LeftBoundary), reg(bcs.east, 1, RightBoundary),
                                   reg(bcs.south, 2, LeftBoundary), north,
                                   reg(bcs.bottom, 3, LeftBoundary), reg(bcs.top, 3, RightBoundary),
                                   regularize_immersed_boundary_condition(bcs.immersed, grid, loc, field_name, prognostic_names))
end

# src/tripolar_grid_extensions.jl:57-80
function Field((LX, LY, LZ)::Tuple, grid::TRG, data, old_bcs, indices::Tuple, op, status)
    indices = validate_indices(indices, (LX, LY, LZ), grid)
    validate_field_data((LX, LY, LZ), data, grid, indices)
    validate_boundary_conditions((LX, LY, LZ), grid, old_bcs)
    new_bcs = old_bcs
    if !(isnothing(old_bcs) || ismissing(old_bcs))
        north = old_bcs.north isa ZBC ? old_bcs.north : ZipperBoundaryCondition(sign(LX, LY))
        new_bcs = FieldBoundaryConditions(; west = old_bcs.west, east = old_bcs.east, south = old_bcs.south,
                                            north, top = old_bcs.top, bottom = old_bcs.bottom)
    end
    buffers = FieldBoundaryBuffers(grid, data, new_bcs)
    return Field{LX, LY, LZ}(grid, data, new_bcs, indices, op, status, buffers)
end

# src/distributed_tripolar_grid.jl:159-198
function Field((LX, LY, LZ)::Tuple, grid::DTRG, data, old_bcs, indices::Tuple, op, status)
    arch = architecture(grid)
    indices = validate_indices(indices, (LX, LY, LZ), grid)
    validate_field_data((LX, LY, LZ), data, grid, indices)
    validate_boundary_conditions((LX, LY, LZ), grid, old_bcs)
    new_bcs = old_bcs
    if !(isnothing(old_bcs) || ismissing(old_bcs))
        inj = inject_halo_communication_boundary_conditions(old_bcs, arch.local_rank, arch.connectivity, topology(grid))
        last_rank = arch.local_rank == ranks(arch.partition)[2] - 1
        north = last_rank ? (old_bcs.north isa ZBC ? old_bcs.north : ZipperBoundaryCondition(sign(LX, LY))) : inj.north
        new_bcs = FieldBoundaryConditions(; west = inj.west, east = inj.east, south = inj.south, north,
                                            top = inj.top, bottom = inj.bottom)
    end
    buffers = FieldBoundaryBuffers(grid, data, new_bcs)
    return Field{LX, LY, LZ}(grid, data, new_bcs, indices, op, status, buffers)
end

# ---------------------------------------------------------------------------------------------------------------------
# 5. The halo-fill hook                                                 src/zipper_boundary_condition.jl:140-155
# ---------------------------------------------------------------------------------------------------------------------
# The reference's `_fill_north_halo!(i, k, grid, c, bc::ZBC, loc, args...)` is a per-thread function inlined into
# Oceananigans' south/north halo kernel; a ccall cannot live there.  The interception point is one level up: the method of
# the south/north launcher whose north condition is a ZBC ([recalled] `fill_south_and_north_halo!(c, south_bc, north_bc,
# size, offset, loc, arch, grid, args...)` in Oceananigans 0.95-0.99; a tuple `c` is the tupled fill).
loc_code(::Center) = Int8(0);  loc_code(::Type{Center}) = Int8(0)
loc_code(::Face)   = Int8(1);  loc_code(::Type{Face})   = Int8(1)

# (Nz, Hz) of ONE field from its own parent and z-location: a reduced field (LZ = Nothing, e.g. bottom_height at
# (Center, Center, Nothing), test/test_zipper_boundary_conditions.jl:47-54) has one level and no z halo; a z-Face field
# has Nz + 1 levels.  The grid's Nz / Hz are NOT used: indexing a 1-level parent with them would write out of bounds.
function field_levels(c, loc, grid)
    LZ = loc[3]
    (LZ === Nothing || LZ isa Nothing) && return 1, 0
    Hz = halo_size(grid)[3]
    nlev = size(parent(c), 3)
    return nlev - 2Hz, Hz                              # Center: Nz, Face (Bounded z): Nz + 1
end

zipper_sign(bc::ZBC) = Int32(bc.condition)

# one C call per group of fields that share (element type, Nz, Hz): a tupled fill mixes 3-D and reduced fields
function zipper_groups(fields, locs, grid)
    groups = Dict{Tuple{DataType, Int, Int}, Vector{Int}}()
    for (n, (c, loc)) in enumerate(zip(fields, locs))
        Nz, Hz = field_levels(c, loc, grid)
        push!(get!(groups, (eltype(parent(c)), Nz, Hz), Int[]), n)
    end
    return groups
end

"""
    zipper_fill!(fields, bcs, locs, grid; periodic_x = false)

`fold_north_*!` for every (i, k) of every field (src/zipper_boundary_condition.jl:70-155), one batched launch per
geometry group; with `periodic_x = true` the whole `fill_halo_regions!` order zipper -> periodic west/east (pinned by
test/test_zipper_boundary_conditions.jl:42-45; small 2-D fields take one fused launch inside the library).
"""
function zipper_fill!(fields, bcs, locs, grid; periodic_x::Bool = false)
    Nx, Ny, _ = size(grid)
    Hx, Hy, _ = halo_size(grid)
    for ((FT, Nz, Hz), idx) in zipper_groups(fields, locs, grid)
        fs   = [fields[n] for n in idx]
        ptrs = Ptr{Cvoid}[device_pointer(f) for f in fs]
        xloc = Int8[loc_code(locs[n][1]) for n in idx]
        yloc = Int8[loc_code(locs[n][2]) for n in idx]
        sgn  = Int32[zipper_sign(bcs[n]) for n in idx]
        GC.@preserve fs begin
            status = periodic_x ?
                ccall((:tpg_fill_halo_regions, libtripolar), Cint,
                      (Ptr{Ptr{Cvoid}}, Cint, Ptr{Int8}, Ptr{Int8}, Ptr{Int32},
                       Cint, Cint, Cint, Cint, Cint, Cint, Cint, Cint, Ptr{Cvoid}),
                      ptrs, length(fs), xloc, yloc, sgn, Nx, Ny, Nz, Hx, Hy, Hz, 1, ft_code(FT), current_stream()) :
                ccall((:tpg_zipper_fill, libtripolar), Cint,
                      (Ptr{Ptr{Cvoid}}, Cint, Ptr{Int8}, Ptr{Int8}, Ptr{Int32},
                       Cint, Cint, Cint, Cint, Cint, Cint, Cint, Cint, Cint, Ptr{Cvoid}),
                      ptrs, length(fs), xloc, yloc, sgn, Nx, Ny, Nz, Hx, Hy, Hz, 1, Nz, ft_code(FT), current_stream())
            check(status)
        end
    end
    return nothing
end

# The launcher's name is Oceananigans-internal and version dependent: the methods are added only where it exists, so that the
# rest of the module (grid construction, metadata, exchange, geometry) loads on any version; `zipper_fill!` stays callable.
@static if isdefined(Oceananigans.BoundaryConditions, :fill_south_and_north_halo!)

import Oceananigans.BoundaryConditions: fill_south_and_north_halo!

# single field: what fill_halo_regions!(field) reaches
function fill_south_and_north_halo!(c, south_bc, north_bc::ZBC, size, offset, loc, arch, grid::Union{TRG, DTRG}, args...; kwargs...)
    # the south side stays Oceananigans' (the reference's own fills leave it `nothing`, src/tripolar_grid.jl:148)
    isnothing(south_bc) || Oceananigans.BoundaryConditions.fill_south_halo!(c, south_bc, size, offset, loc, arch, grid, args...; kwargs...)   # [recalled]
    zipper_fill!((c,), (north_bc,), (loc,), grid)
    return nothing
end

# tupled fill: fill_halo_regions!((u, v, c, ...)) hands tuples of data, conditions and locations
function fill_south_and_north_halo!(c::NTuple, south_bc, north_bc::NTuple{N, <:ZBC}, size, offset, loc, arch,
                                    grid::Union{TRG, DTRG}, args...; kwargs...) where N
    for n in 1:N
        isnothing(south_bc[n]) || Oceananigans.BoundaryConditions.fill_south_halo!(c[n], south_bc[n], size, offset, loc[n], arch, grid, args...; kwargs...)   # [recalled]
    end
    zipper_fill!(c, north_bc, loc, grid)
    return nothing
end

end # @static if: launcher present

# ---------------------------------------------------------------------------------------------------------------------
# 6. Latitude-band seam exchange over RCCL                         src/distributed_tripolar_grid.jl:171,195 (transport)
# ---------------------------------------------------------------------------------------------------------------------
mutable struct SeamComm
    handle::Ptr{Cvoid}
    rank::Int
    nranks::Int
end

"128-byte ncclUniqueId: draw it on rank 0 and broadcast it (MPI.Bcast!, a file, ...)"
function comm_unique_id()
    id = zeros(UInt8, 128)
    check(ccall((:tpg_comm_unique_id, libtripolar), Cint, (Ptr{UInt8},), id))
    return id
end

function SeamComm(id::Vector{UInt8}, rank::Integer, nranks::Integer)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:tpg_comm_init_rank, libtripolar), Cint, (Ref{Ptr{Cvoid}}, Cint, Ptr{UInt8}, Cint), h, nranks, id, rank))
    return SeamComm(h[], rank, nranks)
end

destroy!(c::SeamComm) = (check(ccall((:tpg_comm_destroy, libtripolar), Cint, (Ptr{Cvoid},), c.handle)); c.handle = C_NULL; nothing)

"""
    halo_exchange_y!(comm, fields, locs, grid; buffers = nothing)

The y-seam exchange of one `fill_halo_regions!` on a distributed tripolar grid: ONE RCCL send/recv group on the current
stream, no host wait.  `buffers = (send_south, send_north, recv_south, recv_north)` device arrays of
`tpg_y_halo_buffer_elems` elements select the packed form; `nothing` the pack-free form (per-level seam windows sent
from / received into the fields).  Call after the zipper (last rank) and the periodic-x pass of the same fill.
"""
function halo_exchange_y!(comm::SeamComm, fields, locs, grid; buffers = nothing)
    Nx, Ny, _ = size(grid)
    Hx, Hy, _ = halo_size(grid)
    for ((FT, Nz, Hz), idx) in zipper_groups(fields, locs, grid)
        fs   = [fields[n] for n in idx]
        ptrs = Ptr{Cvoid}[device_pointer(f) for f in fs]
        bp   = isnothing(buffers) ? ntuple(_ -> C_NULL, 4) : map(b -> isnothing(b) ? C_NULL : device_pointer(b), buffers)
        GC.@preserve fs buffers begin
            check(ccall((:tpg_halo_exchange_y, libtripolar), Cint,
                        (Ptr{Cvoid}, Cint, Cint, Ptr{Ptr{Cvoid}}, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid},
                         Cint, Cint, Cint, Cint, Cint, Cint, Cint, Ptr{Cvoid}),
                        comm.handle, comm.rank, comm.nranks, ptrs, length(fs), bp[1], bp[2], bp[3], bp[4],
                        Nx, Ny, Nz, Hx, Hy, Hz, ft_code(FT), current_stream()))
        end
    end
    return nothing
end

y_halo_buffer_elems(nfields, grid, Nz, Hz) =
    Int(ccall((:tpg_y_halo_buffer_elems, libtripolar), Csize_t, (Cint, Cint, Cint, Cint, Cint, Cint),
              nfields, size(grid, 1), Nz, halo_size(grid)[1], halo_size(grid)[2], Hz))

# ---------------------------------------------------------------------------------------------------------------------
# 7. Geometry utilities                 test/test_tripolar_grid.jl:8-34,70; examples/convert_to_latlong_frame.jl:12-55
# ---------------------------------------------------------------------------------------------------------------------
"angle (degrees, minus 90) between the grid lines through every Face-Face node; `immersed`: dense Nx x Ny UInt8 device array or nothing"
function nonorthogonality_angle!(angle, grid::TripolarGrid; immersed = nothing)
    Nx, Ny, _ = size(grid)
    Hx, Hy, _ = halo_size(grid)
    GC.@preserve angle immersed begin
        check(ccall((:tpg_nonorthogonality_angle, libtripolar), Cint,
                    (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cint, Cint, Cint, Cint, Cint, Ptr{Cvoid}),
                    device_pointer(grid.λᶠᶠᵃ), device_pointer(grid.φᶠᶠᵃ), isnothing(immersed) ? C_NULL : device_pointer(immersed),
                    device_pointer(angle), Nx, Ny, Hx, Hy, ft_code(eltype(grid)), current_stream()))
    end
    return angle
end

function convert_frame!(u_out, v_out, u, v, grid::TripolarGrid; to_native::Bool = false)
    Nx, Ny, Nz = size(grid)
    Hx, Hy, Hz = halo_size(grid)
    GC.@preserve u_out v_out u v begin
        check(ccall((:tpg_convert_frame, libtripolar), Cint,
                    (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cint,
                     Cint, Cint, Cint, Cint, Cint, Cint, Cint, Ptr{Cvoid}),
                    device_pointer(grid.φᶜᶠᵃ), device_pointer(grid.φᶠᶜᵃ), device_pointer(grid.Δyᶜᶜᵃ), device_pointer(grid.Δxᶜᶜᵃ),
                    device_pointer(u), device_pointer(v), device_pointer(u_out), device_pointer(v_out), to_native ? 1 : 0,
                    Nx, Ny, Nz, Hx, Hy, Hz, ft_code(eltype(grid)), current_stream()))
    end
    return u_out, v_out
end
convert_to_latlong_frame!(u_out, v_out, u, v, grid) = convert_frame!(u_out, v_out, u, v, grid; to_native = false)
convert_to_native_frame!(u_out, v_out, u, v, grid)  = convert_frame!(u_out, v_out, u, v, grid; to_native = true)

end # module
