# TripolarHIP.jl -- thin Julia glue over libtripolar_hip.so (include/tripolar_hip.h).
#
# NOT exercised in the build container (no Julia toolchain, SURVEY.md 8c): this file is the
# reference-side binding a maintainer would add so that Oceananigans keeps seeing
# TripolarGrid() / ZipperBoundaryCondition / fill_halo_regions! while the numerics run in the
# hand-written HIP kernels.  No CUDA.jl, no KernelAbstractions / AMDGPU.jl code generation:
# device memory is reached through raw pointers (any array type `A` for which `device_pointer(A)`
# returns a Ptr{Cvoid} into HBM, e.g. an AMDGPU.ROCArray or a hipMalloc-backed wrapper).
#
# Each method below names the reference method it replaces (file:line in
# CliMA/OrthogonalSphericalShellGrids.jl v0.2.1).
module TripolarHIP

using Oceananigans
using Oceananigans.Grids: R_Earth, RightConnected, FullyConnected, OrthogonalSphericalShellGrid,
                          generate_coordinate
using OffsetArrays

const libtripolar = get(ENV, "LIBTRIPOLAR_HIP", "libtripolar_hip.so")

# ---------------------------------------------------------------------------------------------
# C structs / enums (include/tripolar_hip.h)
# ---------------------------------------------------------------------------------------------
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
    # same exception types as the reference: ArgumentError for odd Nlambda (tripolar_grid.jl:81-83)
    # and for a non-y partition (distributed_tripolar_grid.jl:28-31)
    (status == -2 || status == -3) && throw(ArgumentError(msg))
    throw(TripolarHIPError(status, msg))
end

# device pointer / stream hooks: specialise for the array backend in use
device_pointer(a) = Ptr{Cvoid}(pointer(parent(a)))
current_stream() = C_NULL          # hipStream_t of the task; NULL = default stream

# ---------------------------------------------------------------------------------------------
# TripolarGrid(arch, FT; ...)   replaces src/tripolar_grid.jl:59-333
# ---------------------------------------------------------------------------------------------
struct Tripolar{N, F, S}           # src/tripolar_grid.jl:6-10
    north_poles_latitude::N
    first_pole_longitude::F
    southernmost_latitude::S
end

"""
    TripolarGrid(arch, FT = Float64; size, southernmost_latitude = -80, halo = (4, 4, 4),
                 radius = R_Earth, z = (0, 1), north_poles_latitude = 55, first_pole_longitude = 70,
                 jrange = (1, size[2]), allocate)

`allocate(FT, dims...)` must return a device array (HBM) of that shape; the 20 padded metric
arrays are filled by ONE `tpg_build_grid` call (no host passes, no H2D copies).
"""
function TripolarGrid(arch, FT::DataType = Float64; size, southernmost_latitude = -80,
                      halo = (4, 4, 4), radius = R_Earth, z = (0, 1), north_poles_latitude = 55,
                      first_pole_longitude = 70, jrange = (1, size[2]), allocate)
    Nλ, Nφ, Nz = size
    Hλ, Hφ, Hz = halo
    isodd(Nλ) && throw(ArgumentError("The number of cells in the longitude dimension should be even!"))
    jstart, jend = jrange
    ny = jend - jstart + 1

    p = Ref(TpgParams(Nλ, Nφ, Nz, Hλ, Hφ, Hz, southernmost_latitude, north_poles_latitude,
                      first_pole_longitude, radius, ft_code(FT), jstart, jend, 0))
    arrays = [allocate(FT, Nλ + 2Hλ, ny + 2Hφ) for _ in 1:20]
    nbytes = ccall((:tpg_build_grid_workspace_bytes, libtripolar), Csize_t, (Ref{TpgParams},), p)
    workspace = allocate(UInt8, Int(nbytes))
    ptrs = Ptr{Cvoid}[device_pointer(a) for a in arrays]
    GC.@preserve arrays workspace begin
        check(ccall((:tpg_build_grid, libtripolar), Cint,
                    (Ref{TpgParams}, Ptr{Ptr{Cvoid}}, Ptr{Cvoid}, Csize_t, Ptr{Cvoid}),
                    p, ptrs, device_pointer(workspace), nbytes, current_stream()))
    end
    # order of enum tpg_array == positional order of src/tripolar_grid.jl:308-328
    off(a) = OffsetArray(a, -Hλ, -Hφ)
    λcc, λfc, λcf, λff, φcc, φfc, φcf, φff,
    Δxcc, Δxfc, Δxcf, Δxff, Δycc, Δycf, Δyfc, Δyff, Azcc, Azfc, Azcf, Azff = off.(arrays)

    topology = (Periodic, RightConnected, Bounded)
    Lz, zc = generate_coordinate(FT, topology, size, halo, z, :z, 3, CPU())   # z stays with Oceananigans
    LY = jstart == 1 ? RightConnected : FullyConnected                         # distributed_tripolar_grid.jl:75
    return OrthogonalSphericalShellGrid{Periodic, LY, Bounded}(arch, Nλ, ny, Nz, Hλ, Hφ, Hz, convert(FT, Lz),
               λcc, λfc, λcf, λff, φcc, φfc, φcf, φff, Oceananigans.on_architecture(arch, zc),
               Δxcc, Δxfc, Δxcf, Δxff, Δycc, Δycf, Δyfc, Δyff, Azcc, Azfc, Azcf, Azff,
               convert(FT, radius), Tripolar(north_poles_latitude, first_pole_longitude, southernmost_latitude))
end

# ---------------------------------------------------------------------------------------------
# Zipper: metadata identical to src/zipper_boundary_condition.jl:8,52-64
# ---------------------------------------------------------------------------------------------
using Oceananigans.BoundaryConditions: AbstractBoundaryConditionClassification, BoundaryCondition
import Oceananigans.BoundaryConditions: bc_str

struct Zipper <: AbstractBoundaryConditionClassification end
ZipperBoundaryCondition(sign = 1) = BoundaryCondition(Zipper(), sign)
const ZBC = BoundaryCondition{<:Zipper}
bc_str(::ZBC) = "Zipper"

loc_code(::Center) = Int8(0)
loc_code(::Face)   = Int8(1)

"""
    zipper_fill!(fields::Vector, bcs::Vector{<:ZBC}, locs, grid)

Replaces the per-(i,k) `_fill_north_halo!(i, k, grid, c, bc::ZBC, loc, args...)`
(src/zipper_boundary_condition.jl:146-155) for a whole batch of fields with ONE kernel launch.
Hook: a method of Oceananigans' south/north halo launcher specialised on `north_bc::ZBC`
(Oceananigans-internal generic, version dependent: `fill_south_and_north_halo!` in 0.95-0.99)
collects the fields of a `fill_halo_regions!(fields...)` call and forwards them here.
"""
function zipper_fill!(fields::Vector, bcs::Vector, locs::Vector, grid)
    Nx, Ny, Nz = size(grid)
    Hx, Hy, Hz = Oceananigans.Grids.halo_size(grid)
    FT = eltype(parent(first(fields)))
    ptrs = Ptr{Cvoid}[device_pointer(f) for f in fields]
    xloc = Int8[loc_code(l[1]) for l in locs]
    yloc = Int8[loc_code(l[2]) for l in locs]
    sign = Int32[bc.condition for bc in bcs]
    GC.@preserve fields begin
        check(ccall((:tpg_zipper_fill, libtripolar), Cint,
                    (Ptr{Ptr{Cvoid}}, Cint, Ptr{Int8}, Ptr{Int8}, Ptr{Int32},
                     Cint, Cint, Cint, Cint, Cint, Cint, Cint, Cint, Cint, Ptr{Cvoid}),
                    ptrs, length(fields), xloc, yloc, sign, Nx, Ny, Nz, Hx, Hy, Hz, 1, Nz,
                    ft_code(FT), current_stream()))
    end
    return nothing
end

"""
    fill_zipper_and_periodic!(fields::Vector, bcs::Vector{<:ZBC}, locs, grid)

The whole `fill_halo_regions!` of fields on a serial tripolar grid (zipper, then periodic west / east:
order pinned by test/test_zipper_boundary_conditions.jl:42-45) in one call; small (2-D) fields such as
the split-explicit free surface take a single fused launch inside the library.
"""
function fill_zipper_and_periodic!(fields::Vector, bcs::Vector, locs::Vector, grid)
    Nx, Ny, Nz = size(grid)
    Hx, Hy, Hz = Oceananigans.Grids.halo_size(grid)
    FT = eltype(parent(first(fields)))
    ptrs = Ptr{Cvoid}[device_pointer(f) for f in fields]
    xloc = Int8[loc_code(l[1]) for l in locs]
    yloc = Int8[loc_code(l[2]) for l in locs]
    sign = Int32[bc.condition for bc in bcs]
    GC.@preserve fields begin
        check(ccall((:tpg_fill_halo_regions, libtripolar), Cint,
                    (Ptr{Ptr{Cvoid}}, Cint, Ptr{Int8}, Ptr{Int8}, Ptr{Int32},
                     Cint, Cint, Cint, Cint, Cint, Cint, Cint, Cint, Ptr{Cvoid}),
                    ptrs, length(fields), xloc, yloc, sign, Nx, Ny, Nz, Hx, Hy, Hz, 1,
                    ft_code(FT), current_stream()))
    end
    return nothing
end

"""
    pack_y_halo!(buffer, fields, side, grid) / unpack_y_halo!(fields, buffer, side, grid)

Device-side gather / scatter of the Hy seam rows of a y-slab partition; the transport (MPI.jl
Isend/Irecv on ROCm-aware MPI, or RCCL) stays in Oceananigans' DistributedComputations, which the
reference reaches from src/distributed_tripolar_grid.jl:171,195.
"""
function pack_y_halo!(buffer, fields::Vector, side::Integer, grid; pack::Bool = true)
    Nx, Ny, Nz = size(grid)
    Hx, Hy, Hz = Oceananigans.Grids.halo_size(grid)
    FT = eltype(parent(first(fields)))
    ptrs = Ptr{Cvoid}[device_pointer(f) for f in fields]
    GC.@preserve fields buffer begin
        # (the symbol of a ccall must be a literal: two call sites, one per direction)
        status = pack ?
            ccall((:tpg_pack_y_halo, libtripolar), Cint,
                  (Ptr{Ptr{Cvoid}}, Cint, Ptr{Cvoid}, Cint, Cint, Cint, Cint, Cint, Cint, Cint, Cint, Ptr{Cvoid}),
                  ptrs, length(fields), device_pointer(buffer), side, Nx, Ny, Nz, Hx, Hy, Hz, ft_code(FT), current_stream()) :
            ccall((:tpg_unpack_y_halo, libtripolar), Cint,
                  (Ptr{Ptr{Cvoid}}, Cint, Ptr{Cvoid}, Cint, Cint, Cint, Cint, Cint, Cint, Cint, Cint, Ptr{Cvoid}),
                  ptrs, length(fields), device_pointer(buffer), side, Nx, Ny, Nz, Hx, Hy, Hz, ft_code(FT), current_stream())
        check(status)
    end
    return nothing
end
unpack_y_halo!(fields, buffer, side, grid) = pack_y_halo!(buffer, fields, side, grid; pack = false)

end # module
