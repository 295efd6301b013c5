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
# No CUDA.jl, no KernelAbstractions / AMDGPU.jl code generation.  Device memory is OWNED here: `HIPArray` (section 1b) is a
# minimal DenseArray over hipMalloc / hipFree / hipMemcpy, called through `ccall` on libamdhip64, and `HIPGPU()` is the
# Oceananigans architecture value that carries it.  `device_pointer` accepts nothing but (views / OffsetArrays of) a HIPArray: no
# path in this file can hand a host pointer or a CuArray to a `tpg_*` entry point.
#
# LIMIT of HIPGPU(): it is an architecture for THIS path only.  It has no KernelAbstractions backend, so Oceananigans' own kernels
# (tendencies, its south / bottom / top halo fills, generic `fill_halo_regions!`) cannot launch on a grid built there; every place in
# this file that would have to hand a field to them on HIPGPU() raises an ArgumentError that says so instead of a MethodError.  A
# whole-model caller keeps its grid and fields in its own backend's device arrays and reaches the `tpg_*` entry points through the three
# hooks `device_pointer` / `device_array` / `stream_for`, which ext/TripolarHIPBackendExt.jl (a weak-dependency extension, never
# imported here) defines for that backend: pointer and stream extraction only, no kernels.
module TripolarHIP

export TripolarGrid, ZipperBoundaryCondition           # src/OrthogonalSphericalShellGrids.jl:4
export HIPArray, HIPGPU                                # device memory + architecture of this binding (no reference counterpart)

using Oceananigans
using Oceananigans.Architectures: AbstractArchitecture, CPU, child_architecture
import Oceananigans.Architectures: architecture, on_architecture, array_type
using Oceananigans.Grids: AbstractGrid
using Oceananigans.DistributedComputations: DistributedGrid
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
import MPI                                   # a declared dependency of the reference (Project.toml:9): ferries the 128-byte RCCL id

import Oceananigans.BoundaryConditions: bc_str, apply_y_north_bc!, regularize_field_boundary_conditions
import Oceananigans.Fields: Field, validate_boundary_condition_location
import Oceananigans.Grids: x_domain, y_domain, with_halo
import Oceananigans.DistributedComputations: reconstruct_global_grid

const libtripolar = get(ENV, "LIBTRIPOLAR_HIP", "libtripolar_hip.so")
const libhip      = get(ENV, "LIBAMDHIP64", "libamdhip64.so")      # the HIP runtime itself (/opt/rocm/lib); plain C entry points only

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

# ---------------------------------------------------------------------------------------------------------------------
# 1b. Device memory without CUDA.jl / AMDGPU.jl: HIPArray, HIPGPU            replaces on_architecture(arch, map(FT, A)) x 20,
#                                                                             src/tripolar_grid.jl:303-328
# ---------------------------------------------------------------------------------------------------------------------
# The reference builds its 20 arrays on the CPU and hands them over with `on_architecture(arch, ...)`, which for Oceananigans'
# `GPU()` means CuArray (CUDA.jl) -- absent on an MI355X host -- and for `CPU()` a host Array whose pointer a HIP kernel must never
# see.  Here the arrays are born in HBM: `HIPArray{T,N}` owns one hipMalloc allocation (freed by a finalizer through hipFree) and
# is all the array type this path needs -- size / pointer / unsafe_convert for `ccall`, copies to and from host Arrays through
# hipMemcpy, `Adapt` rules; scalar indexing is refused (it would be a PCIe round trip per element).  Prototypes as in
# /opt/rocm/include/hip/hip_runtime_api.h (checked statically by tests/test_julia_glue_static.py):
#   hipError_t hipMalloc(void** ptr, size_t size);             hipError_t hipFree(void* ptr);
#   hipError_t hipMemcpy(void* dst, const void* src, size_t sizeBytes, hipMemcpyKind kind);
#   hipError_t hipMemset(void* dst, int value, size_t sizeBytes);
#   hipError_t hipStreamCreateWithFlags(hipStream_t* stream, unsigned int flags);
#   hipError_t hipStreamSynchronize(hipStream_t stream);       hipError_t hipDeviceSynchronize(void);
#   hipError_t hipStreamDestroy(hipStream_t stream);
#   const char* hipGetErrorString(hipError_t hipError);
const hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice = Cint(1), Cint(2), Cint(3)   # enum hipMemcpyKind
const hipStreamNonBlocking = Cuint(1)

function hipcheck(status::Cint)
    status == 0 && return nothing
    throw(TripolarHIPError(status, unsafe_string(ccall((:hipGetErrorString, libhip), Cstring, (Cint,), status))))
end

mutable struct HIPArray{T, N} <: DenseArray{T, N}
    ptr::Ptr{Cvoid}                    # device address (hipMalloc); C_NULL once freed or for an empty array
    dims::NTuple{N, Int}
    function HIPArray{T, N}(::UndefInitializer, dims::NTuple{N, Int}) where {T, N}
        isbitstype(T) || throw(ArgumentError("HIPArray: element type $T is not a plain bits type"))
        nbytes = prod(dims) * sizeof(T)
        p = Ref{Ptr{Cvoid}}(C_NULL)
        nbytes > 0 && hipcheck(ccall((:hipMalloc, libhip), Cint, (Ref{Ptr{Cvoid}}, Csize_t), p, nbytes))
        a = new{T, N}(p[], dims)
        finalizer(unsafe_free!, a)
        return a
    end
end

"return the allocation to the HIP runtime now (idempotent; also the finalizer)"
function unsafe_free!(a::HIPArray)
    a.ptr == C_NULL || ccall((:hipFree, libhip), Cint, (Ptr{Cvoid},), a.ptr)       # status ignored: a finalizer must not throw
    a.ptr = C_NULL
    return nothing
end

HIPArray{T, N}(::UndefInitializer, dims::Vararg{Integer, N}) where {T, N} = HIPArray{T, N}(undef, map(Int, dims))
HIPArray{T}(::UndefInitializer, dims::NTuple{N, Integer}) where {T, N} = HIPArray{T, N}(undef, map(Int, dims))
HIPArray{T}(::UndefInitializer, dims::Integer...) where {T} = HIPArray{T}(undef, dims)
HIPArray(a::Array{T, N}) where {T, N} = copyto!(HIPArray{T, N}(undef, size(a)), a)
HIPArray{T}(a::Array{S, N}) where {T, S, N} = HIPArray(convert(Array{T, N}, a))
Base.Array(a::HIPArray{T, N}) where {T, N} = copyto!(Array{T, N}(undef, size(a)), a)

Base.size(a::HIPArray) = a.dims
Base.sizeof(a::HIPArray{T}) where {T} = prod(a.dims) * sizeof(T)
Base.elsize(::Type{<:HIPArray{T}}) where {T} = sizeof(T)
Base.pointer(a::HIPArray{T}) where {T} = Ptr{T}(a.ptr)                              # a DEVICE address: for ccall only
Base.unsafe_convert(::Type{Ptr{T}}, a::HIPArray{T}) where {T} = Ptr{T}(a.ptr)
Base.unsafe_convert(::Type{Ptr{Cvoid}}, a::HIPArray) = a.ptr
Base.similar(a::HIPArray, ::Type{T}, dims::Dims{N}) where {T, N} = HIPArray{T, N}(undef, dims)
Base.getindex(::HIPArray, I...) = error("HIPArray: scalar indexing reads device memory element by element; copy with Array(a) first")
Base.setindex!(::HIPArray, v, I...) = error("HIPArray: scalar indexing writes device memory element by element; use copyto!(a, host_array)")
Base.show(io::IO, a::HIPArray{T, N}) where {T, N} = print(io, join(a.dims, "x"), " HIPArray{", T, ",", N, "} @ ", a.ptr)
Base.show(io::IO, ::MIME"text/plain", a::HIPArray) = show(io, a)

# copies are synchronous (hipMemcpy on the null stream): they order themselves after every kernel this library enqueued
function Base.copyto!(dst::HIPArray{T}, src::Array{T}) where {T}
    length(dst) == length(src) || throw(DimensionMismatch("copyto!: $(size(src)) -> $(size(dst))"))
    GC.@preserve dst src hipcheck(ccall((:hipMemcpy, libhip), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Csize_t, Cint),
                                        dst.ptr, pointer(src), sizeof(src), hipMemcpyHostToDevice))
    return dst
end
function Base.copyto!(dst::Array{T}, src::HIPArray{T}) where {T}
    length(dst) == length(src) || throw(DimensionMismatch("copyto!: $(size(src)) -> $(size(dst))"))
    GC.@preserve dst src hipcheck(ccall((:hipMemcpy, libhip), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Csize_t, Cint),
                                        pointer(dst), src.ptr, sizeof(dst), hipMemcpyDeviceToHost))
    return dst
end
function Base.copyto!(dst::HIPArray{T}, src::HIPArray{T}) where {T}
    length(dst) == length(src) || throw(DimensionMismatch("copyto!: $(size(src)) -> $(size(dst))"))
    GC.@preserve dst src hipcheck(ccall((:hipMemcpy, libhip), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Csize_t, Cint),
                                        dst.ptr, src.ptr, sizeof(dst), hipMemcpyDeviceToDevice))
    return dst
end
Base.copy(a::HIPArray{T, N}) where {T, N} = copyto!(HIPArray{T, N}(undef, size(a)), a)
function Base.fill!(a::HIPArray{T}, v) where {T}                                       # zeros only need hipMemset; other values go through the host
    if iszero(v)
        sizeof(a) > 0 && hipcheck(ccall((:hipMemset, libhip), Cint, (Ptr{Cvoid}, Cint, Csize_t), a.ptr, 0, sizeof(a)))
        return a
    end
    return copyto!(a, fill(convert(T, v), size(a)))
end
device_synchronize() = hipcheck(ccall((:hipDeviceSynchronize, libhip), Cint, ()))

# host <-> device conversion rules (what `on_architecture` and Adapt-based constructors use)
Adapt.adapt_storage(::Type{<:HIPArray}, a::Array) = HIPArray(a)
Adapt.adapt_storage(::Type{<:Array}, a::HIPArray) = Array(a)
Adapt.adapt_storage(::Type{<:HIPArray}, a::HIPArray) = a

"""
    HIPGPU()

The Oceananigans architecture value of this binding: one MI355X reached through the HIP runtime and libtripolar_hip, with
`HIPArray` as its array type.  `TripolarGrid(HIPGPU(); size = ...)` (and `Distributed(HIPGPU(); partition = Partition(1, R))`)
builds the grid in HBM; `on_architecture(CPU(), grid)` brings it back.  Oceananigans' own `GPU()` is CUDA.jl's (0.95-0.99) and
is refused by `device_array`; `CPU()` is refused as well (a host pointer must never reach a HIP kernel).
"""
struct HIPGPU <: AbstractArchitecture end
array_type(::HIPGPU) = HIPArray                                                        # [recalled: Oceananigans.Architectures.array_type]
architecture(::HIPArray) = HIPGPU()
on_architecture(::HIPGPU, a::Array) = HIPArray(a)
on_architecture(::HIPGPU, a::HIPArray) = a
on_architecture(::CPU, a::HIPArray) = Array(a)
on_architecture(arch::HIPGPU, a::OffsetArray) = OffsetArray(on_architecture(arch, parent(a)), a.offsets...)
on_architecture(arch::HIPGPU, a::AbstractRange) = on_architecture(arch, collect(a))

# backend hooks -------------------------------------------------------------------------------------------------------
# raw HBM address of the memory behind `a`: a HIPArray, or an OffsetArray / reshape whose root parent is one.
# Anything else -- a host Array, a CuArray, a ROCArray this file did not allocate -- is refused: the C ABI dereferences the
# address on the device.
device_pointer(a::HIPArray) = (a.ptr == C_NULL && length(a) > 0) ? error("HIPArray used after unsafe_free!") : a.ptr
device_pointer(a::OffsetArray) = device_pointer(parent(a))
device_pointer(a::Base.ReshapedArray) = device_pointer(parent(a))
device_pointer(a) = throw(ArgumentError("libtripolar_hip needs device memory owned by a HIPArray (HIPGPU() architecture); got $(typeof(a)): " *
                                        "host Arrays and CuArrays are refused"))

# hipStream_t of the calling Julia task: one non-blocking stream per task, created at first use (fills issued from different
# tasks run on different streams and never share seam buffers, section 5); `use_default_stream!()` selects the NULL stream instead.
# The handle lives in the task's own `task_local_storage()` -- no table shared between threads, nothing keeps a finished Task alive --
# and its finalizer returns the stream to the runtime (hipError_t hipStreamDestroy(hipStream_t stream)) when the task is collected.
const USE_DEFAULT_STREAM = Ref(true)
use_default_stream!(flag::Bool = true) = (USE_DEFAULT_STREAM[] = flag)
function new_stream()
    s = Ref{Ptr{Cvoid}}(C_NULL)
    hipcheck(ccall((:hipStreamCreateWithFlags, libhip), Cint, (Ref{Ptr{Cvoid}}, Cuint), s, hipStreamNonBlocking))
    return s[]
end
mutable struct StreamHandle
    ptr::Ptr{Cvoid}
    function StreamHandle()
        h = new(new_stream())
        finalizer(destroy_stream!, h)
        return h
    end
end
function destroy_stream!(h::StreamHandle)
    h.ptr == C_NULL || ccall((:hipStreamDestroy, libhip), Cint, (Ptr{Cvoid},), h.ptr)       # status ignored: a finalizer must not throw
    h.ptr = C_NULL
    return nothing
end
task_stream(key::Symbol) = (get!(StreamHandle, task_local_storage(), key)::StreamHandle).ptr
current_stream() = USE_DEFAULT_STREAM[] ? C_NULL : task_stream(:tripolar_hip_stream)
synchronize_stream(s = current_stream()) = hipcheck(ccall((:hipStreamSynchronize, libhip), Cint, (Ptr{Cvoid},), s))
# the stream this library enqueues on for arrays of `arch`: this file's own task stream for HIPGPU(); a backend extension returns its
# backend's current stream handle, so that the `tpg_*` kernels order themselves with the host model's own kernels
serial_arch(arch) = arch                                        # the per-process architecture under a Distributed wrapper
serial_arch(arch::Distributed) = child_architecture(arch)
stream_for(arch) = current_stream()
stream_for(arch::Distributed) = stream_for(serial_arch(arch))

# process-wide tables of this file (seam communicators, table workspaces) are guarded by one lock: fills may come from several threads
const STATE_LOCK = ReentrantLock()

# An UNINITIALISED device array (HBM) of that shape: tpg_build_grid overwrites every element of the 20 arrays, halos included,
# so nothing is zeroed on the host and nothing crosses PCIe.
device_array(::HIPGPU, FT, dims...) = HIPArray{FT}(undef, dims...)
device_array(arch::Distributed, FT, dims...) = device_array(child_architecture(arch), FT, dims...)
device_array(arch, FT, dims...) = throw(ArgumentError("TripolarHIP builds grids on HIPGPU() (or Distributed(HIPGPU(); ...)) only; got $(typeof(arch)): " *
                                                      "Oceananigans' GPU() is CUDA.jl's and CPU() memory cannot be given to a HIP kernel"))

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
# The 1-D tables tpg_build_grid leaves in its workspace, kept for later builds of the same geometry (TPG_BUILD_TABLES_VALID,
# include/tripolar_hip.h): `key` = exactly what the tables depend on -- global Nx, Ny, Hy, element type, southernmost latitude,
# north-poles latitude, radius -- plus the architecture; jstart / jend, Hx, Hz, Nz and first_pole_longitude do not enter.  A workspace
# lives as long as a grid built on it: GRID_WORKSPACES holds (WeakRef(the grid's lambda_cc parent array), workspace) entries, so with_halo
# (same size, new Hx / Hz: src/with_halo.jl:5-44), reconstruct_global_grid after a band build (src/distributed_tripolar_grid.jl:201-226)
# and consecutive band builds of one geometry find the tables of the first build, and nothing outlives the grids.  A build whose key
# differs gets a NEW workspace: the tables of a live grid are never overwritten.
# Reuse is EXPLICIT (as in the Python host, grids.py): with_halo and reconstruct_global_grid hand the old grid's workspace to the new build
# (`tables_from`), always; finding the tables of ANY live grid of the same geometry happens only inside `share_tables() do ... end`, so that
# outside it a build never depends on which other grids happen to be alive.
# The owner array is matched by IDENTITY (===) only: a dictionary keyed by a device array would hash / compare its ELEMENTS
# (Base.hash(::AbstractArray) indexes them), which a HIPArray refuses (scalar indexing) and any device array makes a PCIe round trip.
const TPG_BUILD_TABLES_VALID = Int32(1)
mutable struct TableWorkspace
    key::Any
    buffer::Any                    # device array of tpg_build_grid_workspace_bytes
    stream::Ptr{Cvoid}             # the stream the table kernel ran on
end
struct WorkspaceOwner
    owner::WeakRef                 # the grid's lambda_cc parent array: compared with === only, never hashed, never indexed
    workspace::TableWorkspace
end
const GRID_WORKSPACES = WorkspaceOwner[]
table_key(arch, FT, Nλ, Nφ, Hφ, south, npl, radius) = (serial_arch(arch), FT, Int(Nλ), Int(Nφ), Int(Hφ), Float64(south), Float64(npl), Float64(radius))
"entries whose grid is gone leave the table (callers hold STATE_LOCK)"
prune_workspaces!(table) = filter!(e -> e.owner.value !== nothing, table)
function live_workspace(key, nbytes)
    lock(STATE_LOCK) do
        for e in prune_workspaces!(GRID_WORKSPACES)
            e.workspace.key == key && sizeof(e.workspace.buffer) >= nbytes && return e.workspace
        end
        return nothing
    end
end
"`share_tables() do ... end`: builds inside (this task only) may take the 1-D tables of any live grid of the same geometry (e.g. the bands of an emulated chain)"
share_tables(f) = task_local_storage(f, :tpg_share_tables, true)
sharing_tables() = get(task_local_storage(), :tpg_share_tables, false)

function keep_workspace!(owner, ws::TableWorkspace)
    lock(STATE_LOCK) do
        push!(prune_workspaces!(GRID_WORKSPACES), WorkspaceOwner(WeakRef(owner), ws))
    end
    return ws
end
function table_workspace(grid)
    a = parent(grid.λᶜᶜᵃ)
    lock(STATE_LOCK) do
        for e in GRID_WORKSPACES
            e.owner.value === a && return e.workspace
        end
        return nothing
    end
end

# ONE tpg_build_grid call fills the 20 padded arrays of the latitude band jstart:jend in HBM (no host passes, no H2D).
function build_band(arch, FT, size, halo, southernmost_latitude, radius, z, north_poles_latitude, first_pole_longitude,
                    jstart, jend, LY; tables_from = nothing)
    Nλ, Nφ, Nz = size
    Hλ, Hφ, Hz = halo
    isodd(Nλ) && throw(ArgumentError("The number of cells in the longitude dimension should be even!"))   # :81-83
    ny = jend - jstart + 1
    arrays = [device_array(arch, FT, Nλ + 2Hλ, ny + 2Hφ) for _ in 1:20]
    p0 = Ref(TpgParams(Nλ, Nφ, Nz, Hλ, Hφ, Hz, southernmost_latitude, north_poles_latitude, first_pole_longitude,
                       radius, ft_code(FT), jstart, jend, 0))
    nbytes = ccall((:tpg_build_grid_workspace_bytes, libtripolar), Csize_t, (Ref{TpgParams},), p0)
    s   = stream_for(arch)
    key = table_key(arch, FT, Nλ, Nφ, Hφ, southernmost_latitude, north_poles_latitude, radius)
    # the tables of the grid we were derived from (explicit hand-over), else -- inside share_tables() only -- of any live grid of this key
    ws  = (tables_from !== nothing && tables_from.key == key && sizeof(tables_from.buffer) >= nbytes) ? tables_from :
          (sharing_tables() ? live_workspace(key, nbytes) : nothing)
    reuse = ws !== nothing
    if reuse
        ws.stream == s || synchronize_stream(ws.stream)          # tables written on another stream: wait for them (a build is a one-off)
    else
        ws = TableWorkspace(key, device_array(arch, UInt8, Int(nbytes)), s)
    end
    p = Ref(TpgParams(Nλ, Nφ, Nz, Hλ, Hφ, Hz, southernmost_latitude, north_poles_latitude, first_pole_longitude,
                      radius, ft_code(FT), jstart, jend, reuse ? TPG_BUILD_TABLES_VALID : Int32(0)))
    workspace = ws.buffer
    ptrs = Ptr{Cvoid}[device_pointer(a) for a in arrays]
    GC.@preserve arrays workspace begin
        check(ccall((:tpg_build_grid, libtripolar), Cint,
                    (Ref{TpgParams}, Ptr{Ptr{Cvoid}}, Ptr{Cvoid}, Csize_t, Ptr{Cvoid}),
                    p, ptrs, device_pointer(workspace), nbytes, s))
    end
    keep_workspace!(arrays[1], ws)                                # arrays[1] = parent of lambda_cc: the grid keeps its tables alive
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
    TripolarGrid(arch = HIPGPU(), FT = Float64; size, southernmost_latitude = -80, halo = (4, 4, 4),
                 radius = R_Earth, z = (0, 1), north_poles_latitude = 55, first_pole_longitude = 70)

Same keywords, defaults, return type and `ArgumentError` as src/tripolar_grid.jl:59-66,81-83,304-330; the default architecture
is `HIPGPU()` (the reference's is `CPU()`, which this binding refuses: see `device_array`).
"""
function TripolarGrid(arch::AbstractArchitecture = HIPGPU(), FT::DataType = Float64; size, southernmost_latitude = -80,
                      halo = (4, 4, 4), radius = R_Earth, z = (0, 1), north_poles_latitude = 55, first_pole_longitude = 70,
                      _tables_from = nothing)                      # not a reference keyword: with_halo / reconstruct_global_grid pass the old grid's tables
    return build_band(arch, FT, size, halo, southernmost_latitude, radius, z, north_poles_latitude,
                      first_pole_longitude, 1, size[2], RightConnected; tables_from = _tables_from)
end

"""
    TripolarGrid(arch::Distributed, FT = Float64; halo = (4, 4, 4), kwargs...)

src/distributed_tripolar_grid.jl:24-110: y-partitioning only; the rank's band is evaluated directly on its device (the
reference builds the whole globe on every rank's CPU and slices it).
"""
function TripolarGrid(arch::Distributed, FT::DataType = Float64; halo = (4, 4, 4), size, southernmost_latitude = -80,
                      radius = R_Earth, z = (0, 1), north_poles_latitude = 55, first_pole_longitude = 70, _tables_from = nothing)
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
                      first_pole_longitude, jstart, jend, LY; tables_from = _tables_from)
end

# src/tripolar_grid_extensions.jl:20-21
x_domain(grid::TRG) = 0, 360
y_domain(grid::TRG) = minimum(parent(grid.φᶠᶠᵃ)), 90

# src/with_halo.jl:5-23 (serial) and :25-44 (distributed: `radius` is not forwarded there -- kept).  The old grid's 1-D tables are handed
# to the constructor (`_tables_from`), which builds with TPG_BUILD_TABLES_VALID when only Hx / Hz change; a new Hy changes the key.
function with_halo(new_halo, old_grid::TripolarGrid)
    cm = old_grid.conformal_mapping
    return TripolarGrid(architecture(old_grid), eltype(old_grid); size = (old_grid.Nx, old_grid.Ny, old_grid.Nz),
                        z = cpu_face_constructor_z(old_grid), halo = new_halo, radius = old_grid.radius,
                        north_poles_latitude = cm.north_poles_latitude, first_pole_longitude = cm.first_pole_longitude,
                        southernmost_latitude = cm.southernmost_latitude, _tables_from = table_workspace(old_grid))
end

function with_halo(new_halo, old_grid::DistributedTripolarGrid)
    arch = old_grid.architecture
    N  = map(sum, concatenate_local_sizes(size(old_grid), arch))
    cm = old_grid.conformal_mapping
    return TripolarGrid(arch, eltype(old_grid); halo = new_halo, size = N, z = cpu_face_constructor_z(old_grid),
                        north_poles_latitude = cm.north_poles_latitude, first_pole_longitude = cm.first_pole_longitude,
                        southernmost_latitude = cm.southernmost_latitude, _tables_from = table_workspace(old_grid))
end

# src/distributed_tripolar_grid.jl:201-226; the band's 1-D tables are the globe's (indexed by global row): the global build reuses them
function reconstruct_global_grid(grid::DistributedTripolarGrid)
    arch = grid.architecture
    cm = grid.conformal_mapping
    return TripolarGrid(child_architecture(arch), eltype(grid); halo = halo_size(grid),
                        size = map(sum, concatenate_local_sizes(size(grid), arch)), z = cpu_face_constructor_z(grid),
                        north_poles_latitude = cm.north_poles_latitude, first_pole_longitude = cm.first_pole_longitude,
                        southernmost_latitude = cm.southernmost_latitude, _tables_from = table_workspace(grid))
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
# 5. Latitude-band seam communicator (RCCL through the C ABI)          src/distributed_tripolar_grid.jl:171,195 (transport)
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

# One communicator per Distributed architecture, created at the first distributed fill: every rank first reports whether it can
# bind librccl (tpg_comm_available: no collective call) and the ranks agree BEFORE the collective ncclCommInitRank; the 128-byte
# id travels over the architecture's own MPI communicator (`arch.communicator`, `MPI.Allreduce` / `MPI.Bcast!` [recalled]).
const SEAM_COMMS = IdDict{Any, SeamComm}()
function new_seam_comm(arch::Distributed)
    ok = ccall((:tpg_comm_available, libtripolar), Cint, ()) == 0 ? 1 : 0
    MPI.Allreduce(ok, MPI.MIN, arch.communicator) == 1 || error("librccl cannot be bound on every rank: no seam communicator")
    id = arch.local_rank == 0 ? comm_unique_id() : zeros(UInt8, 128)
    MPI.Bcast!(id, 0, arch.communicator)
    return SeamComm(id, arch.local_rank, ranks(arch.partition)[2])
end
seam_comm(arch::Distributed) = lock(() -> get!(() -> new_seam_comm(arch), SEAM_COMMS, arch), STATE_LOCK)

y_halo_buffer_elems(nfields, grid, Nz, Hz) =
    Int(ccall((:tpg_y_halo_buffer_elems, libtripolar), Csize_t, (Cint, Cint, Cint, Cint, Cint, Cint),
              nfields, size(grid, 1), Nz, halo_size(grid)[1], halo_size(grid)[2], Hz))

# message buffers of one (architecture, nfields, element type, Nz, Hz) fill, kept across fills in the TASK's own storage: fills issued
# from different tasks (= different streams) never share staging memory, and the buffers go when their task goes
function seam_buffers(arch, grid, nfields, FT, Nz, Hz)
    get!(task_local_storage(), (:tripolar_hip_seam_buffers, arch, nfields, FT, Nz, Hz, size(grid), halo_size(grid))) do
        n = y_halo_buffer_elems(nfields, grid, Nz, Hz)
        ntuple(_ -> device_array(arch, FT, n), 4)
    end
end

# ---------------------------------------------------------------------------------------------------------------------
# 6. fill_halo_regions! on a tripolar grid                               src/zipper_boundary_condition.jl:140-155
# ---------------------------------------------------------------------------------------------------------------------
# The reference's `_fill_north_halo!(i, k, grid, c, bc::ZBC, loc, args...)` is a per-thread function inlined into
# Oceananigans' south/north halo kernel; a ccall cannot live there.  The interception point is the method family where
# Oceananigans hands over ALL sides of a field: `fill_halo_regions!(c::OffsetArray, bcs, indices, loc, grid, args...)`
# (BoundaryConditions/fill_halo_regions.jl; for a DistributedGrid the method with the extra `buffers` argument,
# DistributedComputations/halo_communication.jl) [recalled signatures, Oceananigans 0.95-0.99].  Specialising them on
# `grid::TRG` / `grid::DTRG` takes the zipper, the periodic x pass AND the latitude-band seams away from KernelAbstractions
# and MPI: they become tpg_fill_halo_regions / tpg_fill_halo_regions_distributed (hand-written HIP + RCCL).
#
# Order.  Oceananigans fills the three side pairs in the order given by `permute_boundary_conditions`: periodic and
# communicating pairs LAST [recalled], i.e. on a tripolar field  south/north (the zipper)  ->  bottom/top  ->  west/east
# (periodic; pinned against the zipper by test/test_zipper_boundary_conditions.jl:42-45)  [-> seams].  The bottom/top
# conditions are Oceananigans' own (not part of the reference): when a field has none -- the 2-D free-surface and barotropic
# fields, `bottom_height`, the reference's own coordinate / metric fills (src/tripolar_grid.jl:137-199,230-273) -- the whole
# fill is ONE C call (one fused or merged launch); when it has some, the z pass must sit between the fold and the periodic
# pass exactly as in the reference, so the fill is  tpg_zipper_fill -> Oceananigans' bottom/top launcher -> tpg_periodic_x_fill
# (-> tpg_halo_exchange_y).
loc_code(::Center) = Int8(0);  loc_code(::Type{Center}) = Int8(0)
loc_code(::Face)   = Int8(1);  loc_code(::Type{Face})   = Int8(1)
loc_code(::Nothing) = Int8(0); loc_code(::Type{Nothing}) = Int8(0)

# (Nz, Hz) of ONE field from its own parent, z-location and indices -- never the grid's: a reduced field (LZ = Nothing, e.g.
# bottom_height at (Center, Center, Nothing), test/test_zipper_boundary_conditions.jl:47-54) and a field windowed in z
# (indices[3] a range: the parent holds exactly those levels, no z halo) have Hz = 0; a z-Face field has Nz + 1 levels.
function field_levels(c, loc, indices, grid)
    nlev = size(parent(c), 3)
    LZ = loc[3]
    (LZ === Nothing || LZ isa Nothing || !(indices[3] isa Colon)) && return nlev, 0
    Hz = halo_size(grid)[3]
    return nlev - 2Hz, Hz                              # Center: Nz, Face (Bounded z): Nz + 1
end

zipper_sign(bc::ZBC) = Int32(bc.condition)
is_periodic(bc) = bc isa BoundaryCondition{<:Oceananigans.BoundaryConditions.Periodic}
full_xy(indices) = indices[1] isa Colon && indices[2] isa Colon          # the kernels address the whole padded (x, y) parent

"can the C ABI take this field's fill?  Periodic x, a Zipper (last rank / serial) or a seam on the north side, whole (x, y)"
hip_fill_applies(bcs, indices, zipper_expected) =
    full_xy(indices) && is_periodic(bcs.west) && is_periodic(bcs.east) && (!zipper_expected || bcs.north isa ZBC)

# one C call per group of fields that share (element type, Nz, Hz): a tupled fill mixes 3-D and reduced fields.
# The groups come back as a Vector of `key => field indices` in FIRST-APPEARANCE order, never as a Dict: on a latitude-band grid every
# group issues one RCCL send / recv group, and group(k) of a rank pairs with group(k) of its neighbour (include/tripolar_hip.h) -- the
# order must be a function of the argument list alone (the same on every rank), not of a hash table's iteration order.  The Python host
# groups the same way (fields.py: insertion order).
function fill_groups(fields, locs, indices, grid)
    groups = Pair{Tuple{DataType, Int, Int}, Vector{Int}}[]
    for (n, (c, loc)) in enumerate(zip(fields, locs))
        Nz, Hz = field_levels(c, loc, indices, grid)
        key = (eltype(parent(c)), Nz, Hz)
        at  = findfirst(g -> first(g) == key, groups)
        at === nothing ? push!(groups, key => [n]) : push!(last(groups[at]), n)
    end
    return groups
end

as_tuple(x::Tuple) = x
as_tuple(x) = (x,)

"""
    hip_fill!(fields, bcs, locs, indices, grid; stage, comm = nothing, arch = nothing)

The C-ABI part of one `fill_halo_regions!`, one batched call per geometry group.  `stage`:
  :all       zipper -> periodic x (-> seams)   tpg_fill_halo_regions / tpg_fill_halo_regions_distributed  (no z conditions)
  :zipper    the fold alone                     tpg_zipper_fill            (fold_north_*!, src/zipper_boundary_condition.jl:70-155)
  :periodic  periodic x (-> seams)              tpg_periodic_x_fill (+ tpg_halo_exchange_y)
"""
const TPG_MAX_FIELDS = 16                      # include/tripolar_hip.h: fields per seam message / per distributed call
# Seam exchange form of distributed fills: 0 = monolithic (pack all -> ONE RCCL group -> unpack all); k > 0 = pipelined in stages of
# k fields with the RCCL groups on a second stream (tpg_fill_halo_regions_distributed_pipelined); identical results
const SEAM_FIELDS_PER_STAGE = Ref(0)
seam_exchange_pipelined!(fields_per_stage::Integer = 1) = (SEAM_FIELDS_PER_STAGE[] = Int(fields_per_stage))
comm_stream() = task_stream(:tripolar_hip_comm_stream)          # the second stream of the pipelined exchange, one per Julia task
# EVERY RANK must select the same form: group(k) of a rank pairs with group(k) of its neighbour (include/tripolar_hip.h).  The value is a
# per-process setting; `seam_exchange_pipelined!(k, arch)` with a Distributed architecture checks the agreement once, collectively.
function seam_exchange_pipelined!(fields_per_stage::Integer, arch::Distributed)
    k = Int(fields_per_stage)
    lo, hi = MPI.Allreduce(k, MPI.MIN, arch.communicator), MPI.Allreduce(k, MPI.MAX, arch.communicator)
    lo == hi || throw(ArgumentError("seam_exchange_pipelined!: ranks disagree on fields_per_stage ($lo .. $hi)"))
    return SEAM_FIELDS_PER_STAGE[] = k
end

function hip_fill!(fields, bcs, locs, indices, grid; stage::Symbol, comm = nothing, arch = nothing)
    Nx, Ny, _ = size(grid)
    Hx, Hy, _ = halo_size(grid)
    zips = comm === nothing || comm.rank == comm.nranks - 1                     # the fold lives on the serial grid / the last rank
    for ((FT, Nz, Hz), group) in fill_groups(fields, locs, indices, grid)
      # one seam message holds at most TPG_MAX_FIELDS fields: a larger group goes in batches (as HaloFillPlan does in fields.py)
      for idx in Iterators.partition(group, TPG_MAX_FIELDS)
        fs   = [fields[n] for n in idx]
        ptrs = Ptr{Cvoid}[device_pointer(f) for f in fs]
        xloc = Int8[loc_code(locs[n][1]) for n in idx]
        yloc = Int8[loc_code(locs[n][2]) for n in idx]
        sgn  = Int32[zips ? zipper_sign(bcs[n].north) : Int32(1) for n in idx]
        bufs = comm === nothing ? ntuple(_ -> nothing, 4) : seam_buffers(arch, grid, length(fs), FT, Nz, Hz)
        bp   = map(b -> b === nothing ? C_NULL : device_pointer(b), bufs)
        s    = stream_for(arch === nothing ? architecture(grid) : arch)
        fps  = SEAM_FIELDS_PER_STAGE[]
        GC.@preserve fs bufs begin
            if stage === :zipper
                zips && check(ccall((:tpg_zipper_fill, libtripolar), Cint,
                            (Ptr{Ptr{Cvoid}}, Cint, Ptr{Int8}, Ptr{Int8}, Ptr{Int32}, Cint, Cint, Cint, Cint, Cint, Cint, Cint, Cint, Cint, Ptr{Cvoid}),
                            ptrs, length(fs), xloc, yloc, sgn, Nx, Ny, Nz, Hx, Hy, Hz, 1, Nz, ft_code(FT), s))
            elseif stage === :periodic
                check(ccall((:tpg_periodic_x_fill, libtripolar), Cint,
                            (Ptr{Ptr{Cvoid}}, Cint, Cint, Cint, Cint, Cint, Cint, Cint, Cint, Ptr{Cvoid}),
                            ptrs, length(fs), Nx, Ny, Nz, Hx, Hy, Hz, ft_code(FT), s))
                if comm !== nothing && fps > 0
                    check(ccall((:tpg_halo_exchange_y_pipelined, libtripolar), Cint,
                            (Ptr{Cvoid}, Cint, Cint, Ptr{Ptr{Cvoid}}, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid},
                             Cint, Cint, Cint, Cint, Cint, Cint, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Cint),
                            comm.handle, comm.rank, comm.nranks, ptrs, length(fs), bp[1], bp[2], bp[3], bp[4],
                            Nx, Ny, Nz, Hx, Hy, Hz, ft_code(FT), s, comm_stream(), fps))
                elseif comm !== nothing
                    check(ccall((:tpg_halo_exchange_y, libtripolar), Cint,
                            (Ptr{Cvoid}, Cint, Cint, Ptr{Ptr{Cvoid}}, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid},
                             Cint, Cint, Cint, Cint, Cint, Cint, Cint, Ptr{Cvoid}),
                            comm.handle, comm.rank, comm.nranks, ptrs, length(fs), bp[1], bp[2], bp[3], bp[4],
                            Nx, Ny, Nz, Hx, Hy, Hz, ft_code(FT), s))
                end
            elseif comm === nothing                                              # :all, serial grid: ONE launch (fused / merged)
                check(ccall((:tpg_fill_halo_regions, libtripolar), Cint,
                            (Ptr{Ptr{Cvoid}}, Cint, Ptr{Int8}, Ptr{Int8}, Ptr{Int32}, Cint, Cint, Cint, Cint, Cint, Cint, Cint, Cint, Ptr{Cvoid}),
                            ptrs, length(fs), xloc, yloc, sgn, Nx, Ny, Nz, Hx, Hy, Hz, 1, ft_code(FT), s))
            elseif fps > 0                                                       # :all, latitude band, pipelined seam exchange
                check(ccall((:tpg_fill_halo_regions_distributed_pipelined, libtripolar), Cint,
                            (Ptr{Cvoid}, Cint, Cint, Ptr{Ptr{Cvoid}}, Cint, Ptr{Int8}, Ptr{Int8}, Ptr{Int32},
                             Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cint, Cint, Cint, Cint, Cint, Cint, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Cint),
                            comm.handle, comm.rank, comm.nranks, ptrs, length(fs), xloc, yloc, sgn, bp[1], bp[2], bp[3], bp[4],
                            Nx, Ny, Nz, Hx, Hy, Hz, ft_code(FT), s, comm_stream(), fps))
            else                                                                 # :all, latitude band: zipper (last rank) -> periodic x -> RCCL seams
                check(ccall((:tpg_fill_halo_regions_distributed, libtripolar), Cint,
                            (Ptr{Cvoid}, Cint, Cint, Ptr{Ptr{Cvoid}}, Cint, Ptr{Int8}, Ptr{Int8}, Ptr{Int32},
                             Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cint, Cint, Cint, Cint, Cint, Cint, Cint, Ptr{Cvoid}),
                            comm.handle, comm.rank, comm.nranks, ptrs, length(fs), xloc, yloc, sgn, bp[1], bp[2], bp[3], bp[4],
                            Nx, Ny, Nz, Hx, Hy, Hz, ft_code(FT), s))
            end
        end
      end
    end
    return nothing
end

"`zipper_fill!(fields, bcs, locs, grid)`: the fold alone for fields given with their north conditions (kept for direct use)"
zipper_fill!(fields, north_bcs, locs, grid; periodic_x::Bool = false) =
    hip_fill!(as_tuple(fields), map(b -> (north = b,), as_tuple(north_bcs)), as_tuple(locs), (:, :, :), grid; stage = periodic_x ? :all : :zipper)

# Oceananigans' own launchers for the sides this library does not fill (south; bottom / top) [recalled names and argument order:
# fill_halo_event!(c, kernel!, bcs, indices, loc, arch, grid, args...) with the pair launchers fill_south_and_north_halo! /
# fill_bottom_and_top_halo!; a `nothing` condition makes the corresponding side a no-op there]
const OBC = Oceananigans.BoundaryConditions
# Oceananigans' own halo kernels are KernelAbstractions kernels: they need a backend, and HIPGPU() has none (see the head of this file).
# Every hand-over to them goes through this check, so that a field this library cannot fill on HIPGPU() ends in an ArgumentError naming
# the condition -- not in a MethodError from `device(::HIPGPU)` -- while on a backend architecture (ext/TripolarHIPBackendExt.jl) the
# hand-over proceeds.
has_ka_backend(arch) = true
has_ka_backend(::HIPGPU) = false
has_ka_backend(arch::Distributed) = has_ka_backend(serial_arch(arch))
needs_oceananigans_kernels(grid, what) = has_ka_backend(architecture(grid)) ||
    throw(ArgumentError("fill_halo_regions! on a TripolarGrid built on HIPGPU(): $what is Oceananigans' own halo kernel to fill, and HIPGPU() has " *
                        "no KernelAbstractions backend to launch it on (libtripolar_hip fills the Zipper north side, periodic x and the " *
                        "latitude-band seams); keep the grid and its fields in the host backend's arrays (ext/TripolarHIPBackendExt.jl) " *
                        "for fields that carry such conditions"))
why_not(bcs, indices, zipper_expected) =
    !full_xy(indices) ? "a field windowed in x or y" :
    !(is_periodic(bcs.west) && is_periodic(bcs.east)) ? "a non-periodic west / east condition" : "a north condition that is not a Zipper"
side(c, bs, f) = c isa Tuple ? map(f, bs) : f(bs[1])                          # tupled fill: tuples of conditions; single field: one
south_only!(c, bs, indices, loc, arch, grid, args...; kwargs...) =
    OBC.fill_halo_event!(c, OBC.fill_south_and_north_halo!, (side(c, bs, b -> fills_south(b.south) ? b.south : nothing), side(c, bs, _ -> nothing)), indices, loc, arch, grid, args...; kwargs...)
bottom_and_top!(c, bs, indices, loc, arch, grid, args...; kwargs...) =
    OBC.fill_halo_event!(c, OBC.fill_bottom_and_top_halo!, (side(c, bs, b -> b.bottom), side(c, bs, b -> b.top)), indices, loc, arch, grid, args...; kwargs...)

is_communication(bc) = bc isa BoundaryCondition{<:Oceananigans.BoundaryConditions.DistributedCommunication}   # [recalled classification name; alias DCBC]
fills_south(bc) = !isnothing(bc) && !is_communication(bc)

function tripolar_fill!(c, bcs, indices, loc, grid, comm, args...; kwargs...)
    arch = architecture(grid)
    cs, bs, ls = as_tuple(c), as_tuple(bcs), c isa Tuple ? loc : (loc,)
    # Oceananigans' own south fill, for a south condition that is a real one: not `nothing` (src/tripolar_grid.jl:148: the reference's own
    # fills) and not the halo-communication condition injected on ranks > 0 (src/distributed_tripolar_grid.jl:171) -- that side is a seam,
    # filled by the RCCL exchange inside the C call; driving Oceananigans' south/north launcher with it would take the MPI path
    any(b -> fills_south(b.south), bs) && needs_oceananigans_kernels(grid, "a south boundary condition") &&
        south_only!(c, bs, indices, loc, arch, grid, args...; kwargs...)
    if all(b -> isnothing(b.bottom) && isnothing(b.top), bs)
        hip_fill!(cs, bs, ls, indices, grid; stage = :all, comm, arch)                      # one C call: fused / merged launch (+ seams)
    else
        needs_oceananigans_kernels(grid, "a bottom / top boundary condition")
        hip_fill!(cs, bs, ls, indices, grid; stage = :zipper, comm, arch)
        bottom_and_top!(c, bs, indices, loc, arch, grid, args...; kwargs...)
        hip_fill!(cs, bs, ls, indices, grid; stage = :periodic, comm, arch)
    end
    return nothing
end

import Oceananigans.BoundaryConditions: fill_halo_regions!

# serial tripolar grid: single field (c::OffsetArray, bcs::FieldBoundaryConditions) and the tupled fill (tuples of both).
# Fields the C ABI cannot take (non-periodic x, (x, y)-windowed indices, a non-Zipper north side) go to Oceananigans' GENERIC method
# through `invoke` -- on a backend architecture; on HIPGPU() (no KernelAbstractions backend) they end in an ArgumentError that names the
# condition (needs_oceananigans_kernels).  `invoke` needs a signature the generic method is applicable to and this method is not: the generic one is declared on
# `c::Union{OffsetArray, NTuple{<:Any, OffsetArray}}` with an untyped grid [recalled], so (typeof(c), ..., AbstractGrid, ...) selects it --
# `AbstractGrid` is not a subtype of TRG, hence never this method again; an all-`Any` signature would match no method at all.
function fill_halo_regions!(c::Union{OffsetArray, NTuple{N, OffsetArray} where N}, bcs, indices, loc, grid::TRG, args...; kwargs...)
    bad = findfirst(b -> !hip_fill_applies(b, indices, true), as_tuple(bcs))
    isnothing(bad) || (needs_oceananigans_kernels(grid, why_not(as_tuple(bcs)[bad], indices, true)) &&
        return invoke(fill_halo_regions!, Tuple{typeof(c), Any, Any, Any, AbstractGrid, Vararg{Any}}, c, bcs, indices, loc, grid, args...; kwargs...))
    return tripolar_fill!(c, bcs, indices, loc, grid, nothing, args...; kwargs...)
end

# latitude bands: the seams leave Oceananigans' MPI path (inject_halo_communication_boundary_conditions / FieldBoundaryBuffers,
# src/distributed_tripolar_grid.jl:171,195) for ONE RCCL send/recv group per fill inside the C call; `buffers` (Oceananigans'
# MPI staging buffers) stays unused
function fill_halo_regions!(c::Union{OffsetArray, NTuple{N, OffsetArray} where N}, bcs, indices, loc, grid::DTRG, buffers, args...; kwargs...)
    arch = architecture(grid)
    last = arch.local_rank == ranks(arch.partition)[2] - 1
    bad = findfirst(b -> !hip_fill_applies(b, indices, last), as_tuple(bcs))
    isnothing(bad) || (needs_oceananigans_kernels(grid, why_not(as_tuple(bcs)[bad], indices, last)) &&
        return invoke(fill_halo_regions!, Tuple{typeof(c), Any, Any, Any, DistributedGrid, Any, Vararg{Any}}, c, bcs, indices, loc, grid, buffers, args...; kwargs...))
    return tripolar_fill!(c, bcs, indices, loc, grid, seam_comm(arch), args...; kwargs...)
end

"""
    halo_exchange_y!(comm, fields, locs, grid; buffers = nothing)

The y-seam exchange alone (what the DTRG method above issues after the periodic pass), for direct use: ONE RCCL send/recv
group on the current stream, no host wait.  `buffers = (send_south, send_north, recv_south, recv_north)` device arrays of
`y_halo_buffer_elems` elements select the packed form; `nothing` the pack-free form (one RCCL operation per (field, level):
2-D and few-level fields only).
"""
function halo_exchange_y!(comm::SeamComm, fields, locs, grid; buffers = nothing)
    Nx, Ny, _ = size(grid)
    Hx, Hy, _ = halo_size(grid)
    for ((FT, Nz, Hz), idx) in fill_groups(fields, locs, (:, :, :), grid)
        fs   = [fields[n] for n in idx]
        ptrs = Ptr{Cvoid}[device_pointer(f) for f in fs]
        bp   = isnothing(buffers) ? ntuple(_ -> C_NULL, 4) : map(b -> isnothing(b) ? C_NULL : device_pointer(b), buffers)
        GC.@preserve fs buffers begin
            check(ccall((:tpg_halo_exchange_y, libtripolar), Cint,
                        (Ptr{Cvoid}, Cint, Cint, Ptr{Ptr{Cvoid}}, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid},
                         Cint, Cint, Cint, Cint, Cint, Cint, Cint, Ptr{Cvoid}),
                        comm.handle, comm.rank, comm.nranks, ptrs, length(fs), bp[1], bp[2], bp[3], bp[4],
                        Nx, Ny, Nz, Hx, Hy, Hz, ft_code(FT), stream_for(architecture(grid))))
        end
    end
    return nothing
end

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
                    device_pointer(angle), Nx, Ny, Hx, Hy, ft_code(eltype(grid)), stream_for(architecture(grid))))
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
                    Nx, Ny, Nz, Hx, Hy, Hz, ft_code(eltype(grid)), stream_for(architecture(grid))))
    end
    return u_out, v_out
end
convert_to_latlong_frame!(u_out, v_out, u, v, grid) = convert_frame!(u_out, v_out, u, v, grid; to_native = false)
convert_to_native_frame!(u_out, v_out, u, v, grid)  = convert_frame!(u_out, v_out, u, v, grid; to_native = true)

end # module
