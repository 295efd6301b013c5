# TripolarHIPBackendExt.jl -- the OPPOSITE hook of HIPArray / HIPGPU(): let a grid and fields that live in the HOST MODEL'S OWN device
# arrays reach the `tpg_*` entry points, so that Oceananigans' KernelAbstractions kernels (tendencies, its south / bottom / top halo
# fills, everything `on_architecture` touches) and the hand-written HIP kernels of libtripolar_hip run on the SAME memory, ordered on the
# SAME stream.  Replaces, for such a host, `on_architecture(arch, map(FT, A))` x 20 of src/tripolar_grid.jl:303-328: the 20 arrays are
# allocated as the backend's arrays and filled in place by ONE tpg_build_grid call.
#
# A package extension (Project.toml:  [weakdeps] AMDGPU = "21141c5a-9bdb-4563-92ae-f87d6854732e"
#                                     [extensions] TripolarHIPBackendExt = "AMDGPU"):
# loaded only when the host has loaded AMDGPU.jl itself; TripolarHIP.jl never imports it (tests/test_julia_glue_static.py checks both
# directions).  It contains pointer and stream EXTRACTION only -- no kernel, no launch, no code generation: north_star's "no
# KernelAbstractions / AMDGPU.jl multi-backend codegen" is about how this path computes, and this path still computes in
# libtripolar_hip.so alone.
#
# NOT executed anywhere (no Julia toolchain in the build container; parity of this file is unpinned).  Names of AMDGPU.jl / Oceananigans
# internals are [recalled]: ROCArray, AMDGPU.stream() returning a HIPStream whose `.stream` field is the hipStream_t, Oceananigans'
# `GPU(AMDGPU.ROCBackend())` architecture of its AMDGPU extension (Oceananigans >= 0.96).
module TripolarHIPBackendExt

using TripolarHIP
using AMDGPU: AMDGPU, ROCArray, ROCBackend
using Oceananigans.Architectures: GPU

import TripolarHIP: device_pointer, device_array, stream_for, has_ka_backend

const ROCGPU = GPU{<:ROCBackend}                      # Oceananigans' architecture value for an AMD device [recalled]

# raw HBM address of a backend array: the C ABI dereferences it on the device; GC.@preserve at the call sites keeps `a` alive
device_pointer(a::ROCArray) = Ptr{Cvoid}(pointer(a))

# an UNINITIALISED backend array of that shape (tpg_build_grid overwrites every element, halos included)
device_array(::ROCGPU, FT, dims...) = ROCArray{FT}(undef, dims...)

# the hipStream_t the host model's own kernels are ordered on (AMDGPU.jl keeps one per task): libtripolar_hip enqueues there too, so a
# tendency kernel launched after fill_halo_regions! sees filled halos without any host synchronisation
stream_for(::ROCGPU) = Ptr{Cvoid}(AMDGPU.stream().stream)

# Oceananigans' own halo kernels CAN launch on this architecture: the `invoke` fall-backs of TripolarHIP.jl proceed instead of throwing
has_ka_backend(::ROCGPU) = true

end # module
