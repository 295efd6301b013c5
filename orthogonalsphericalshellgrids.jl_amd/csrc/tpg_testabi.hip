// tpg_testabi.hip -- the test / bench-only entry points of libtripolar_hip_test.so (include/tripolar_hip_test.h).
// NOT part of the product library: libtripolar_hip.so is linked without this file (and without tpg_probe.hip), reads no
// TPG_* knob and exports exactly the reference-facing symbols of include/tripolar_hip.h (tests/test_abi.py).  The test
// library is the same objects plus these hooks: synthetic field fill, the same-shape copy probe of the fold, the knob
// reload, and the elementary-function probe.
#include "tpg_zipper_kernels.hpp"
#include "../../include/tripolar_hip_test.h"

namespace {

// ---- synthetic fill ---------------------------------------------------------------------------
__device__ __forceinline__ uint64_t splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

template <typename T>
__global__ __launch_bounds__(256) void k_synthetic(T* c, uint64_t seed, double sentinel, Geom g)
{
    long long n = g.plane * (g.Nz + 2 * g.Hz);
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < n;
         idx += (long long)gridDim.x * blockDim.x) {
        long long k = idx / g.plane, rem = idx - k * g.plane;
        int j = (int)(rem / g.sx), i = (int)(rem - (long long)j * g.sx);
        bool interior = i >= g.Hx && i < g.Hx + g.Nx && j >= g.Hy && j < g.Hy + g.Ny && k >= g.Hz && k < g.Hz + g.Nz;
        double v = sentinel;
        if (interior) {
            uint64_t h = splitmix64(seed ^ splitmix64((uint64_t)idx));
            v = ((double)(h >> 11) + 0.5) * 0x1p-52 - 1.0;          // uniform in (-1, 1), never 0
        }
        c[idx] = (T)v;
    }
}

}  // namespace

extern "C" {

int tpg_zipper_copy_probe(void* const fields[], int nfields, const int8_t yloc[],
                          int Nx, int Ny, int Nz, int Hx, int Hy, int Hz, int ft, void* stream,
                          void* start_event, void* stop_event)
{
    int rc = tpg::check_geom(Nx, Ny, Nz, Hx, Hy, Hz, ft);
    if (rc) return rc;
    if ((rc = check_fields(fields, nfields))) return rc;
    if (!yloc) { tpg::set_error("null location table"); return TPG_ERR_INVALID_ARGUMENT; }
    if (nfields > TPG_MAX_FIELDS) { tpg::set_error("copy probe: at most %d fields (one kernel)", TPG_MAX_FIELDS); return TPG_ERR_UNSUPPORTED; }
    int8_t xl[TPG_MAX_FIELDS]; int32_t sg[TPG_MAX_FIELDS];
    for (int f = 0; f < nfields; ++f) { xl[f] = TPG_CENTER; sg[f] = 1; }
    Geom g = tpg::make_geom(Nx, Ny, Nz, Hx, Hy, Hz);
    tpg::ev_start = static_cast<hipEvent_t>(start_event);
    tpg::ev_stop = static_cast<hipEvent_t>(stop_event);
    rc = (ft == TPG_F64) ? zipper_batch<double, true>(fields, nfields, xl, yloc, sg, g, 1, Nz, tpg::as_stream(stream))
                         : zipper_batch<float, true>(fields, nfields, xl, yloc, sg, g, 1, Nz, tpg::as_stream(stream));
    tpg::ev_start = tpg::ev_stop = nullptr;
    return rc;
}

int tpg_fill_synthetic(void* field, uint64_t seed, double halo_sentinel,
                       int Nx, int Ny, int Nz, int Hx, int Hy, int Hz, int ft, void* stream)
{
    int rc = tpg::check_geom(Nx, Ny, Nz, Hx, Hy, Hz, ft);
    if (rc) return rc;
    if (!field) { tpg::set_error("null field"); return TPG_ERR_INVALID_ARGUMENT; }
    Geom g = tpg::make_geom(Nx, Ny, Nz, Hx, Hy, Hz);
    hipStream_t s = tpg::as_stream(stream);
    if (ft == TPG_F64) hipLaunchKernelGGL(k_synthetic<double>, dim3(256 * 16), dim3(256), 0, s, static_cast<double*>(field), seed, halo_sentinel, g);
    else               hipLaunchKernelGGL(k_synthetic<float>, dim3(256 * 16), dim3(256), 0, s, static_cast<float*>(field), seed, halo_sentinel, g);
    return tpg::launch_status("k_synthetic");
}

}  // extern "C"
