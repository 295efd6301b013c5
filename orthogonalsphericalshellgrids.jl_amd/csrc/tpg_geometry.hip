// tpg_geometry.hip -- grid-quality / frame utilities over the arrays tpg_build_grid produces.
//
//  * tpg_nonorthogonality_angle: compute_nonorthogonality_angle! of the reference's orthogonality test
//    (test/test_tripolar_grid.jl:8-34, launched over (Nx-1, Ny-1) at :70): the angle between the two grid lines
//    through every Face-Face node, minus 90 degrees -- an independent device-side check that the family of
//    ellipses and hyperbolae the grid is built from is orthogonal.
//  * tpg_convert_frame: convert_to_latlong_frame / convert_to_native_frame
//    (examples/convert_to_latlong_frame.jl:12-55): rotation of a (u, v) pair between the grid's local frame and
//    the geographic frame, from phi differences and the cell metrics.
// Elementwise, HBM-bound (24 B and 2 x 3 x s B per cell); same deterministic Float64 functions as the metric
// kernels (tpg_math.hpp), so results are bit-identical to the CPU restatement's (tests/).
#include "tpg_common.hpp"
#include "tpg_math.hpp"

using namespace tpgm;

namespace {

constexpr double kRad2Deg = 0x1.ca5dc1a63c1f8p+5;          // 180 / Float64(pi)  (Julia rad2deg)

struct V3 { double x, y, z; };

template <typename T>
__device__ __forceinline__ V3 node(const T* __restrict__ lam, const T* __restrict__ phi, long long idx)
{
    double sl, cl, sp, cp;
    sincosd((double)lam[idx], sl, cl);
    sincosd((double)phi[idx], sp, cp);
    return V3{ cl * cp, sl * cp, sp };                       // lat_lon_to_cartesian(phi, lambda, 1)
}

template <typename T>
__global__ __launch_bounds__(256) void k_nonorthogonality(const T* __restrict__ lam, const T* __restrict__ phi,
                                                          const uint8_t* __restrict__ immersed, double* __restrict__ angle,
                                                          int Nx, int Ny, int Hx, int Hy)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x + 1;  // 1-based
    const int j = blockIdx.y + 1;
    if (i > Nx) return;
    const long long o = (long long)(i - 1) + (long long)Nx * (j - 1);
    if (i > Nx - 1 || j > Ny - 1) { angle[o] = 0.0; return; } // outside the (Nx-1, Ny-1) launch: zeros(size(grid)...) (:64)
    const long long sx = Nx + 2 * Hx;
    const long long c = (long long)(i + Hx - 1) + sx * (j + Hy - 1);
    const V3 p0 = node(lam, phi, c), p1 = node(lam, phi, c + 1), p2 = node(lam, phi, c + sx);
    const double ax = p1.x - p0.x, ay = p1.y - p0.y, az = p1.z - p0.z;          // v1 (:23)
    const double bx = p2.x - p0.x, by = p2.y - p0.y, bz = p2.z - p0.z;          // v2 (:24)
    const double n1 = sqrt(ax * ax + ay * ay + az * az);
    const double n2 = sqrt(bx * bx + by * by + bz * bz);
    const double cs = (ax * bx + ay * by + az * bz) / (n1 * n2);                // :27
    const bool imm = immersed && immersed[o] != 0;
    const double a = (imm ? kPio2Hi : acosD(cs)) - kPio2Hi;                     // :29
    angle[o] = a * kRad2Deg;                                                    // :32
}

template <typename T> __device__ __forceinline__ T root(T x);
template <> __device__ __forceinline__ double root<double>(double x) { return sqrt(x); }
template <> __device__ __forceinline__ float root<float>(float x) { return sqrtf(x); }

struct FrameArgs { int Nx, Ny, Nz, Hx, Hy, Hz, sx; long long plane; int to_native; };

template <typename T>
__global__ __launch_bounds__(256) void k_convert_frame(const T* __restrict__ phi_cf, const T* __restrict__ phi_fc,
                                                       const T* __restrict__ dy_cc, const T* __restrict__ dx_cc,
                                                       const T* __restrict__ u, const T* __restrict__ v,
                                                       T* __restrict__ uo, T* __restrict__ vo, FrameArgs a)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x + 1;
    const int j = blockIdx.y + 1;
    if (i > a.Nx) return;
    const long long c2 = (long long)(i + a.Hx - 1) + (long long)a.sx * (j + a.Hy - 1);
    const T d2r = (T)kDeg2Rad;
    const T ut = ((phi_cf[c2 + a.sx] - phi_cf[c2]) * d2r) / dy_cc[c2];          // :14-18
    const T vt = -((phi_fc[c2 + 1] - phi_fc[c2]) * d2r) / dx_cc[c2];            // :20-24
    const T U = root<T>(ut * ut + vt * vt);                                     // :26
    const T d1 = ut / U, d2 = vt / U;                                           // :28-29
    for (int k = blockIdx.z; k < a.Nz; k += gridDim.z) {
        const long long c3 = c2 + a.plane * (k + a.Hz);
        const T p = u[c3], q = v[c3];
        if (a.to_native) { uo[c3] = p * d1 + q * d2; vo[c3] = p * d2 - q * d1; }   // :54
        else             { uo[c3] = p * d1 - q * d2; vo[c3] = p * d2 + q * d1; }   // :31
    }
}

}  // namespace

extern "C" {

int tpg_nonorthogonality_angle(const void* lambda_ff, const void* phi_ff, const uint8_t* immersed, double* angle,
                               int Nx, int Ny, int Hx, int Hy, int ft, void* stream)
{
    int rc = tpg::check_geom(Nx, Ny, 1, Hx, Hy, 0, ft);
    if (rc) return rc;
    if (!lambda_ff || !phi_ff || !angle) { tpg::set_error("null array"); return TPG_ERR_INVALID_ARGUMENT; }
    if (Ny > 65535) { tpg::set_error("Ny > 65535"); return TPG_ERR_UNSUPPORTED; }
    dim3 grid((Nx + 255) / 256, Ny);
    hipStream_t s = tpg::as_stream(stream);
    if (ft == TPG_F64) hipLaunchKernelGGL(k_nonorthogonality<double>, grid, dim3(256), 0, s, static_cast<const double*>(lambda_ff),
                                          static_cast<const double*>(phi_ff), immersed, angle, Nx, Ny, Hx, Hy);
    else               hipLaunchKernelGGL(k_nonorthogonality<float>, grid, dim3(256), 0, s, static_cast<const float*>(lambda_ff),
                                          static_cast<const float*>(phi_ff), immersed, angle, Nx, Ny, Hx, Hy);
    return tpg::launch_status("k_nonorthogonality");
}

int tpg_convert_frame(const void* phi_cf, const void* phi_fc, const void* dy_cc, const void* dx_cc,
                      const void* u, const void* v, void* u_out, void* v_out, int to_native,
                      int Nx, int Ny, int Nz, int Hx, int Hy, int Hz, int ft, void* stream)
{
    int rc = tpg::check_geom(Nx, Ny, Nz, Hx, Hy, Hz, ft);
    if (rc) return rc;
    if (!phi_cf || !phi_fc || !dy_cc || !dx_cc || !u || !v || !u_out || !v_out) { tpg::set_error("null array"); return TPG_ERR_INVALID_ARGUMENT; }
    if (Hx < 1 || Hy < 1) { tpg::set_error("the rotation reads phi at i+1 and j+1: halo (%d,%d) too small", Hx, Hy); return TPG_ERR_UNSUPPORTED; }
    if (Ny > 65535) { tpg::set_error("Ny > 65535"); return TPG_ERR_UNSUPPORTED; }
    tpg::Geom g = tpg::make_geom(Nx, Ny, Nz, Hx, Hy, Hz);
    FrameArgs a{ Nx, Ny, Nz, Hx, Hy, Hz, g.sx, g.plane, to_native ? 1 : 0 };
    dim3 grid((Nx + 255) / 256, Ny, Nz < 64 ? Nz : 64);
    hipStream_t s = tpg::as_stream(stream);
    if (ft == TPG_F64)
        hipLaunchKernelGGL(k_convert_frame<double>, grid, dim3(256), 0, s, static_cast<const double*>(phi_cf), static_cast<const double*>(phi_fc),
                           static_cast<const double*>(dy_cc), static_cast<const double*>(dx_cc), static_cast<const double*>(u),
                           static_cast<const double*>(v), static_cast<double*>(u_out), static_cast<double*>(v_out), a);
    else
        hipLaunchKernelGGL(k_convert_frame<float>, grid, dim3(256), 0, s, static_cast<const float*>(phi_cf), static_cast<const float*>(phi_fc),
                           static_cast<const float*>(dy_cc), static_cast<const float*>(dx_cc), static_cast<const float*>(u),
                           static_cast<const float*>(v), static_cast<float*>(u_out), static_cast<float*>(v_out), a);
    return tpg::launch_status("k_convert_frame");
}

}  // extern "C"
