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
#include <type_traits>
#include "tpg_math.hpp"

using namespace tpgm;

namespace {

constexpr double kRad2Deg = 0x1.ca5dc1a63c1f8p+5;          // 180 / Float64(pi)  (Julia rad2deg)

struct V3 { double x, y, z; };

template <typename T>
__device__ __forceinline__ V3 node_vector(const T* __restrict__ lam, const T* __restrict__ phi, long long idx)
{
    double sl, cl, sp, cp;
    sincosd((double)lam[idx], sl, cl);
    sincosd((double)phi[idx], sp, cp);
    return V3{ cl * cp, sl * cp, sp };                       // lat_lon_to_cartesian(phi, lambda, 1)
}

// A block of 64 x NR threads evaluates one node per thread (the expensive part: two sincosd), parks the unit vectors in
// LDS and, after one barrier, every thread that owns a cell (lanes 0..62, rows 0..NR-2: tiles overlap by one column and one
// row) forms its two chords from its own node and the east / north neighbour's: 63 x (NR-1) cells from 64 x NR node
// evaluations instead of three evaluations per cell.
constexpr int NR = 8;
template <typename T>
__global__ __launch_bounds__(64 * NR) void k_nonorthogonality(const T* __restrict__ lam, const T* __restrict__ phi,
                                                              const uint8_t* __restrict__ immersed, double* __restrict__ angle,
                                                              int Nx, int Ny, int Hx, int Hy)
{
    __shared__ double P[3][NR][64];
    const int lane = threadIdx.x & 63, p = threadIdx.x >> 6;
    const int i = blockIdx.x * 63 + lane + 1;                    // 1-based node / cell column
    const int j = blockIdx.y * (NR - 1) + p + 1;
    const bool node = i <= Nx && j <= Ny;
    V3 p0 = { 0.0, 0.0, 0.0 };
    if (node) {
        const long long c = (long long)(i + Hx - 1) + (long long)(Nx + 2 * Hx) * (j + Hy - 1);
        p0 = node_vector(lam, phi, c);
    }
    P[0][p][lane] = p0.x; P[1][p][lane] = p0.y; P[2][p][lane] = p0.z;
    __syncthreads();
    if (!node || lane == 63 || p == NR - 1) return;              // apron column / row: the next tile owns these cells
    const long long o = (long long)(i - 1) + (long long)Nx * (j - 1);
    if (i > Nx - 1 || j > Ny - 1) { angle[o] = 0.0; return; }    // outside the (Nx-1, Ny-1) launch: zeros(size(grid)...) (:64)
    const V3 p1 = { P[0][p][lane + 1], P[1][p][lane + 1], P[2][p][lane + 1] };   // node (i+1, j)
    const V3 p2 = { P[0][p + 1][lane], P[1][p + 1][lane], P[2][p + 1][lane] };   // node (i, j+1)
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
constexpr int ZCH = 16;                                        // levels per thread of k_convert_frame

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
    // a thread owns ZCH consecutive levels of its column: the direction cosines (2 divisions + sqrt + 2 divisions) are
    // paid once per ZCH cells, and 4 levels of loads are in flight before the first store
    const int k0 = blockIdx.z * ZCH, k1 = k0 + ZCH < a.Nz ? k0 + ZCH : a.Nz;
    long long c3 = c2 + a.plane * (k0 + a.Hz);
    int k = k0;
    for (; k + 4 <= k1; k += 4, c3 += 4 * a.plane) {
        T p[4], q[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) { p[e] = __builtin_nontemporal_load(u + c3 + e * a.plane); q[e] = __builtin_nontemporal_load(v + c3 + e * a.plane); }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            T x, y;
            if (a.to_native) { x = p[e] * d1 + q[e] * d2; y = p[e] * d2 - q[e] * d1; }     // :54
            else             { x = p[e] * d1 - q[e] * d2; y = p[e] * d2 + q[e] * d1; }     // :31
            __builtin_nontemporal_store(x, uo + c3 + e * a.plane); __builtin_nontemporal_store(y, vo + c3 + e * a.plane);
        }
    }
    for (; k < k1; ++k, c3 += a.plane) {
        const T p = u[c3], q = v[c3];
        if (a.to_native) { uo[c3] = p * d1 + q * d2; vo[c3] = p * d2 - q * d1; }
        else             { uo[c3] = p * d1 - q * d2; vo[c3] = p * d2 + q * d1; }
    }
}

// 16-byte form: a thread owns W adjacent columns (2 doubles / 4 floats) of ZCH consecutive levels -- W sets of direction
// cosines, 16-B streaming loads and stores (Nx a multiple of W; otherwise the scalar kernel).  LOOSE = true: the same chunks accessed
// element-aligned, for an Hx that is not a multiple of W -- the reference's model halo (5, 5, 5), examples/bickley_jet.jl:21 -- or
// 16-B-misaligned arrays (the halo-fill kernels' GEN form does the same: csrc/tpg_zipper_kernels.hpp).
// Threads are numbered over (chunk, row) jointly: a row of 3600 columns is 1800 chunks = 7.03 blocks of 256, and a grid with one block
// row per grid row would leave every eighth block with 8 live lanes (round 5, tools/frame_ab.py: -3 .. -5 % with the flat numbering;
// more loads in flight, plain loads, 8 / 32 / all levels per thread: all within +-2 %).
template <typename T, int W, bool LOOSE>
__global__ __launch_bounds__(256) void k_convert_frame_vec(const T* __restrict__ phi_cf, const T* __restrict__ phi_fc,
                                                           const T* __restrict__ dy_cc, const T* __restrict__ dx_cc,
                                                           const T* __restrict__ u, const T* __restrict__ v,
                                                           T* __restrict__ uo, T* __restrict__ vo, FrameArgs a)
{
    typedef T aligned_t __attribute__((ext_vector_type(W)));
    typedef T loose_t __attribute__((ext_vector_type(W), aligned(sizeof(T))));
    typedef typename std::conditional<LOOSE, loose_t, aligned_t>::type vec_t;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int per_row = a.Nx / W;
    if (t >= (long long)per_row * a.Ny) return;
    const int j = (int)(t / per_row) + 1;
    const int i = (int)(t - (long long)(j - 1) * per_row) * W + 1;          // first of the W columns
    const long long c2 = (long long)(i + a.Hx - 1) + (long long)a.sx * (j + a.Hy - 1);
    const T d2r = (T)kDeg2Rad;
    T d1[W], d2[W];
#pragma unroll
    for (int e = 0; e < W; ++e) {
        const T ut = ((phi_cf[c2 + e + a.sx] - phi_cf[c2 + e]) * d2r) / dy_cc[c2 + e];
        const T vt = -((phi_fc[c2 + e + 1] - phi_fc[c2 + e]) * d2r) / dx_cc[c2 + e];
        const T U = root<T>(ut * ut + vt * vt);
        d1[e] = ut / U; d2[e] = vt / U;
    }
    const int k0 = blockIdx.y * ZCH, k1 = k0 + ZCH < a.Nz ? k0 + ZCH : a.Nz;
    long long c3 = c2 + a.plane * (k0 + a.Hz);
    int k = k0;
    for (; k + 4 <= k1; k += 4, c3 += 4 * a.plane) {
        vec_t p[4], q[4];
#pragma unroll
        for (int l = 0; l < 4; ++l) {
            p[l] = __builtin_nontemporal_load(reinterpret_cast<const vec_t*>(u + c3 + l * a.plane));
            q[l] = __builtin_nontemporal_load(reinterpret_cast<const vec_t*>(v + c3 + l * a.plane));
        }
#pragma unroll
        for (int l = 0; l < 4; ++l) {
            vec_t x, y;
#pragma unroll
            for (int e = 0; e < W; ++e) {
                if (a.to_native) { x[e] = p[l][e] * d1[e] + q[l][e] * d2[e]; y[e] = p[l][e] * d2[e] - q[l][e] * d1[e]; }
                else             { x[e] = p[l][e] * d1[e] - q[l][e] * d2[e]; y[e] = p[l][e] * d2[e] + q[l][e] * d1[e]; }
            }
            __builtin_nontemporal_store(x, reinterpret_cast<vec_t*>(uo + c3 + l * a.plane));
            __builtin_nontemporal_store(y, reinterpret_cast<vec_t*>(vo + c3 + l * a.plane));
        }
    }
    for (; k < k1; ++k, c3 += a.plane) {
        const vec_t p = *reinterpret_cast<const vec_t*>(u + c3), q = *reinterpret_cast<const vec_t*>(v + c3);
        vec_t x, y;
#pragma unroll
        for (int e = 0; e < W; ++e) {
            if (a.to_native) { x[e] = p[e] * d1[e] + q[e] * d2[e]; y[e] = p[e] * d2[e] - q[e] * d1[e]; }
            else             { x[e] = p[e] * d1[e] - q[e] * d2[e]; y[e] = p[e] * d2[e] + q[e] * d1[e]; }
        }
        *reinterpret_cast<vec_t*>(uo + c3) = x; *reinterpret_cast<vec_t*>(vo + c3) = y;
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
    if ((Ny + NR - 2) / (NR - 1) > 65535) { tpg::set_error("Ny too large"); return TPG_ERR_UNSUPPORTED; }
    dim3 grid((Nx + 62) / 63, (Ny + NR - 2) / (NR - 1));
    hipStream_t s = tpg::as_stream(stream);
    if (ft == TPG_F64) hipLaunchKernelGGL(k_nonorthogonality<double>, grid, dim3(64 * NR), 0, s, static_cast<const double*>(lambda_ff),
                                          static_cast<const double*>(phi_ff), immersed, angle, Nx, Ny, Hx, Hy);
    else               hipLaunchKernelGGL(k_nonorthogonality<float>, grid, dim3(64 * NR), 0, s, static_cast<const float*>(lambda_ff),
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
    hipStream_t s = tpg::as_stream(stream);
    const int W = ft == TPG_F64 ? 2 : 4;
    if (Nx % W == 0) {
        bool aligned = Hx % W == 0;
        for (const void* q : { u, v, (const void*)u_out, (const void*)v_out }) aligned = aligned && ((uintptr_t)q % 16) == 0;
        dim3 grid((unsigned)(((long long)(Nx / W) * Ny + 255) / 256), (Nz + ZCH - 1) / ZCH);      // (chunk, row) jointly; level groups on y
#define TPG_FRAME_LAUNCH(T, W_, LOOSE_)                                                                                                    \
        hipLaunchKernelGGL((k_convert_frame_vec<T, W_, LOOSE_>), grid, dim3(256), 0, s, static_cast<const T*>(phi_cf), static_cast<const T*>(phi_fc), \
                           static_cast<const T*>(dy_cc), static_cast<const T*>(dx_cc), static_cast<const T*>(u), static_cast<const T*>(v),   \
                           static_cast<T*>(u_out), static_cast<T*>(v_out), a)
        if (ft == TPG_F64) { if (aligned) TPG_FRAME_LAUNCH(double, 2, false); else TPG_FRAME_LAUNCH(double, 2, true); }
        else               { if (aligned) TPG_FRAME_LAUNCH(float, 4, false);  else TPG_FRAME_LAUNCH(float, 4, true); }
#undef TPG_FRAME_LAUNCH
        return tpg::launch_status("k_convert_frame_vec");
    }
    dim3 grid((Nx + 255) / 256, Ny, (Nz + ZCH - 1) / ZCH);
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
