// tpg_probe.hip -- validation entry point: evaluates the deterministic Float64 elementary functions
// (tpg_math.hpp) and their straight-line batch forms (tpg_batch.hpp) on caller-supplied arguments.
// Used by tests/test_gpu_math.py to prove, argument by argument, that (a) the device functions
// return the bits of the CPU restatement's functions and (b) every batch form returns the bits of
// its scalar function on the domain it claims (and flags what lies outside).
#include "tpg_common.hpp"
#include "tpg_math.hpp"
#include "tpg_batch.hpp"

using namespace tpgm;

namespace {

enum { F_SIN, F_COS, F_SIND, F_COSD, F_TAND, F_ATAN, F_ASIN, F_ASINH, F_SINH, F_COSH, F_ACOS,          // scalar
       F_SQRT_NR = 20, F_DIV_NR = 21,                                                          // unscaled sqrt / division
       B_SIN_SMALL = 100, B_COS, B_ATAN, B_ATAN_TAB, B_ATAN_SMALL, B_ASIN_SMALL, B_SIND, B_COSD,      // batch
       B_SIND_LAT, B_COSD_LAT, B_COS_LAT };                                                            // latitude-domain batch forms

__global__ __launch_bounds__(256) void k_probe(int which, const double* __restrict__ x, double* __restrict__ y,
                                               int* __restrict__ rare, long long n)
{
    __shared__ __attribute__((aligned(16))) double atab[TPG_ATAN_TABLE_DOUBLES];
    tpgb::atan_table_init(atab, threadIdx.x);
    __syncthreads();
    long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    // batch forms take 4 consecutive arguments per thread
    if (which >= 100) {
        long long base = t * 4;
        if (base >= n) return;
        double a[4], o[4], o2[4];
        for (int e = 0; e < 4; ++e) a[e] = x[base + e < n ? base + e : n - 1];
        bool r = false;
        switch (which) {
        case B_SIN_SMALL:  r = tpgb::sin_small_b<4>(a, o); break;
        case B_COS:        r = tpgb::cos_b<4>(a, o); break;
        case B_ATAN:       tpgb::atan_b<4>(a, o); break;
        case B_ATAN_TAB:   tpgb::atan_tab_b<4>(a, o, atab); break;
        case B_ATAN_SMALL: r = tpgb::atan_small_b<4>(a, o); break;
        case B_ASIN_SMALL: r = tpgb::asin_small_b<4>(a, o); break;
        case B_SIND:       tpgb::sincosd_b<4>(a, o, o2); break;
        case B_COSD:       tpgb::sincosd_b<4>(a, o2, o); break;
        case B_SIND_LAT:   r = tpgb::sincosd_lat_b<4>(a, o, o2); break;
        case B_COSD_LAT:   r = tpgb::sincosd_lat_b<4>(a, o2, o); break;
        case B_COS_LAT:    r = tpgb::cos_lat_b<4>(a, o); break;
        default: return;
        }
        for (int e = 0; e < 4; ++e) if (base + e < n) { y[base + e] = o[e]; rare[base + e] = r ? 1 : 0; }
        return;
    }
    if (t >= n) return;
    double v = x[t], s, c;
    switch (which) {
    case F_SIN:   y[t] = sinD(v); break;
    case F_COS:   y[t] = cosD(v); break;
    case F_SIND:  y[t] = sind(v); break;
    case F_COSD:  y[t] = cosd(v); break;
    case F_TAND:  y[t] = tand(v); break;
    case F_ATAN:  y[t] = atanD(v); break;
    case F_ASIN:  y[t] = asinD(v); break;
    case F_ASINH: y[t] = asinhD(v); break;
    case F_SINH:  sinh_cosh(v, s, c); y[t] = s; break;
    case F_COSH:  sinh_cosh(v, s, c); y[t] = c; break;
    case F_ACOS:  y[t] = acosD(v); break;
    case F_SQRT_NR: y[t] = sqrt_nr(v); break;
    case F_DIV_NR:  y[t] = div_nr(x[t & ~1ll], x[(t | 1) < n ? (t | 1) : t]); break;   // pairs (a, b): both slots get a / b
    default: y[t] = 0.0;
    }
    rare[t] = 0;
}

}  // namespace

extern "C" int tpg_math_probe(int which, const void* x, void* y, void* rare, long long n, void* stream)
{
    if (!x || !y || !rare || n < 0) { tpg::set_error("tpg_math_probe: bad arguments"); return TPG_ERR_INVALID_ARGUMENT; }
    if (n == 0) return TPG_OK;
    long long threads = which >= 100 ? (n + 3) / 4 : n;
    hipLaunchKernelGGL(k_probe, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, tpg::as_stream(stream),
                       which, static_cast<const double*>(x), static_cast<double*>(y), static_cast<int*>(rare), n);
    return tpg::launch_status("k_probe");
}
