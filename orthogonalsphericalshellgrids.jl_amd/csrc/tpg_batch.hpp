// tpg_batch.hpp -- straight-line ("batched") forms of the tpg_math.hpp functions.
//
// The metric kernel is FP64 latency-bound: at 2 waves/SIMD a chain of dependent v_fma_f64 leaves
// the SIMD idle ~1/3 of the time, and the early-outs / range branches of the scalar functions cut
// the code into small basic blocks the scheduler cannot interleave.  Each function here evaluates
// N independent arguments in one basic block, stage by stage ("vertical" order), with selects
// instead of branches.  Every output has EXACTLY the bits of the scalar function:
//   * the msun early-outs (|x| < 2^-27, |x| >= 2^66, ...) are shortcuts, not different values: the
//     general formula rounds to the same result there (argued next to each function);
//   * where a scalar function genuinely takes another path (|x| > pi/4 in sin, |x| >= 0.5 in asin,
//     deep cancellation in the pi/2 reduction) the batch form reports a `rare` flag and the caller
//     re-evaluates through the scalar function.
#pragma once
#include "tpg_math.hpp"

namespace tpgb {
using namespace tpgm;

#define TPG_UNROLL _Pragma("unroll")

// ksin / kcos of x + y, elementwise (any N)
template <int N> TPG_DEV void ksin_b(const double (&x)[N], const double (&y)[N], double (&out)[N])
{
    double z[N], r[N], v[N];
    TPG_UNROLL for (int e = 0; e < N; ++e) z[e] = x[e] * x[e];
    TPG_UNROLL for (int e = 0; e < N; ++e) r[e] = fmaD(z[e], 1.58969099521155010221e-10, -2.50507602534068634195e-08);
    TPG_UNROLL for (int e = 0; e < N; ++e) r[e] = fmaD(z[e], r[e], 2.75573137070700676789e-06);
    TPG_UNROLL for (int e = 0; e < N; ++e) r[e] = fmaD(z[e], r[e], -1.98412698298579493134e-04);
    TPG_UNROLL for (int e = 0; e < N; ++e) r[e] = fmaD(z[e], r[e], 8.33333333332248946124e-03);
    TPG_UNROLL for (int e = 0; e < N; ++e) v[e] = z[e] * x[e];
    TPG_UNROLL for (int e = 0; e < N; ++e)
        out[e] = x[e] - ((z[e] * (0.5 * y[e] - v[e] * r[e]) - y[e]) - v[e] * -1.66666666666666324348e-01);
}
template <int N> TPG_DEV void kcos_b(const double (&x)[N], const double (&y)[N], double (&out)[N])
{
    double z[N], r[N];
    TPG_UNROLL for (int e = 0; e < N; ++e) z[e] = x[e] * x[e];
    TPG_UNROLL for (int e = 0; e < N; ++e) r[e] = fmaD(z[e], -1.13596475577881948265e-11, 2.08757232129817482790e-09);
    TPG_UNROLL for (int e = 0; e < N; ++e) r[e] = fmaD(z[e], r[e], -2.75573143513906633035e-07);
    TPG_UNROLL for (int e = 0; e < N; ++e) r[e] = fmaD(z[e], r[e], 2.48015872894767294178e-05);
    TPG_UNROLL for (int e = 0; e < N; ++e) r[e] = fmaD(z[e], r[e], -1.38888888888741095749e-03);
    TPG_UNROLL for (int e = 0; e < N; ++e) r[e] = fmaD(z[e], r[e], 4.16666666666666019037e-02);
    TPG_UNROLL for (int e = 0; e < N; ++e) r[e] = z[e] * r[e];
    TPG_UNROLL for (int e = 0; e < N; ++e) {
        double hz = 0.5 * z[e];
        double w = 1.0 - hz;
        out[e] = w + (((1.0 - w) - hz) + (z[e] * r[e] - x[e] * y[e]));
    }
}

// sin(x) for |x| <= pi/4 (the haversine's half-differences).  tpgm::sinD returns x for
// |x| < 2^-26: ksin(x, 0) = x - x^3/6 (1 - ...) rounds to x there (|x^2/6| < 2^-54), also for +-0.
// rare: some |x| > pi/4 (seam-crossing longitude differences) -> caller uses tpgm::sinD.
// ksin(x, 0): the y = 0 form of ksin_b with the two operations on y removed.  ksin_b evaluates x - ((z (0.5 y - v r) - y) - v S1);
// with y = +0 the inner difference is 0 - v r and "- y" is the identity.  0 - t equals -t except for t = +0 (0 - 0 = +0, -(+0) = -0),
// and that sign never reaches the result: t = v r = +0 only when v = +0 (r > 0), z (-+0) is a zero of either sign, the next term
// v S1 is then -0 (S1 < 0) and (+-0) - (-0) = +0 in both cases.  So z * -(v r) gives the bits of z * (0 - v r) - 0 everywhere
// (checked argument by argument against the scalar sinD, signed zeros and subnormals included: tests/test_gpu_math.py) -- one VALU
// instruction less per sine, 16 per cell of the metric kernel.
template <int N> TPG_DEV void ksin0_b(const double (&x)[N], double (&out)[N])
{
    double z[N], r[N], v[N];
    TPG_UNROLL for (int e = 0; e < N; ++e) z[e] = x[e] * x[e];
    TPG_UNROLL for (int e = 0; e < N; ++e) r[e] = fmaD(z[e], 1.58969099521155010221e-10, -2.50507602534068634195e-08);
    TPG_UNROLL for (int e = 0; e < N; ++e) r[e] = fmaD(z[e], r[e], 2.75573137070700676789e-06);
    TPG_UNROLL for (int e = 0; e < N; ++e) r[e] = fmaD(z[e], r[e], -1.98412698298579493134e-04);
    TPG_UNROLL for (int e = 0; e < N; ++e) r[e] = fmaD(z[e], r[e], 8.33333333332248946124e-03);
    TPG_UNROLL for (int e = 0; e < N; ++e) v[e] = z[e] * x[e];
    TPG_UNROLL for (int e = 0; e < N; ++e) {
        const double t = v[e] * r[e];
        out[e] = x[e] - (z[e] * -t - v[e] * -1.66666666666666324348e-01);
    }
}

template <int N> TPG_DEV bool sin_small_b(const double (&x)[N], double (&out)[N])
{
    bool rare = false;
    TPG_UNROLL for (int e = 0; e < N; ++e) rare |= !(absD(x[e]) <= kPio4Hi);
    ksin0_b<N>(x, out);
    return rare;
}

// cos(a) through the first Cody-Waite step for every argument (n = 0 reproduces the direct
// kcos(a, 0) of tpgm::cosD exactly: fn = +-0, y0 = a, y1 = 0).
// rare: the reduction would need its 2nd iteration (a within 2^-16 relative of an odd multiple of
// pi/2, e.g. the geographic pole phi = 90) -> caller uses tpgm::cosD.
template <int N> TPG_DEV bool cos_b(const double (&a)[N], double (&out)[N])
{
    double fn[N], y0[N], y1[N], S[N], C[N];
    bool rare = false;
    TPG_UNROLL for (int e = 0; e < N; ++e) fn[e] = __builtin_rint(a[e] * kInvPio2);
    TPG_UNROLL for (int e = 0; e < N; ++e) {
        double r = a[e] - fn[e] * kPio2_1;
        double w = fn[e] * kPio2_1t;
        y0[e] = r - w;
        y1[e] = (r - y0[e]) - w;
        // the scalar code iterates when the exponents of a and y0 differ by more than 16, i.e. |y0| < 2^-16 |a| (both normal);
        // |y0| < 2^-15 |a| is a superset of that (it sends a few more lanes to the scalar function, whose result the fast path
        // equals wherever it is valid) and costs a multiply and a compare instead of two exponent extractions, a subtract and a compare
        rare |= absD(y0[e]) < absD(a[e]) * 0x1p-15;
    }
    // lanes of a wave sit on neighbouring latitudes: usually every argument needs the same kernel
    bool odd = false, even = false;
    TPG_UNROLL for (int e = 0; e < N; ++e) { const int n = (int)fn[e]; odd |= (n & 1) != 0; even |= (n & 1) == 0; }
    TPG_UNROLL for (int e = 0; e < N; ++e) { S[e] = 0.0; C[e] = 0.0; }
    if (__any(odd)) ksin_b<N>(y0, y1, S);
    if (__any(even)) kcos_b<N>(y0, y1, C);
    TPG_UNROLL for (int e = 0; e < N; ++e) {
        const int n = (int)fn[e];
        const double v = (n & 1) ? S[e] : C[e];
        const double nv = -v;
        out[e] = ((n + 1) & 2) ? nv : v;
    }
    return rare;
}

// cos(a) for a = deg2rad(latitude), |a| <= pi/2 (+ rounding): the Cody-Waite quotient n = rint(a 2/pi) is -1, 0 or +1, so the
// quadrant logic of cos_b -- convert to int, test bit 0, test bit 1 of n + 1 -- collapses to two compares of the quotient itself:
// n = 0 -> kcos(y), n = +1 -> -ksin(y), n = -1 -> +ksin(y).  Same reduction, same kernels, same bits as cos_b / tpgm::cosD.
// rare: |n| > 1 (not a latitude), or the reduction would need its 2nd iteration (see cos_b) -> caller uses tpgm::cosD.
template <int N> TPG_DEV bool cos_lat_b(const double (&a)[N], double (&out)[N])
{
    double fn[N], y0[N], y1[N], S[N], C[N];
    bool rare = false;
    TPG_UNROLL for (int e = 0; e < N; ++e) fn[e] = __builtin_rint(a[e] * kInvPio2);
    TPG_UNROLL for (int e = 0; e < N; ++e) {
        double r = a[e] - fn[e] * kPio2_1;
        double w = fn[e] * kPio2_1t;
        y0[e] = r - w;
        y1[e] = (r - y0[e]) - w;
        rare |= absD(y0[e]) < absD(a[e]) * 0x1p-15;
        rare |= !(absD(fn[e]) <= 1.0);
    }
    bool odd = false, even = false;
    TPG_UNROLL for (int e = 0; e < N; ++e) { odd |= fn[e] != 0.0; even |= fn[e] == 0.0; }
    TPG_UNROLL for (int e = 0; e < N; ++e) { S[e] = 0.0; C[e] = 0.0; }
    if (__any(odd)) ksin_b<N>(y0, y1, S);
    if (__any(even)) kcos_b<N>(y0, y1, C);
    TPG_UNROLL for (int e = 0; e < N; ++e) {
        const double v = (fn[e] != 0.0) ? S[e] : C[e];
        const double nv = -v;
        out[e] = (fn[e] > 0.0) ? nv : v;
    }
    return rare;
}

// atan(x), all x (finite, +-Inf; NaN -> NaN).  The scalar early-outs are redundant value-wise:
//  |x| < 2^-27: t - t (s1+s2) with s1+s2 ~ t^2/3 < 2^-55 rounds to t;
//  |x| >= 2^66 (and Inf): t = -1/|x| is below half an ulp of pi/2, the formula gives RN(hi + lo) = hi.
template <int N> TPG_DEV void atan_b(const double (&x)[N], double (&out)[N])
{
    double t[N], hi[N], lo[N], s[N];
    bool direct[N];
    TPG_UNROLL for (int e = 0; e < N; ++e) {
        double ax = absD(x[e]);
        direct[e] = ax < 0.4375;
        const bool b0 = ax < 0.6875, b1 = ax < 1.1875, b2 = ax < 2.4375;
        // every candidate is computed, then chosen with plain selects (arms are variables, so the
        // compiler emits v_cndmask instead of divergent branches)
        const double n0 = 2.0 * ax - 1.0, n1 = ax - 1.0, n2 = ax - 1.5;
        const double d0 = 2.0 + ax, d1 = ax + 1.0, d2 = 1.0 + 1.5 * ax;
        double num = -1.0, den = ax, h = kPio2Hi, l = kPio2Lo;
        const double h2 = 0x1.f730bd281f69bp-1, l2 = 0x1.007887af0cbbdp-56;
        const double h1 = 0x1.921fb54442d18p-1, l1 = 0x1.1a62633145c07p-55;
        const double h0 = 0x1.dac670561bb4fp-2, l0 = 0x1.a2b7f222f65e2p-56;
        num = b2 ? n2 : num; den = b2 ? d2 : den; h = b2 ? h2 : h; l = b2 ? l2 : l;
        num = b1 ? n1 : num; den = b1 ? d1 : den; h = b1 ? h1 : h; l = b1 ? l1 : l;
        num = b0 ? n0 : num; den = b0 ? d0 : den; h = b0 ? h0 : h; l = b0 ? l0 : l;
        hi[e] = h; lo[e] = l;
        double q = num / den;
        asm volatile("" : "+v"(q));          // keep the division unconditional: no branch around it
        t[e] = direct[e] ? ax : q;
    }
    {
        double z[N], w[N], s1[N], s2[N];
        TPG_UNROLL for (int e = 0; e < N; ++e) { z[e] = t[e] * t[e]; w[e] = z[e] * z[e]; }
        TPG_UNROLL for (int e = 0; e < N; ++e) { s1[e] = fmaD(w[e], 1.62858201153657823623e-02, 4.97687799461593236017e-02);
                                                 s2[e] = fmaD(w[e], -3.65315727442169155270e-02, -5.83357013379057348645e-02); }
        TPG_UNROLL for (int e = 0; e < N; ++e) { s1[e] = fmaD(w[e], s1[e], 6.66107313738753120669e-02);
                                                 s2[e] = fmaD(w[e], s2[e], -7.69187620504482999495e-02); }
        TPG_UNROLL for (int e = 0; e < N; ++e) { s1[e] = fmaD(w[e], s1[e], 9.09088713343650656196e-02);
                                                 s2[e] = fmaD(w[e], s2[e], -1.11111104054623557880e-01); }
        TPG_UNROLL for (int e = 0; e < N; ++e) { s1[e] = fmaD(w[e], s1[e], 1.42857142725034663711e-01);
                                                 s2[e] = fmaD(w[e], s2[e], -1.99999999998764832476e-01); }
        TPG_UNROLL for (int e = 0; e < N; ++e) { s1[e] = fmaD(w[e], s1[e], 3.33333333333329318027e-01);
                                                 s2[e] = w[e] * s2[e]; }
        TPG_UNROLL for (int e = 0; e < N; ++e) s[e] = z[e] * s1[e] + s2[e];
    }
    TPG_UNROLL for (int e = 0; e < N; ++e) {
        const double ts = t[e] * s[e];
        const double rd = t[e] - ts, rg = hi[e] - ((ts - lo[e]) - t[e]);
        const double r = direct[e] ? rd : rg;
        out[e] = csign(r, x[e]);
    }
}

// Range-reduction constants of atan, one row per msun interval (the "direct" interval |x| < 0.4375 first):
//   t = (p*ax - q) / (r + s*ax),  result = hi - ((t*poly - lo) - t)
// (p*ax, s*ax are exact for p,s in {0,1,2}; 1.5*ax rounds exactly as the scalar code's 1.5*ax).  The direct
// interval is the row p = 1, q = 0, r = 1, s = 0, hi = lo = 0: t = ax / 1 = ax exactly and
// 0 - ((t*poly - 0) - t) is the scalar branch's t - t*poly bit for bit (both differences are exact
// negations of each other; a zero difference is +0 either way).
// Looked up per lane from LDS (3 ds_read_b128) instead of selects over the unused candidates.
struct AtanRow { double p, q, r, s, hi, lo; };
// The interval is looked up too (round 3): the four thresholds 0.4375, 0.6875, 1.1875, 2.4375 have zero low words and high words
// that are multiples of 2^15, so |x| >= T  <=>  (high word of |x|) >> 15 >= (high word of T) >> 15, and the 81 values
// u = clamp(((hi >> 15) & 0xFFFF) - 0x7FB7, 0, 80) cover all five intervals (u = 0: below 0.4375 ... u = 80: from 2.4375 up, +Inf
// and NaN included -- a NaN argument gives NaN through any row).  lut[u] = byte offset of the row: 3 integer instructions and one
// ds_read_u8 instead of 4 compares + 4 selects.  The LUT sits behind the 30 row doubles (12 more doubles per copy).
#define TPG_ATAN_ROWS_DOUBLES 30
#define TPG_ATAN_TABLE_DOUBLES 42
// LUT packed four bytes to a word: byte b = row offset of u = b (0 | 1..20 -> 48 | 21..46 -> 96 | 47..79 -> 144 | 80.. -> 192)
__device__ const unsigned kAtanLutWords[24] = {
    0x30303000u, 0x30303030u, 0x30303030u, 0x30303030u, 0x30303030u, 0x60606030u, 0x60606060u, 0x60606060u,
    0x60606060u, 0x60606060u, 0x60606060u, 0x90606060u, 0x90909090u, 0x90909090u, 0x90909090u, 0x90909090u,
    0x90909090u, 0x90909090u, 0x90909090u, 0x90909090u, 0xC0C0C0C0u, 0xC0C0C0C0u, 0xC0C0C0C0u, 0xC0C0C0C0u };
TPG_DEV void atan_table_init(double* tab, int tid)
{
    if (tid < 24) reinterpret_cast<unsigned*>(tab + TPG_ATAN_ROWS_DOUBLES)[tid] = kAtanLutWords[tid];
    const double rows[5][6] = {
        { 1.0, 0.0, 1.0, 0.0, 0.0, 0.0 },                                        // [0, 0.4375): direct
        { 2.0, 1.0, 2.0, 1.0, 0x1.dac670561bb4fp-2, 0x1.a2b7f222f65e2p-56 },     // [0.4375, 0.6875): (2x-1)/(2+x), atan(0.5)
        { 1.0, 1.0, 1.0, 1.0, 0x1.921fb54442d18p-1, 0x1.1a62633145c07p-55 },     // [0.6875, 1.1875): (x-1)/(x+1),  atan(1)
        { 1.0, 1.5, 1.0, 1.5, 0x1.f730bd281f69bp-1, 0x1.007887af0cbbdp-56 },     // [1.1875, 2.4375): (x-1.5)/(1+1.5x), atan(1.5)
        { 0.0, 1.0, 0.0, 1.0, kPio2Hi, kPio2Lo } };                              // [2.4375, inf]:   -1/x, pi/2
    if (tid < TPG_ATAN_ROWS_DOUBLES) tab[tid] = rows[tid / 6][tid % 6];
}

// atan(x), all x, with the interval constants taken from the LDS table (see atan_b for the
// value-equivalence of dropping the msun early-outs).  p = 0 on the last interval: the numerator
// uses min(ax, 2^1000) so that an infinite argument (y/x at x = +-0) still gives 0*ax - 1 = -1.
// FINITE = true: the caller guarantees finite arguments (e.g. sqrt(x^2 + y^2)) and the clamp is skipped.
template <int N, bool FINITE = false> TPG_DEV void atan_tab_b(const double (&x)[N], double (&out)[N], const double* tab)
{
    double t[N], hi[N], lo[N], s[N];
    TPG_UNROLL for (int e = 0; e < N; ++e) {
        const double ax = absD(x[e]);
        int u = (int)__builtin_amdgcn_ubfe((unsigned)__double2hiint(x[e]), 15u, 16u) - 0x7FB7;     // bits 15..30 of the high word: sign dropped
        u = u < 0 ? 0 : (u > 80 ? 80 : u);
        const unsigned off = reinterpret_cast<const unsigned char*>(tab + TPG_ATAN_ROWS_DOUBLES)[u];   // bytes
        const AtanRow row = *reinterpret_cast<const AtanRow*>(reinterpret_cast<const char*>(tab) + off);
        hi[e] = row.hi; lo[e] = row.lo;
        const double axn = (!FINITE && ax > 0x1p1000) ? 0x1p1000 : ax;      // NaN stays NaN (and selects the direct row)
        const double num = row.p * axn - row.q;
        // den in [1, 1.5 * 2^1000] (axn, not ax: for ax >= 2^1000 the quotient -1/den is below 2^-999 either
        // way, its square underflows to 0 and hi - ((t*0 - lo) - t) rounds to hi - (-lo) regardless), so the
        // unscaled division is exact-equivalent; a NaN argument propagates through both forms
        const double den = row.r + row.s * axn;
        double q = tpgm::div_nr(num, den);
        asm volatile("" : "+v"(q));          // scheduling fence: keeps the live set of 4-wave kernels under 128 VGPRs
        t[e] = q;
    }
    {
        double z[N], w[N], s1[N], s2[N];
        TPG_UNROLL for (int e = 0; e < N; ++e) { z[e] = t[e] * t[e]; w[e] = z[e] * z[e]; }
        TPG_UNROLL for (int e = 0; e < N; ++e) { s1[e] = fmaD(w[e], 1.62858201153657823623e-02, 4.97687799461593236017e-02);
                                                 s2[e] = fmaD(w[e], -3.65315727442169155270e-02, -5.83357013379057348645e-02); }
        TPG_UNROLL for (int e = 0; e < N; ++e) { s1[e] = fmaD(w[e], s1[e], 6.66107313738753120669e-02);
                                                 s2[e] = fmaD(w[e], s2[e], -7.69187620504482999495e-02); }
        TPG_UNROLL for (int e = 0; e < N; ++e) { s1[e] = fmaD(w[e], s1[e], 9.09088713343650656196e-02);
                                                 s2[e] = fmaD(w[e], s2[e], -1.11111104054623557880e-01); }
        TPG_UNROLL for (int e = 0; e < N; ++e) { s1[e] = fmaD(w[e], s1[e], 1.42857142725034663711e-01);
                                                 s2[e] = fmaD(w[e], s2[e], -1.99999999998764832476e-01); }
        TPG_UNROLL for (int e = 0; e < N; ++e) { s1[e] = fmaD(w[e], s1[e], 3.33333333333329318027e-01);
                                                 s2[e] = w[e] * s2[e]; }
        TPG_UNROLL for (int e = 0; e < N; ++e) s[e] = z[e] * s1[e] + s2[e];
    }
    TPG_UNROLL for (int e = 0; e < N; ++e) {
        const double ts = t[e] * s[e];
        const double r = hi[e] - ((ts - lo[e]) - t[e]);
        out[e] = csign(r, x[e]);
    }
}

// atan(x) for |x| < 0.4375: the msun "direct" branch only (no argument reduction, no division).
// The eight spherical-triangle tangents of a cell are O(cell area) ~ 1e-6, so this is their path.
// rare: some |x| >= 0.4375 (or NaN) -> caller uses atan_b.
template <int N> TPG_DEV bool atan_small_b(const double (&x)[N], double (&out)[N])
{
    double t[N], z[N], w[N], s1[N], s2[N];
    bool rare = false;
    TPG_UNROLL for (int e = 0; e < N; ++e) { t[e] = absD(x[e]); rare |= !(t[e] < 0.4375); }
    TPG_UNROLL for (int e = 0; e < N; ++e) { z[e] = t[e] * t[e]; w[e] = z[e] * z[e]; }
    TPG_UNROLL for (int e = 0; e < N; ++e) { s1[e] = fmaD(w[e], 1.62858201153657823623e-02, 4.97687799461593236017e-02);
                                             s2[e] = fmaD(w[e], -3.65315727442169155270e-02, -5.83357013379057348645e-02); }
    TPG_UNROLL for (int e = 0; e < N; ++e) { s1[e] = fmaD(w[e], s1[e], 6.66107313738753120669e-02);
                                             s2[e] = fmaD(w[e], s2[e], -7.69187620504482999495e-02); }
    TPG_UNROLL for (int e = 0; e < N; ++e) { s1[e] = fmaD(w[e], s1[e], 9.09088713343650656196e-02);
                                             s2[e] = fmaD(w[e], s2[e], -1.11111104054623557880e-01); }
    TPG_UNROLL for (int e = 0; e < N; ++e) { s1[e] = fmaD(w[e], s1[e], 1.42857142725034663711e-01);
                                             s2[e] = fmaD(w[e], s2[e], -1.99999999998764832476e-01); }
    TPG_UNROLL for (int e = 0; e < N; ++e) { s1[e] = fmaD(w[e], s1[e], 3.33333333333329318027e-01);
                                             s2[e] = w[e] * s2[e]; }
    TPG_UNROLL for (int e = 0; e < N; ++e) {
        const double sm = z[e] * s1[e] + s2[e];
        const double r = t[e] - t[e] * sm;
        out[e] = csign(r, x[e]);
    }
    return rare;
}

// asin(x) for |x| < 0.5.  tpgm::asinD returns x for |x| < 2^-26: x + x (p/q) with p/q ~ x^2/6
// rounds to x there.  rare: some |x| >= 0.5 (edges longer than 60 degrees: the overwritten row
// j = 1, toy grids) -> caller uses tpgm::asinD.
template <int N> TPG_DEV bool asin_small_b(const double (&x)[N], double (&out)[N])
{
    double t[N], p[N], q[N];
    bool rare = false;
    TPG_UNROLL for (int e = 0; e < N; ++e) { t[e] = x[e] * x[e]; rare |= !(absD(x[e]) < 0.5); }
    TPG_UNROLL for (int e = 0; e < N; ++e) { p[e] = fmaD(t[e], 3.47933107596021167570e-05, 7.91534994289814532176e-04);
                                             q[e] = fmaD(t[e], 7.70381505559019352791e-02, -6.88283971605453293030e-01); }
    TPG_UNROLL for (int e = 0; e < N; ++e) { p[e] = fmaD(t[e], p[e], -4.00555345006794114027e-02);
                                             q[e] = fmaD(t[e], q[e], 2.02094576023350569471e+00); }
    TPG_UNROLL for (int e = 0; e < N; ++e) { p[e] = fmaD(t[e], p[e], 2.01212532134862925881e-01);
                                             q[e] = fmaD(t[e], q[e], -2.40339491173441421878e+00); }
    TPG_UNROLL for (int e = 0; e < N; ++e) { p[e] = fmaD(t[e], p[e], -3.25565818622400915405e-01);
                                             q[e] = fmaD(t[e], q[e], 1.0); }
    TPG_UNROLL for (int e = 0; e < N; ++e) p[e] = fmaD(t[e], p[e], 1.66666666666666657415e-01);
    TPG_UNROLL for (int e = 0; e < N; ++e) p[e] = t[e] * p[e];
    // q in (0.77, 1] for t < 0.25 and |p| <= 0.05 with p = 0 or |p| >= t/7 (t = x*x of a clamped sqrt: 0 or
    // >= 1e-40 here), so the unscaled division is exact-equivalent; lanes with |x| >= 0.5 or NaN are `rare`
    TPG_UNROLL for (int e = 0; e < N; ++e) out[e] = x[e] + x[e] * tpgm::div_nr(p[e], q[e]);
    return rare;
}

// sind / cosd pairs for |x| < 360 (longitudes in [0,360), latitudes in [-90,90]: rem(x,360) = x)
template <int N> TPG_DEV void sincosd_b(const double (&x)[N], double (&sn)[N], double (&cs)[N])
{
    // The octant logic is kept in booleans (lane masks in SGPRs, combined by scalar instructions) rather
    // than in integer counters: m = number of thresholds passed, m & 1 = parity of the four compares,
    // m >= 3 <=> third compare, m == 2 <=> second and not third, 90 m = a select among exact constants.
    double h[N], l[N], S[N], C[N], r[N], d[N];
    bool sodd[N], s_ge3[N], s_eq2[N], codd[N], c_eq1[N], c_eq2[N];
    TPG_UNROLL for (int e = 0; e < N; ++e) {
        r[e] = absD(x[e]);
        const bool s1 = r[e] >= 45.0, s2 = r[e] > 135.0, s3 = r[e] >= 225.0, s4 = r[e] > 315.0;    // sind's octants
        const bool c1 = r[e] > 45.0, c2 = r[e] >= 135.0, c3 = r[e] > 225.0, c4 = r[e] >= 315.0;    // cosd's octants
        sodd[e] = (s1 != s2) != (s3 != s4); s_ge3[e] = s3; s_eq2[e] = s2 && !s3;
        codd[e] = (c1 != c2) != (c3 != c4); c_eq1[e] = c1 && !c2; c_eq2[e] = c2 && !c3;
        double m90 = 0.0;
        m90 = s1 ? 90.0 : m90; m90 = s2 ? 180.0 : m90; m90 = s3 ? 270.0 : m90; m90 = s4 ? 360.0 : m90;
        d[e] = m90 - r[e];
        double t = absD(d[e]);
        double hh = t * kDeg2Rad;
        l[e] = fmaD(t, kDeg2Rad, -hh) + t * kDeg2RadLo;
        h[e] = hh;
    }
    ksin_b<N>(h, l, S);
    kcos_b<N>(h, l, C);
    TPG_UNROLL for (int e = 0; e < N; ++e) {
        const double sg = csign(1.0, x[e]), nsg = -sg;
        const double dsg = csign(1.0, d[e]);
        const double bs = sodd[e] ? C[e] : S[e];
        const double f2 = dsg * sg;
        double fs = sg;
        fs = s_ge3[e] ? nsg : fs;
        fs = s_eq2[e] ? f2 : fs;
        sn[e] = fs * bs;
        const double bc = codd[e] ? S[e] : C[e];
        const double a1 = 90.0 - r[e], a3 = r[e] - 270.0;
        const double arg = c_eq1[e] ? a1 : a3;
        const double fodd = csign(1.0, arg);
        const double feven = c_eq2[e] ? -1.0 : 1.0;
        const double fc = codd[e] ? fodd : feven;
        cs[e] = fc * bc;
    }
}

// sind / cosd pairs for latitudes, |x| <= 90: of sincosd_b's eight octant compares only the first of each set can be true, 90 m is
// 0 or 90, the sine carries the sign of x and the cosine is never negated (90 - r >= +0):
//    r < 45 (sind) / r <= 45 (cosd):  sind = sign(x) ksin(r),        cosd = kcos(r)
//    otherwise:                        sind = sign(x) kcos(90 - r),   cosd = ksin(90 - r)
// -- the same reduction t = |90 m - r|, the same kernels and therefore the same bits as sincosd_b / tpgm::sind, cosd.
// rare: some |x| > 90 or NaN -> caller uses sincosd_b.
template <int N> TPG_DEV bool sincosd_lat_b(const double (&x)[N], double (&sn)[N], double (&cs)[N])
{
    double h[N], l[N], S[N], C[N];
    bool s1[N], c1[N], rare = false;
    TPG_UNROLL for (int e = 0; e < N; ++e) {
        const double r = absD(x[e]);
        rare |= !(r <= 90.0);
        s1[e] = r >= 45.0; c1[e] = r > 45.0;
        const double d = (s1[e] ? 90.0 : 0.0) - r;
        const double t = absD(d);
        const double hh = t * kDeg2Rad;
        l[e] = fmaD(t, kDeg2Rad, -hh) + t * kDeg2RadLo;
        h[e] = hh;
    }
    ksin_b<N>(h, l, S);
    kcos_b<N>(h, l, C);
    TPG_UNROLL for (int e = 0; e < N; ++e) {
        const double bs = s1[e] ? C[e] : S[e];
        sn[e] = csign(1.0, x[e]) * bs;
        cs[e] = c1[e] ? S[e] : C[e];
    }
    return rare;
}

// fmod(x, 360) for 0 <= x < 720 (the second application in ((l % 360) + 360) % 360: l % 360 is in (-360, 360))
TPG_DEV double fmod360_pos(double x)
{
    // x - (x >= 360 ? 360 : 0): both constants have a zero low word, so the select is ONE v_cndmask on the high word, and x - 0.0 = x
    // for every x (signed zeros included)
    return x - __hiloint2double(!(x < 360.0) ? 0x40768000 : 0, 0);
}

// exact fmod(x, 360) for |x| < 720 (select form of tpgm::fmod360)
TPG_DEV double fmod360_small(double x)
{
    const double ax = absD(x);
    return csign(ax - __hiloint2double(!(ax < 360.0) ? 0x40768000 : 0, 0), x);       // csign(|x| - 0, x) = x
}

}  // namespace tpgb
