// fillbench.hip -- standalone GPU-box microbenchmark for the two HBM-bound halo kernels at config 3
// (3600 x 1800 x 75, halo 4, Float64, 4 fields c/u/v/zeta).  NOT part of the product library: it holds
// experimental variants, same-shape pure-copy ceilings and an in-kernel timeline probe.
//
//   fillbench <experiment> [rounds]       (run under `rocprofv3 --kernel-trace` for device durations)
//
// Each experiment launches its kernel in three cache states per round, in this order:
//   cold-dirty (after a 1 GiB in-place write), cold-clean (after a 1 GiB read-only pass), warm (relaunch).
// Durations printed here are hipExtLaunchKernelGGL start/stop events (they include the launch's
// end-of-kernel write-back); rocprofv3's kernel trace of the same run gives the dispatch durations.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <string>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)

constexpr int NX = 3600, NY = 1800, NZ = 75, H = 4;
constexpr int SX = NX + 2 * H, SY = NY + 2 * H, LEV = NZ + 2 * H;
constexpr long long PLANE = (long long)SX * SY;
#ifndef FB_NF
#define FB_NF 4                                      // fields per launch (run.sh timeline builds 4, 8 and 16)
#endif
constexpr int NF = FB_NF;
constexpr int NCH = NX / 2;                       // 16-B chunks per row

typedef double d2 __attribute__((ext_vector_type(2)));
typedef double d2l __attribute__((ext_vector_type(2), aligned(8)));
typedef unsigned int u4 __attribute__((ext_vector_type(4)));

struct Fields { double* p[NF]; int xl[NF], yl[NF], sg[NF]; };

// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_flush_dirty(double* b, long long n)
{
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) b[i] += 1.0;
}
__global__ __launch_bounds__(256) void k_flush_clean(const double* b, long long n, double* out)
{
    double s = 0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) s += b[i];
    if (s == 1.2345e301) out[0] = s;
}

__global__ __launch_bounds__(256) void k_init(double* b, long long n, double seed)
{
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) b[i] = seed + (double)(i % 1000003) * 1e-3;
}

// ------------------------------------------------------------------------------------------------
// Column kernel (the product's k_zipper_cols restated) with knobs:
//   COPY  : no reversal, no sign: dst column = src column (same rows, same bytes): the copy ceiling
//   LD    : 0 plain, 1 nontemporal loads
//   ST    : 0 plain, 1 nontemporal, 2 write-through (sc1) buffer stores, 3 sc0 sc1
//   STAMP : per-wave s_memrealtime stamps {start, loads landed, stores issued, stores acked}
template <bool COPY, int LD, int ST, bool STAMP>
__global__ __launch_bounds__(256) void k_cols(Fields ft, int kcount, unsigned long long* stamps)
{
    const int f = blockIdx.y;
    const int item = blockIdx.x * 256 + threadIdx.x;
    unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0;
    if (STAMP) t0 = __builtin_amdgcn_s_memrealtime();
    if (item < kcount * NCH) {
        const int xl = ft.xl[f], yl = ft.yl[f];
        const double s = (double)ft.sg[f], as = s < 0 ? -s : s;
        double* c = ft.p[f];
        const int kk = item / NCH, ch = item - kk * NCH;
        const int i = ch * 2 + 1;
        double* lvl = c + PLANE * (kk + H) + H;
        const int prow = NY + H - 1;
        const int ysh = yl;
        const bool fix = (yl == 0) && (ch >= NCH / 2);
        const bool wrap = !COPY && xl == 1 && ch == 0;
        const int soff = COPY ? (i - 1) : (NX - i - 1 + xl);
        d2 v[H]; double w0[H];
#pragma unroll
        for (int jr = 1; jr <= H; ++jr) {
            const double* row = lvl + (long long)SX * (prow - jr + ysh);
            const d2l* p = reinterpret_cast<const d2l*>(row + soff);
            if (LD == 1) v[jr - 1] = __builtin_nontemporal_load(p); else v[jr - 1] = *p;
            w0[jr - 1] = wrap ? row[0] : 0.0;
        }
        d2 vf = {}, old = {};
        if (fix) {
            vf = *reinterpret_cast<const d2l*>(lvl + (long long)SX * prow + soff);
            if (i <= NX / 2) old = *reinterpret_cast<const d2*>(lvl + (long long)SX * prow + (i - 1));
        }
        if (STAMP) {
            // consume the loaded registers so that the stamp sits behind the loads' s_waitcnt
            asm volatile("" :: "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(vf));
            t1 = __builtin_amdgcn_s_memrealtime();
        }
        __amdgpu_buffer_rsrc_t rsrc;
        const int kk0 = __builtin_amdgcn_readfirstlane(kk);      // a wave may straddle two levels: descriptor on the first lane's
        if (ST >= 2) rsrc = __builtin_amdgcn_make_buffer_rsrc(c + PLANE * (kk0 + H) + H + (long long)SX * prow, 0, (int)(2 * PLANE * 8), 0x00020000);
#pragma unroll
        for (int jr = 1; jr <= H; ++jr) {
            d2 o;
            if (COPY) o = v[jr - 1];
            else { o[0] = s * v[jr - 1][1]; o[1] = s * v[jr - 1][0]; if (wrap) o[0] = as * w0[jr - 1]; }
            d2* q = reinterpret_cast<d2*>(lvl + (long long)SX * (prow + jr) + (i - 1));
            if (ST == 1) __builtin_nontemporal_store(o, q);
            else if (ST >= 2) {
                u4 bits = __builtin_bit_cast(u4, o);
                __builtin_amdgcn_raw_buffer_store_b128(bits, rsrc, (int)(((kk - kk0) * PLANE + jr * SX + (i - 1)) * 8), 0, ST == 2 ? 16 : 17);
            } else *q = o;
        }
        if (fix) {
            d2 o;
            if (COPY) { o[0] = (i > NX / 2) ? vf[0] : old[0]; o[1] = (i + 1 > NX / 2) ? vf[1] : old[1]; }
            else { o[0] = (i > NX / 2) ? s * vf[1] : old[0]; o[1] = (i + 1 > NX / 2) ? s * vf[0] : old[1]; }
            *reinterpret_cast<d2*>(lvl + (long long)SX * prow + (i - 1)) = o;
        }
        if (STAMP) {
            t2 = __builtin_amdgcn_s_memrealtime();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            t3 = __builtin_amdgcn_s_memrealtime();
        }
    }
    if (STAMP && (threadIdx.x & 63) == 0) {
        const long long w = ((long long)blockIdx.y * gridDim.x + blockIdx.x) * 4 + (threadIdx.x >> 6);
        stamps[w * 4 + 0] = t0; stamps[w * 4 + 1] = t1; stamps[w * 4 + 2] = t2; stamps[w * 4 + 3] = t3;
    }
}

// Persistent, software-pipelined form: a fixed grid of waves strides over the (field, level, chunk) items;
// the loads of item n+1 are in flight before the stores of item n are issued.
struct Item { d2 v[H]; d2 vf, old; double w0[H]; };

template <bool COPY, int LD>
__device__ __forceinline__ void item_load(const Fields& ft, long long it, Item& r)
{
    const int per_field = NZ * NCH;
    const int f = (int)(it / per_field);
    const int item = (int)(it - (long long)f * per_field);
    const int xl = ft.xl[f], yl = ft.yl[f];
    const int kk = item / NCH, ch = item - kk * NCH;
    const int i = ch * 2 + 1;
    const double* lvl = ft.p[f] + PLANE * (kk + H) + H;
    const int prow = NY + H - 1;
    const bool fix = (yl == 0) && (ch >= NCH / 2);
    const bool wrap = !COPY && xl == 1 && ch == 0;
    const int soff = COPY ? (i - 1) : (NX - i - 1 + xl);
#pragma unroll
    for (int jr = 1; jr <= H; ++jr) {
        const double* row = lvl + (long long)SX * (prow - jr + yl);
        const d2l* p = reinterpret_cast<const d2l*>(row + soff);
        if (LD == 1) r.v[jr - 1] = __builtin_nontemporal_load(p); else r.v[jr - 1] = *p;
        r.w0[jr - 1] = wrap ? row[0] : 0.0;
    }
    r.vf = d2{0, 0}; r.old = d2{0, 0};
    if (fix) {
        r.vf = *reinterpret_cast<const d2l*>(lvl + (long long)SX * prow + soff);
        if (i <= NX / 2) r.old = *reinterpret_cast<const d2*>(lvl + (long long)SX * prow + (i - 1));
    }
}

template <bool COPY, int ST = 0>
__device__ __forceinline__ void item_store(const Fields& ft, long long it, const Item& r)
{
    const int per_field = NZ * NCH;
    const int f = (int)(it / per_field);
    const int item = (int)(it - (long long)f * per_field);
    const int xl = ft.xl[f], yl = ft.yl[f];
    const double s = (double)ft.sg[f], as = s < 0 ? -s : s;
    const int kk = item / NCH, ch = item - kk * NCH;
    const int i = ch * 2 + 1;
    double* lvl = ft.p[f] + PLANE * (kk + H) + H;
    const int prow = NY + H - 1;
    const bool fix = (yl == 0) && (ch >= NCH / 2);
    const bool wrap = !COPY && xl == 1 && ch == 0;
#pragma unroll
    for (int jr = 1; jr <= H; ++jr) {
        d2 o;
        if (COPY) o = r.v[jr - 1];
        else { o[0] = s * r.v[jr - 1][1]; o[1] = s * r.v[jr - 1][0]; if (wrap) o[0] = as * r.w0[jr - 1]; }
        if (ST == 2) {
            // write-through (sc1) 16-B store; descriptor on the first lane's level (a wave may straddle two levels)
            const int f0 = __builtin_amdgcn_readfirstlane(f), kk0 = __builtin_amdgcn_readfirstlane(kk);
            __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(ft.p[f0] + PLANE * (kk0 + H) + H + (long long)SX * prow, 0, (int)(2 * PLANE * 8), 0x00020000);
            if (f == f0) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, o), rsrc, (int)(((kk - kk0) * PLANE + jr * SX + (i - 1)) * 8), 0, 16);
            else *reinterpret_cast<d2*>(lvl + (long long)SX * (prow + jr) + (i - 1)) = o;
        } else
        *reinterpret_cast<d2*>(lvl + (long long)SX * (prow + jr) + (i - 1)) = o;
    }
    if (fix) {
        d2 o;
        if (COPY) { o[0] = (i > NX / 2) ? r.vf[0] : r.old[0]; o[1] = (i + 1 > NX / 2) ? r.vf[1] : r.old[1]; }
        else { o[0] = (i > NX / 2) ? s * r.vf[1] : r.old[0]; o[1] = (i + 1 > NX / 2) ? s * r.vf[0] : r.old[1]; }
        *reinterpret_cast<d2*>(lvl + (long long)SX * prow + (i - 1)) = o;
    }
}

template <bool COPY, int LD, int ST, int BPC>
__global__ __launch_bounds__(256) void k_persist(Fields ft)
{
    const long long total = (long long)NF * NZ * NCH;
    const long long stride = (long long)gridDim.x * 256;
    long long it = (long long)blockIdx.x * 256 + threadIdx.x;
    if (it >= total) return;
    Item a, b;
    item_load<COPY, LD>(ft, it, a);
    for (;;) {
        const long long nx = it + stride;
        if (nx < total) item_load<COPY, LD>(ft, nx, b);
        item_store<COPY, ST>(ft, it, a);
        if (nx >= total) break;
        const long long nx2 = nx + stride;
        if (nx2 < total) item_load<COPY, LD>(ft, nx2, a);
        item_store<COPY, ST>(ft, nx, b);
        if (nx2 >= total) break;
        it = nx2;
    }
}

// Flat copy ceiling: the same bytes as one fold (per (field, level): Hy source rows -> Hy destination rows,
// + for y-Center fields half a row read and written in place), as plain linear 16-B copies, R per thread.
template <int R>
__global__ __launch_bounds__(256) void k_flatcopy(Fields ft)
{
    const int f = blockIdx.y;
    const int yl = ft.yl[f];
    double* c = ft.p[f];
    // linear chunk index over NZ levels x (H rows x SX/2 chunks)
    const int per_level = H * (SX / 2);
    const long long base = ((long long)blockIdx.x * 256 + threadIdx.x) * R;
    d2 v[R];
    long long dsto[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const long long idx = base + r;
        dsto[r] = -1;
        if (idx < (long long)NZ * per_level) {
            const int kk = (int)(idx / per_level), w = (int)(idx - (long long)kk * per_level);
            const long long lvl = PLANE * (kk + H);
            const long long src = lvl + (long long)SX * (NY + H - 1 - H + yl) + 2 * w;       // rows Ny-Hy+yl ..
            dsto[r] = lvl + (long long)SX * (NY + H) + 2 * w;
            v[r] = *reinterpret_cast<const d2*>(c + src);
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) if (dsto[r] >= 0) *reinterpret_cast<d2*>(c + dsto[r]) = v[r];
    // y-Center: the row-Ny substitution's bytes: read the west half, rewrite the east half
    if (yl == 0) {
        const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
        if (idx < (long long)NZ * (NCH / 2)) {
            const int kk = (int)(idx / (NCH / 2)), w = (int)(idx - (long long)kk * (NCH / 2));
            double* row = c + PLANE * (kk + H) + (long long)SX * (NY + H - 1) + H;
            *reinterpret_cast<d2*>(row + NX / 2 + 2 * w) = *reinterpret_cast<const d2*>(row + 2 * w);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Periodic-x variants.  rows = SY * LEV per field.
// P0: the product's form: thread = (row, 16-B chunk v of Hx/2): both sides
__global__ __launch_bounds__(256) void k_per0(Fields ft, long long nrows)
{
    long long item = (long long)blockIdx.x * 256 + threadIdx.x;
    if (item >= nrows * 2) return;
    long long row = item >> 1; int v = (int)(item & 1);
    u4* c = reinterpret_cast<u4*>(ft.p[blockIdx.y] + row * SX);
    u4 w = c[NX / 2 + v], e = c[H / 2 + v];
    c[v] = w; c[H / 2 + NX / 2 + v] = e;
}
// P1: K rows per thread (strided by the grid), all loads first
template <int K>
__global__ __launch_bounds__(256) void k_per1(Fields ft, long long nrows)
{
    long long item = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long stride = (long long)gridDim.x * 256;
    u4 w[K], e[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        long long it = item + k * stride;
        if (it < nrows * 2) {
            u4* c = reinterpret_cast<u4*>(ft.p[blockIdx.y] + (it >> 1) * SX);
            int v = (int)(it & 1);
            w[k] = c[NX / 2 + v]; e[k] = c[H / 2 + v];
        }
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
        long long it = item + k * stride;
        if (it < nrows * 2) {
            u4* c = reinterpret_cast<u4*>(ft.p[blockIdx.y] + (it >> 1) * SX);
            int v = (int)(it & 1);
            c[v] = w[k]; c[H / 2 + NX / 2 + v] = e[k];
        }
    }
}
// P2: one work item per row BOUNDARY b (between parent rows b and b+1 of the whole (level, row) stack of a
// field): the 64 contiguous bytes [halo-E(b) | halo-W(b+1)] are written by 4 lanes (16 B each);
// halo-E(b) <- int-W(b), halo-W(b+1) <- int-E(b+1).  The very first halo-W and very last halo-E are done by
// boundary -1 / nrows-1 (half items).
template <int K>
__global__ __launch_bounds__(256) void k_per2(Fields ft, long long nrows)
{
    const long long nb = nrows + 1;                       // boundaries -1 .. nrows-1, shifted by one
    long long item = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long stride = (long long)gridDim.x * 256;
    u4 val[K];
    u4* dst[K];
    u4* base = reinterpret_cast<u4*>(ft.p[blockIdx.y]);
    constexpr int RC = SX / 2;                            // 16-B chunks per row
#pragma unroll
    for (int k = 0; k < K; ++k) {
        long long it = item + k * stride;
        dst[k] = nullptr;
        if (it < nb * 4) {
            long long b = (it >> 2) - 1; int q = (int)(it & 3);
            if (q < 2) {                                  // halo-E of row b <- int-W of row b
                if (b >= 0) { u4* c = base + b * RC; val[k] = c[H / 2 + q]; dst[k] = c + H / 2 + NX / 2 + q; }
            } else {                                      // halo-W of row b+1 <- int-E of row b+1
                if (b + 1 < nrows) { u4* c = base + (b + 1) * RC; val[k] = c[NX / 2 + (q - 2)]; dst[k] = c + (q - 2); }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < K; ++k) if (dst[k]) *dst[k] = val[k];
}
// P3: P0 with nontemporal stores (the halo lines are not re-read by this pass)
__global__ __launch_bounds__(256) void k_per3(Fields ft, long long nrows)
{
    long long item = (long long)blockIdx.x * 256 + threadIdx.x;
    if (item >= nrows * 2) return;
    long long row = item >> 1; int v = (int)(item & 1);
    u4* c = reinterpret_cast<u4*>(ft.p[blockIdx.y] + row * SX);
    u4 w = c[NX / 2 + v], e = c[H / 2 + v];
    __builtin_nontemporal_store(w, c + v); __builtin_nontemporal_store(e, c + H / 2 + NX / 2 + v);
}
// P4: row-boundary items in 8-byte lanes: 8 lanes cover the 64 written bytes, reads are 8 B per lane too
//     (one wave-instruction = 8 boundaries ... cheap address math; tests whether narrower requests help)
__global__ __launch_bounds__(256) void k_per4(Fields ft, long long nrows)
{
    long long it = (long long)blockIdx.x * 256 + threadIdx.x;
    if (it >= (nrows + 1) * 8) return;
    long long b = (it >> 3) - 1; int q = (int)(it & 7);
    double* base = ft.p[blockIdx.y];
    if (q < 4) { if (b >= 0) { double* c = base + b * SX; c[H + NX + q] = c[H + q]; } }
    else if (b + 1 < nrows) { double* c = base + (b + 1) * SX; c[q - 4] = c[NX + (q - 4)]; }
}

// P5: cache-policy variants of the periodic pass (round 3).  The pass is bound by line transfers, not bytes: per row pair 3
// whole 128-B lines are fetched (for 128 B of sources) and 3 lines are dirtied.  Do scoped / non-temporal accesses change
// what crosses the L2 <-> memory boundary (sector-sized fetches, write-through instead of write-back)?
//   shape: the 8-byte-lane row-boundary form of P4 (scoped atomics exist for 8-byte accesses), or the 16-byte product form (LD/ST <= 1)
//   LD: 0 plain, 1 nontemporal, 2 agent-scope relaxed atomic load (sc1), 3 system-scope (sc0 sc1)
//   ST: 0 plain, 1 nontemporal, 2 agent-scope relaxed atomic store,      3 system-scope
template <int LD> __device__ __forceinline__ double ld8(const double* p)
{
    if (LD == 1) return __builtin_nontemporal_load(p);
    if (LD == 2) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (LD == 3) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    return *p;
}
template <int ST> __device__ __forceinline__ void st8(double* p, double v)
{
    if (ST == 1) __builtin_nontemporal_store(v, p);
    else if (ST == 2) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else if (ST == 3) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    else *p = v;
}
template <int LD, int ST>
__global__ __launch_bounds__(256) void k_per5(Fields ft, long long nrows)
{
    long long it = (long long)blockIdx.x * 256 + threadIdx.x;
    if (it >= (nrows + 1) * 8) return;
    long long b = (it >> 3) - 1; int q = (int)(it & 7);
    double* base = ft.p[blockIdx.y];
    if (q < 4) { if (b >= 0) { double* c = base + b * SX; st8<ST>(c + H + NX + q, ld8<LD>(c + H + q)); } }
    else if (b + 1 < nrows) { double* c = base + (b + 1) * SX; st8<ST>(c + (q - 4), ld8<LD>(c + NX + (q - 4))); }
}
// P6: the 16-byte product form with non-temporal loads (and optionally non-temporal stores)
template <int ST>
__global__ __launch_bounds__(256) void k_per6(Fields ft, long long nrows)
{
    long long item = (long long)blockIdx.x * 256 + threadIdx.x;
    if (item >= nrows * 2) return;
    long long row = item >> 1; int v = (int)(item & 1);
    u4* c = reinterpret_cast<u4*>(ft.p[blockIdx.y] + row * SX);
    u4 w = __builtin_nontemporal_load(c + NX / 2 + v), e = __builtin_nontemporal_load(c + H / 2 + v);
    if (ST) { __builtin_nontemporal_store(w, c + v); __builtin_nontemporal_store(e, c + H / 2 + NX / 2 + v); }
    else { c[v] = w; c[H / 2 + NX / 2 + v] = e; }
}

// Two half-row-apart chunks per thread: 4224 waves = one resident round on 8192 wave slots (the one-chunk
// form launches 8448), 8-10 independent 16-B loads per thread.
template <bool COPY, int LD>
__global__ __launch_bounds__(256) void k_cols2(Fields ft, int kcount)
{
    const int f = blockIdx.y;
    const int item = blockIdx.x * 256 + threadIdx.x;
    constexpr int HC = NCH / 2;
    if (item >= kcount * HC) return;
    const int xl = ft.xl[f], yl = ft.yl[f];
    const double s = (double)ft.sg[f], as = s < 0 ? -s : s;
    double* c = ft.p[f];
    const int kk = item / HC, c0 = item - kk * HC;
    double* lvl = c + PLANE * (kk + H) + H;
    const int prow = NY + H - 1;
    d2 v[2][H]; double w0[H];
    const bool wrap = !COPY && xl == 1 && c0 == 0;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int i = (c0 + h * HC) * 2 + 1;
        const int soff = COPY ? (i - 1) : (NX - i - 1 + xl);
#pragma unroll
        for (int jr = 1; jr <= H; ++jr) {
            const double* row = lvl + (long long)SX * (prow - jr + yl);
            const d2l* p = reinterpret_cast<const d2l*>(row + soff);
            if (LD == 1) v[h][jr - 1] = __builtin_nontemporal_load(p); else v[h][jr - 1] = *p;
            if (h == 0) w0[jr - 1] = wrap ? row[0] : 0.0;
        }
    }
    d2 vf = {};
    const int i1 = (c0 + HC) * 2 + 1;                       // east-half chunk: always i > Nx/2
    if (yl == 0) vf = *reinterpret_cast<const d2l*>(lvl + (long long)SX * prow + (COPY ? (i1 - 1) : (NX - i1 - 1 + xl)));
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int i = (c0 + h * HC) * 2 + 1;
#pragma unroll
        for (int jr = 1; jr <= H; ++jr) {
            d2 o;
            if (COPY) o = v[h][jr - 1];
            else { o[0] = s * v[h][jr - 1][1]; o[1] = s * v[h][jr - 1][0]; if (h == 0 && wrap) o[0] = as * w0[jr - 1]; }
            *reinterpret_cast<d2*>(lvl + (long long)SX * (prow + jr) + (i - 1)) = o;
        }
    }
    if (yl == 0) {
        d2 o;
        if (COPY) o = vf; else { o[0] = s * vf[1]; o[1] = s * vf[0]; }
        *reinterpret_cast<d2*>(lvl + (long long)SX * prow + (i1 - 1)) = o;
    }
}

// Line ceiling of the periodic pass: per pair of rows the pass must fetch 3 distinct 128-B lines and dirty the
// same 3 (row pitch 28 864 B = 225.5 lines).  This kernel reads exactly those lines whole and writes them back
// whole (8 lanes x 16 B per line): if it takes as long as the periodic pass, the pass is bound by the number of
// lines it must touch, not by the bytes it needs from them.
__global__ __launch_bounds__(256) void k_lines(Fields ft, long long nrows, int rw /* 1 read only, 2 write only, 3 both */)
{
    long long it = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long npairs = nrows / 2;
    if (it >= npairs * 24) return;
    const long long pr = it / 24; const int r = (int)(it - pr * 24); const int l = r >> 3, q = r & 7;
    char* base = reinterpret_cast<char*>(ft.p[blockIdx.y]);
    const long long pitch = (long long)SX * 8;
    long long off = l == 0 ? 2 * pr * pitch : (l == 1 ? (2 * pr + 1) * pitch - 64 : (2 * pr + 2) * pitch - 128);
    u4* p = reinterpret_cast<u4*>(base + off) + q;
    u4 v = { 1u, 2u, 3u, (unsigned)it };
    if (rw & 1) v = *p;
    if (rw & 2) { v[0] ^= 1u; *p = v; }
    else if (v[0] == 0x12345678u && v[1] == 0x9abcdef0u) *p = v;       // keep the load alive
}

// ------------------------------------------------------------------------------------------------
static Fields g_ft;
static double* g_flush; static const long long FLUSH_N = 1ll << 27;
static hipEvent_t ev0, ev1;

struct Result { std::vector<float> t[3]; };

template <typename Launch>
static Result run3(Launch launch, int rounds)
{
    Result r;
    for (int it = 0; it < rounds + 2; ++it) {
        for (int mode = 0; mode < 3; ++mode) {
            if (mode == 0) hipLaunchKernelGGL(k_flush_dirty, dim3(8192), dim3(256), 0, 0, g_flush, FLUSH_N);
            if (mode == 1) hipLaunchKernelGGL(k_flush_clean, dim3(8192), dim3(256), 0, 0, g_flush, FLUSH_N, g_flush);
            launch(ev0, ev1);
            CHECK(hipEventSynchronize(ev1));
            float ms; CHECK(hipEventElapsedTime(&ms, ev0, ev1));
            if (it >= 2) r.t[mode].push_back(ms * 1e3f);
        }
    }
    return r;
}

static void report(const char* name, Result& r, double bytes)
{
    const char* modes[3] = { "cold-dirty", "cold-clean", "warm" };
    for (int m = 0; m < 3; ++m) {
        auto& v = r.t[m]; std::sort(v.begin(), v.end());
        float med = v[v.size() / 2];
        printf("%-28s %-10s median %7.2f us  min %7.2f  -> %6.0f GB/s = %4.1f %% of 8 TB/s (events)\n", name, modes[m], med, v[0], bytes / med / 1e3, bytes / med / 1e3 / 80);
    }
    fflush(stdout);
}

template <bool COPY, int LD, int ST>
static void exp_cols(const char* name, int rounds, double bytes)
{
    dim3 grid((NZ * NCH + 255) / 256, NF);
    auto r = run3([&](hipEvent_t a, hipEvent_t b) { hipExtLaunchKernelGGL((k_cols<COPY, LD, ST, false>), grid, dim3(256), 0, 0, a, b, 0, g_ft, NZ, (unsigned long long*)nullptr); }, rounds);
    report(name, r, bytes);
}

static void timeline(const char* name, bool copy, int mode /*0 dirty 1 clean 2 warm*/)
{
    dim3 grid((NZ * NCH + 255) / 256, NF);
    const long long nw = (long long)grid.x * grid.y * 4;
    unsigned long long* d; CHECK(hipMalloc(&d, nw * 4 * 8));
    std::vector<unsigned long long> h(nw * 4);
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipMemset(d, 0, nw * 4 * 8));
        if (mode == 0) hipLaunchKernelGGL(k_flush_dirty, dim3(8192), dim3(256), 0, 0, g_flush, FLUSH_N);
        if (mode == 1) hipLaunchKernelGGL(k_flush_clean, dim3(8192), dim3(256), 0, 0, g_flush, FLUSH_N, g_flush);
        if (mode == 2) hipLaunchKernelGGL((k_cols<false, 1, 0, false>), grid, dim3(256), 0, 0, g_ft, NZ, (unsigned long long*)nullptr);
        if (copy) hipExtLaunchKernelGGL((k_cols<true, 1, 0, true>), grid, dim3(256), 0, 0, ev0, ev1, 0, g_ft, NZ, d);
        else      hipExtLaunchKernelGGL((k_cols<false, 1, 0, true>), grid, dim3(256), 0, 0, ev0, ev1, 0, g_ft, NZ, d);
        CHECK(hipEventSynchronize(ev1));
        float ms; CHECK(hipEventElapsedTime(&ms, ev0, ev1));
        CHECK(hipMemcpy(h.data(), d, nw * 4 * 8, hipMemcpyDeviceToHost));
        unsigned long long tmin = ~0ull;
        for (long long w = 0; w < nw; ++w) if (h[w * 4]) tmin = std::min(tmin, h[w * 4]);
        std::vector<double> s[4];
        for (long long w = 0; w < nw; ++w) if (h[w * 4 + 1]) for (int q = 0; q < 4; ++q) s[q].push_back((double)(h[w * 4 + q] - tmin) * 0.01);   // 100 MHz -> us
        const char* nm[4] = { "wave start", "loads landed", "stores issued", "stores acked" };
        printf("%s mode %d rep %d: kernel (events) %.2f us, %zu active waves\n", name, mode, rep, ms * 1e3, s[0].size());
        for (int q = 0; q < 4; ++q) {
            std::sort(s[q].begin(), s[q].end());
            auto P = [&](double p) { return s[q][(size_t)(p * (s[q].size() - 1))]; };
            printf("   %-14s p0 %6.2f  p10 %6.2f  p50 %6.2f  p90 %6.2f  p99 %6.2f  p100 %6.2f us\n", nm[q], P(0), P(.1), P(.5), P(.9), P(.99), P(1));
        }
        // load latency per wave
        std::vector<double> lat;
        for (long long w = 0; w < nw; ++w) if (h[w * 4 + 1]) lat.push_back((double)(h[w * 4 + 1] - h[w * 4]) * 0.01);
        std::sort(lat.begin(), lat.end());
        printf("   load latency   p0 %6.2f  p10 %6.2f  p50 %6.2f  p90 %6.2f  p100 %6.2f us\n", lat[0], lat[lat.size() / 10], lat[lat.size() / 2], lat[lat.size() * 9 / 10], lat.back());
        // the launch as ramp + saturated window + drain: between the 5 % and the 95 % quantile of "stores acked" 90 % of the bytes complete
        {
            auto Q = [&](int q, double p) { return s[q][(size_t)(p * (s[q].size() - 1))]; };
            const double bytes = (2 * 17.28e6 + 2 * 19.44e6) * NF / 4;
            const double t5 = Q(3, .05), t95 = Q(3, .95), win = t95 - t5;
            printf("   model          first 5 %% of the waves done at %.2f us (ramp), 5..95 %% in %.2f us = %.2f TB/s inside the window, last wave %.2f us after the 95 %% mark, "
                   "kernel - last ack %.2f us\n", t5, win, 0.9 * bytes / win / 1e6, Q(3, 1.0) - t95, ms * 1e3 - Q(3, 1.0));
        }
    }
    CHECK(hipFree(d));
    fflush(stdout);
}

int main(int argc, char** argv)
{
    const std::string exp = argc > 1 ? argv[1] : "all";
    const int rounds = argc > 2 ? atoi(argv[2]) : 10;
    const size_t fbytes = (size_t)PLANE * LEV * 8;
    int xl[NF], yl[NF], sg[NF];                     // locations cycle c (CC,+1), u (FC,-1), v (CF,-1), zeta (FF,+1)
    for (int f = 0; f < NF; ++f) { xl[f] = f & 1; yl[f] = (f >> 1) & 1; sg[f] = ((f & 3) == 0 || (f & 3) == 3) ? 1 : -1; }
    for (int f = 0; f < NF; ++f) {
        CHECK(hipMalloc(&g_ft.p[f], fbytes));
        hipLaunchKernelGGL(k_init, dim3(8192), dim3(256), 0, 0, g_ft.p[f], (long long)(fbytes / 8), 1.0 + f);
        g_ft.xl[f] = xl[f]; g_ft.yl[f] = yl[f]; g_ft.sg[f] = sg[f];
    }
    CHECK(hipMalloc(&g_flush, FLUSH_N * 8));
    CHECK(hipMemset(g_flush, 0, FLUSH_N * 8));
    // timing-only events, as the product's tpg_event_create: a default event makes the launch end with a system-scope release (+ ~2 us)
    CHECK(hipEventCreateWithFlags(&ev0, hipEventDisableSystemFence)); CHECK(hipEventCreateWithFlags(&ev1, hipEventDisableSystemFence));
    const double zbytes = (2 * 17.28e6 + 2 * 19.44e6) * NF / 4;
    const long long nrows = (long long)SY * LEV;
    const double pbytes = (double)nrows * NF * 2 * H * 2 * 8;

    auto want = [&](const char* n) { return exp == "all" || exp == n; };
    if (want("zip")) {
        exp_cols<false, 1, 0>("fold ntload plain-store", rounds, zbytes);
        exp_cols<false, 0, 0>("fold plain plain", rounds, zbytes);
        exp_cols<false, 1, 1>("fold ntload nt-store", rounds, zbytes);
        exp_cols<false, 1, 2>("fold ntload sc1-store", rounds, zbytes);
        exp_cols<false, 1, 3>("fold ntload sc0sc1-store", rounds, zbytes);
    }
    if (want("copy")) {
        exp_cols<true, 1, 0>("copy ntload plain-store", rounds, zbytes);
        exp_cols<true, 0, 0>("copy plain plain", rounds, zbytes);
        exp_cols<true, 1, 2>("copy ntload sc1-store", rounds, zbytes);
        { dim3 grid(((long long)NZ * H * (SX / 2) + 255) / 256, NF);
          auto r = run3([&](hipEvent_t a, hipEvent_t b) { hipExtLaunchKernelGGL(k_flatcopy<1>, grid, dim3(256), 0, 0, a, b, 0, g_ft); }, rounds);
          report("flatcopy R=1", r, zbytes); }
        { dim3 grid(((long long)NZ * H * (SX / 2) / 4 + 255) / 256, NF);
          auto r = run3([&](hipEvent_t a, hipEvent_t b) { hipExtLaunchKernelGGL(k_flatcopy<4>, grid, dim3(256), 0, 0, a, b, 0, g_ft); }, rounds);
          report("flatcopy R=4 (adjacent)", r, zbytes); }
    }

#define PERSIST(COPY, ST, BPC, NAME) { auto r = run3([&](hipEvent_t a, hipEvent_t b) { hipExtLaunchKernelGGL((k_persist<COPY, 1, ST, BPC>), dim3(256 * BPC), dim3(256), 0, 0, a, b, 0, g_ft); }, rounds); report(NAME, r, zbytes); }
    if (want("persist")) {
        PERSIST(false, 0, 2, "persist fold 2/CU"); PERSIST(false, 0, 3, "persist fold 3/CU"); PERSIST(false, 0, 4, "persist fold 4/CU");
        PERSIST(false, 0, 5, "persist fold 5/CU"); PERSIST(false, 0, 6, "persist fold 6/CU");
        PERSIST(false, 2, 4, "persist fold sc1 4/CU"); PERSIST(true, 0, 4, "persist copy 4/CU"); PERSIST(true, 2, 4, "persist copy sc1 4/CU");
    }
    if (want("ab")) {
        // interleaved A/B: per round and cache state, every variant once
        dim3 grid((NZ * NCH + 255) / 256, NF);
        dim3 gridf(((long long)NZ * H * (SX / 2) + 255) / 256, NF);
        for (int it = 0; it < rounds + 2; ++it)
            for (int mode = 0; mode < 2; ++mode)
                for (int v = 0; v < 8; ++v) {
                    if (mode == 0) hipLaunchKernelGGL(k_flush_dirty, dim3(8192), dim3(256), 0, 0, g_flush, FLUSH_N);
                    else hipLaunchKernelGGL(k_flush_clean, dim3(8192), dim3(256), 0, 0, g_flush, FLUSH_N, g_flush);
                    switch (v) {
                    case 0: hipLaunchKernelGGL((k_cols<false, 1, 0, false>), grid, dim3(256), 0, 0, g_ft, NZ, (unsigned long long*)nullptr); break;
                    case 1: hipLaunchKernelGGL((k_cols<false, 1, 2, false>), grid, dim3(256), 0, 0, g_ft, NZ, (unsigned long long*)nullptr); break;
                    case 2: hipLaunchKernelGGL((k_persist<false, 1, 0, 4>), dim3(1024), dim3(256), 0, 0, g_ft); break;
                    case 3: hipLaunchKernelGGL((k_persist<false, 1, 2, 4>), dim3(1024), dim3(256), 0, 0, g_ft); break;
                    case 4: hipLaunchKernelGGL((k_persist<false, 1, 0, 3>), dim3(768), dim3(256), 0, 0, g_ft); break;
                    case 5: hipLaunchKernelGGL((k_cols<true, 1, 0, false>), grid, dim3(256), 0, 0, g_ft, NZ, (unsigned long long*)nullptr); break;
                    case 6: hipLaunchKernelGGL(k_flatcopy<1>, gridf, dim3(256), 0, 0, g_ft); break;
                    case 7: hipLaunchKernelGGL((k_persist<true, 1, 2, 4>), dim3(1024), dim3(256), 0, 0, g_ft); break;
                    }
                }
        CHECK(hipDeviceSynchronize());
    }
    if (want("cols2")) {
        dim3 grid((NZ * (NCH / 2) + 255) / 256, NF);
        auto r = run3([&](hipEvent_t a, hipEvent_t b) { hipExtLaunchKernelGGL((k_cols2<false, 1>), grid, dim3(256), 0, 0, a, b, 0, g_ft, NZ); }, rounds);
        report("cols2 fold ntload", r, zbytes);
        auto r1 = run3([&](hipEvent_t a, hipEvent_t b) { hipExtLaunchKernelGGL((k_cols2<false, 0>), grid, dim3(256), 0, 0, a, b, 0, g_ft, NZ); }, rounds);
        report("cols2 fold plain", r1, zbytes);
        auto r2 = run3([&](hipEvent_t a, hipEvent_t b) { hipExtLaunchKernelGGL((k_cols2<true, 1>), grid, dim3(256), 0, 0, a, b, 0, g_ft, NZ); }, rounds);
        report("cols2 copy ntload", r2, zbytes);
        exp_cols<false, 1, 0>("fold ntload plain-store", rounds, zbytes);
    }
    if (want("lines")) {
        dim3 grid((unsigned)((nrows / 2 * 24 + 255) / 256), NF);
        for (int rw = 3; rw >= 1; --rw) {
            char nm[64]; snprintf(nm, sizeof nm, "lines rw=%d", rw);
            auto r = run3([&](hipEvent_t a, hipEvent_t b) { hipExtLaunchKernelGGL(k_lines, grid, dim3(256), 0, 0, a, b, 0, g_ft, nrows, rw); }, rounds);
            report(nm, r, pbytes);
        }
        dim3 grid0((unsigned)((nrows * 2 + 255) / 256), NF);
        auto r = run3([&](hipEvent_t a, hipEvent_t b) { hipExtLaunchKernelGGL(k_per0, grid0, dim3(256), 0, 0, a, b, 0, g_ft, nrows); }, rounds);
        report("per0 (product form)", r, pbytes);
    }
    if (exp == "seq") {
        // what the fold leaves behind for the periodic pass that follows it in a fill: plain stores vs write-through stores
        dim3 grid((NZ * NCH + 255) / 256, NF);
        dim3 gridp((unsigned)((nrows * 2 + 255) / 256), NF);
        for (int it = 0; it < rounds + 2; ++it)
            for (int v = 0; v < 2; ++v) {
                hipLaunchKernelGGL(k_flush_clean, dim3(8192), dim3(256), 0, 0, g_flush, FLUSH_N, g_flush);
                if (v == 0) hipLaunchKernelGGL((k_cols<false, 1, 0, false>), grid, dim3(256), 0, 0, g_ft, NZ, (unsigned long long*)nullptr);
                else        hipLaunchKernelGGL((k_cols<false, 1, 2, false>), grid, dim3(256), 0, 0, g_ft, NZ, (unsigned long long*)nullptr);
                hipLaunchKernelGGL(k_per0, gridp, dim3(256), 0, 0, g_ft, nrows);
            }
        CHECK(hipDeviceSynchronize());
    }
    if (want("timeline")) {
        for (int mode = 0; mode < 3; ++mode) timeline("fold", false, mode);
        timeline("copy", true, 1);
    }
    if (want("per")) {
        { dim3 grid((unsigned)((nrows * 2 + 255) / 256), NF);
          auto r = run3([&](hipEvent_t a, hipEvent_t b) { hipExtLaunchKernelGGL(k_per0, grid, dim3(256), 0, 0, a, b, 0, g_ft, nrows); }, rounds);
          report("per0 (product form)", r, pbytes);
          auto r3 = run3([&](hipEvent_t a, hipEvent_t b) { hipExtLaunchKernelGGL(k_per3, grid, dim3(256), 0, 0, a, b, 0, g_ft, nrows); }, rounds);
          report("per3 (nt stores)", r3, pbytes); }
        { dim3 grid((unsigned)((nrows * 2 / 2 + 255) / 256), NF);
          auto r = run3([&](hipEvent_t a, hipEvent_t b) { hipExtLaunchKernelGGL(k_per1<2>, grid, dim3(256), 0, 0, a, b, 0, g_ft, nrows); }, rounds);
          report("per1 K=2", r, pbytes); }
        { dim3 grid((unsigned)((nrows * 2 / 4 + 255) / 256), NF);
          auto r = run3([&](hipEvent_t a, hipEvent_t b) { hipExtLaunchKernelGGL(k_per1<4>, grid, dim3(256), 0, 0, a, b, 0, g_ft, nrows); }, rounds);
          report("per1 K=4", r, pbytes); }
        { dim3 grid((unsigned)(((nrows + 1) * 4 + 255) / 256), NF);
          auto r = run3([&](hipEvent_t a, hipEvent_t b) { hipExtLaunchKernelGGL(k_per2<1>, grid, dim3(256), 0, 0, a, b, 0, g_ft, nrows); }, rounds);
          report("per2 boundary K=1", r, pbytes); }
        { dim3 grid((unsigned)(((nrows + 1) * 4 / 4 + 255) / 256), NF);
          auto r = run3([&](hipEvent_t a, hipEvent_t b) { hipExtLaunchKernelGGL(k_per2<4>, grid, dim3(256), 0, 0, a, b, 0, g_ft, nrows); }, rounds);
          report("per2 boundary K=4", r, pbytes); }
        { dim3 grid((unsigned)(((nrows + 1) * 8 + 255) / 256), NF);
          auto r = run3([&](hipEvent_t a, hipEvent_t b) { hipExtLaunchKernelGGL(k_per4, grid, dim3(256), 0, 0, a, b, 0, g_ft, nrows); }, rounds);
          report("per4 boundary 8-B lanes", r, pbytes); }
    }
    if (exp == "perx") {
        // interleaved A/B of the cache-policy variants; device durations come from the rocprofv3 kernel trace of this run
        // (run.sh), the fetch / write bytes from its --pmc passes (run.sh perx pmc)
        dim3 g8((unsigned)(((nrows + 1) * 8 + 255) / 256), NF);
        dim3 g16((unsigned)((nrows * 2 + 255) / 256), NF);
        for (int it = 0; it < rounds + 2; ++it)
            for (int v = 0; v < 11; ++v) {
                hipLaunchKernelGGL(k_flush_clean, dim3(8192), dim3(256), 0, 0, g_flush, FLUSH_N, g_flush);
                switch (v) {
                case 0: hipLaunchKernelGGL(k_per0, g16, dim3(256), 0, 0, g_ft, nrows); break;
                case 1: hipLaunchKernelGGL(k_per6<0>, g16, dim3(256), 0, 0, g_ft, nrows); break;
                case 2: hipLaunchKernelGGL(k_per6<1>, g16, dim3(256), 0, 0, g_ft, nrows); break;
                case 3: hipLaunchKernelGGL((k_per5<0, 0>), g8, dim3(256), 0, 0, g_ft, nrows); break;
                case 4: hipLaunchKernelGGL((k_per5<1, 0>), g8, dim3(256), 0, 0, g_ft, nrows); break;
                case 5: hipLaunchKernelGGL((k_per5<2, 0>), g8, dim3(256), 0, 0, g_ft, nrows); break;
                case 6: hipLaunchKernelGGL((k_per5<3, 0>), g8, dim3(256), 0, 0, g_ft, nrows); break;
                case 7: hipLaunchKernelGGL((k_per5<0, 2>), g8, dim3(256), 0, 0, g_ft, nrows); break;
                case 8: hipLaunchKernelGGL((k_per5<0, 3>), g8, dim3(256), 0, 0, g_ft, nrows); break;
                case 9: hipLaunchKernelGGL((k_per5<3, 3>), g8, dim3(256), 0, 0, g_ft, nrows); break;
                case 10: hipLaunchKernelGGL((k_per5<1, 1>), g8, dim3(256), 0, 0, g_ft, nrows); break;
                }
            }
        CHECK(hipDeviceSynchronize());
    }
    CHECK(hipDeviceSynchronize());
    printf("done\n");
    return 0;
}
