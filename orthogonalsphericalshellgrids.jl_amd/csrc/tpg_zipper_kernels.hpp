// tpg_zipper_kernels.hpp -- device kernels and launch helpers of the halo fill (zipper fold, periodic x, fused / merged
// fills, seam pack / unpack).  Included by tpg_zipper.hip (the product entry points) and by tpg_testabi.hip (the
// test / bench-only entry points of libtripolar_hip_test.so, which instantiate the COPY = true probe form of the column
// kernel); everything here has internal linkage.
#pragma once
#include "tpg_common.hpp"
#include <hip/hip_ext.h>
#include <type_traits>

namespace tpg {
// optional per-launch device timestamps (hipExtLaunchKernelGGL start / stop events): set by the *_timed entry points for
// the duration of one call on this thread; the FIRST kernel the call launches carries them
extern thread_local hipEvent_t ev_start, ev_stop;
}

namespace {

using tpg::Geom;

struct FieldTable {
    void* ptr[TPG_MAX_FIELDS];
    int item0[TPG_MAX_FIELDS + 1];   // prefix sums of work items per field
    int xloc[TPG_MAX_FIELDS];        // 32-bit so that a wave-uniform field index reads them with s_load
    int yloc[TPG_MAX_FIELDS];
    int sign[TPG_MAX_FIELDS];
    int nfields;
};

struct ZipArgs {
    int Nx, Ny, Hx, Hy, Hz;
    int sx;
    long long plane;
    int kstart, kcount;
    int nchunks;     // chunks per destination row (vector kernel) or Nx (scalar kernel)
    int fix0;        // first chunk (vector) / element (scalar, 0-based) of the row-Ny substitution
};

template <typename T, int W> struct Vec;
template <> struct Vec<double, 2> {
    typedef double aligned_t __attribute__((ext_vector_type(2)));
    typedef double loose_t __attribute__((ext_vector_type(2), aligned(8)));
};
template <> struct Vec<float, 4> {
    typedef float aligned_t __attribute__((ext_vector_type(4)));
    typedef float loose_t __attribute__((ext_vector_type(4), aligned(4)));
};
// 8-B chunks: Float32 rows with Nx = 2 mod 4 do not split into 16-B chunks (GEN form only)
template <> struct Vec<float, 2> {
    typedef float aligned_t __attribute__((ext_vector_type(2)));
    typedef float loose_t __attribute__((ext_vector_type(2), aligned(4)));
};

// ---- GEN = true: chunks aligned with the INTERIOR, not with the parent ---------------------------------------------------------------
// The reference's model examples run halo = (5, 5, 5) (examples/bickley_jet.jl:21, examples/distributed_bickley_jet.jl:23): with an odd Hx
// the interior columns of a Float64 parent row start 8 B off the 16-B grid, and so do all its W-element chunks counted from i = 1.  The
// chunked kernels keep exactly their chunking -- chunk c of the interior is i = cW+1 .. cW+W, the x-halo chunks continue it outwards -- and
// only change the ACCESS TYPE: stores and the row-Ny / periodic loads go through the element-aligned `loose_t` (a 16-B access on an 8-B
// boundary), as the mirrored x-Face windows always did.  Measured on 3600 x 1800 x 75, 4 fields: the fold with such stores takes 17.8 us at
// halo 5 (90.7 MB) against 14.9 us at halo 4 (73.4 MB) -- 0.96 of the halo-4 time per byte -- and 17.6-17.8 us with parent-aligned chunks
// and element-wise edge chunks (profiles/r06/halo5_periodic_variants.json): alignment of the stores buys nothing, uniform waves do.
// What does not fit a chunk is the r = Hx mod W outermost columns of each x halo (r = 1 at halo 5): the composed-map kernels give them
// element items of their own (S items: blocks / item ranges apart from the chunk items, so that no wave runs both code paths).
// GEN also serves 16-B-misaligned field pointers (any element-aligned pointer works) and Float32 rows with Nx = 2 mod 4 (W = 2).
// GEN = false is the geometry the kernels had before (Hx, Nx multiples of W, 16-B aligned fields) and compiles to the code it was.
// The PERIODIC part of a GEN fill (the rows below the fold; tpg_periodic_x_fill) is one element per item -- item (row, h) copies
// c[Nx+h] -> c[h] and c[Hx+h] -> c[Hx+Nx+h] --, which measured best at halo 5 (one merged launch, 4 fields, cold): element items 62.7-63.6 us;
// 3 uniform 16-B items per row, the third overlapping the second by one element, 63.2-63.4 us; aligned 16-B items + one leftover item (a
// divergent wave: two dependent memory round trips) 81-82 us; one item per (row, side) 105 us; one item per row 91 us.  The pass is bound by
// 128-B line transfers (2.1 lines per row at this pitch, 1.5 at halo 4): the finest uniform items keep the most requests in flight.

// composed map of ONE written cell at parent column ii: wrapped column iw (1..Nx), mirrored source column ip (1..Nx), factor se
// (x-Center i' = Nx-iw+1; x-Face i' = Nx-iw+2, iw = 1 wrapping to i' = 1 with |sign|: zipper_boundary_condition.jl:73-75, :90-92)
template <typename T>
__device__ __forceinline__ void fold_column(int ii, int Nx, int Hx, int xl, T s, T as, int& iw, int& ip, T& se)
{
    const int i = ii - Hx + 1;
    iw = i < 1 ? i + Nx : (i > Nx ? i - Nx : i);
    ip = Nx - iw + 1 + (xl == TPG_FACE ? 1 : 0);
    se = s;
    if (ip > Nx) { ip -= Nx; se = as; }
}

__device__ __forceinline__ int find_field(const FieldTable& ft, int item)
{
    int f = 0;
#pragma unroll 1
    while (f + 1 < ft.nfields && item >= ft.item0[f + 1]) ++f;
    return f;
}

// ---- vector kernel: W elements (16 B) per work item ----------------------------------------
template <typename T, int W>
__global__ __launch_bounds__(256) void k_zipper_vec(FieldTable ft, ZipArgs a)
{
    typedef typename Vec<T, W>::aligned_t vec_t;
    typedef typename Vec<T, W>::loose_t lvec_t;
    int item = blockIdx.x * blockDim.x + threadIdx.x;
    if (item >= ft.item0[ft.nfields]) return;
    const int f = find_field(ft, item);
    item -= ft.item0[f];
    const int xl = ft.xloc[f], yl = ft.yloc[f];
    const int sgn = ft.sign[f];
    T* __restrict__ c = static_cast<T*>(ft.ptr[f]);

    // items of one level: Hy full rows, then (y-Center only) the upper part of row Ny
    const int nfix = (yl == TPG_CENTER) ? a.nchunks - a.fix0 : 0;
    const int per_level = a.Hy * a.nchunks + nfix;
    const int kk = item / per_level;
    int r = item - kk * per_level;
    int jrow, ch;            // destination row slot: 1..Hy halo rows, 0 = row Ny
    if (r < a.Hy * a.nchunks) { jrow = r / a.nchunks + 1; ch = r - (jrow - 1) * a.nchunks; }
    else { jrow = 0; ch = a.fix0 + (r - a.Hy * a.nchunks); }

    const int k = a.kstart + kk;                                   // 1-based level
    const int jdst = a.Ny + jrow;
    // source row: y-Face  Ny - j + 1, y-Center  Ny - j  (row fix: Ny)        (:80,:97,:115,:130)
    const int jsrc = (jrow == 0) ? a.Ny : ((yl == TPG_FACE) ? a.Ny - jrow + 1 : a.Ny - jrow);
    const long long kbase = a.plane * (k + a.Hz - 1);
    T* dst = c + kbase + (long long)a.sx * (jdst + a.Hy - 1) + a.Hx;     // -> element i = 1
    const T* src = c + kbase + (long long)a.sx * (jsrc + a.Hy - 1) + a.Hx;

    const int i = ch * W + 1;                                      // first destination index
    T out[W];
    if (xl == TPG_CENTER) {
        // i' = Nx - i + 1: destination i..i+W-1 <- source Nx-i+1 .. Nx-i-W+2 (descending)
        const vec_t v = *reinterpret_cast<const vec_t*>(src + (a.Nx - i - W + 1));
#pragma unroll
        for (int e = 0; e < W; ++e) out[e] = (T)sgn * v[W - 1 - e];
    } else {
        // i' = Nx - i + 2, wrapping to 1 with |sign| at i = 1           (:73-75, :90-92)
        if (ch == 0) {
            out[0] = (T)(sgn < 0 ? -sgn : sgn) * src[0];
#pragma unroll
            for (int e = 1; e < W; ++e) out[e] = (T)sgn * src[a.Nx - e];
        } else {
            const lvec_t v = *reinterpret_cast<const lvec_t*>(src + (a.Nx - i - W + 2));
#pragma unroll
            for (int e = 0; e < W; ++e) out[e] = (T)sgn * v[W - 1 - e];
        }
    }
    if (jrow == 0) {
        // c[i,Ny] = ifelse(i > Nx/2, sign*c[i',Ny], c[i,Ny])            (:102, :135)
        const vec_t old = *reinterpret_cast<const vec_t*>(dst + (i - 1));
#pragma unroll
        for (int e = 0; e < W; ++e) if (i + e <= a.Nx / 2) out[e] = old[e];
    }
    vec_t o;
#pragma unroll
    for (int e = 0; e < W; ++e) o[e] = out[e];
    *reinterpret_cast<vec_t*>(dst + (i - 1)) = o;
}

// ---- column kernel: one thread owns one 16-B column chunk of one (field, level) and folds ALL
// Hy halo rows (+ the row-Ny substitution): the Hy (+2) independent 16-B loads are issued before
// the first store, so a wave keeps (Hy+2) KiB in flight instead of 1 KiB -- the fold moves only
// ~70 MB per launch, which makes it latency- rather than bandwidth-limited unless every wave
// carries many outstanding requests.
// Loads are streaming (non-temporal): the fold's sources are not reused soon, and a predecessor that left the
// caches full of dirty lines costs 17.7 instead of 23.6 us that way; stores are plain (tools/fillbench).
// COPY = true is the bench's same-shape copy ceiling (tpg_zipper_copy_probe): identical rows, bytes and launch
// shape, but destination column = source column and no sign.
template <typename T, int W, int HY, bool COPY, bool GEN = false>
__global__ __launch_bounds__(256) void k_zipper_cols(FieldTable ft, ZipArgs a)
{
    static_assert(!(COPY && GEN), "the copy probe exists for the chunk-aligned geometry only");
    typedef typename Vec<T, W>::aligned_t vec_t;
    typedef typename Vec<T, W>::loose_t lvec_t;
    typedef typename std::conditional<GEN, lvec_t, vec_t>::type svec_t;   // destination chunks: 16-B aligned, or (GEN) element-aligned
    const int f = blockIdx.y;                                      // wave-uniform: table reads are scalar loads
    const int item = blockIdx.x * blockDim.x + threadIdx.x;
    if (item >= a.kcount * a.nchunks) return;
    const int xl = ft.xloc[f], yl = ft.yloc[f];
    const int sgn = ft.sign[f];
    T* __restrict__ c = static_cast<T*>(ft.ptr[f]);
    const int kk = item / a.nchunks;
    const int ch = item - kk * a.nchunks;
    const int k = a.kstart + kk;
    const int i = ch * W + 1;
    T* lvl = c + a.plane * (k + a.Hz - 1) + a.Hx;                  // element (i = 1, parent row 0)
    const long long sx = a.sx;
    const int prow_ny = a.Ny + a.Hy - 1;                           // parent row of logical row Ny
    const int ysh = (yl == TPG_FACE) ? 1 : 0;                      // source row Ny - j + ysh
    const bool fix = (yl == TPG_CENTER) && (ch >= a.fix0);
    // x-Face, first chunk: element i = 1 wraps to i' = 1 with |sign| (:73-75, :90-92); the mirrored
    // window then starts one element into the east halo, whose (unused) value is replaced below
    const bool wrap = !COPY && (xl == TPG_FACE) && (ch == 0);

    // mirrored source window: x-Center i' = Nx-i+1, x-Face i' = Nx-i+2 (one element to the right)
    const int soff = COPY ? i - 1 : a.Nx - i - W + 1 + (xl == TPG_FACE ? 1 : 0);
    vec_t v[HY];
    T w0[HY];
#pragma unroll
    for (int jr = 1; jr <= HY; ++jr) {
        const T* row = lvl + sx * (prow_ny - jr + ysh);
        const lvec_t* p = reinterpret_cast<const lvec_t*>(row + soff);
        v[jr - 1] = __builtin_nontemporal_load(p);
        w0[jr - 1] = wrap ? row[0] : (T)0;
    }
    vec_t vf = {}, old = {};
    if (fix) {
        vf = *reinterpret_cast<const lvec_t*>(lvl + sx * prow_ny + soff);
        if (i <= a.Nx / 2)       // only the chunk that straddles Nx/2 keeps part of the old row
            old = *reinterpret_cast<const svec_t*>(lvl + sx * prow_ny + (i - 1));
    }
    const T s = (T)sgn, as = (T)(sgn < 0 ? -sgn : sgn);
#pragma unroll
    for (int jr = 1; jr <= HY; ++jr) {
        vec_t o;
#pragma unroll
        for (int e = 0; e < W; ++e) o[e] = COPY ? v[jr - 1][e] : s * v[jr - 1][W - 1 - e];
        if (wrap) o[0] = as * w0[jr - 1];
        *reinterpret_cast<svec_t*>(lvl + sx * (prow_ny + jr) + (i - 1)) = o;
    }
    if (fix) {
        // c[i,Ny] = ifelse(i > Nx/2, sign*c[i',Ny], c[i,Ny]) (:102,:135); i = 1 is never > Nx/2
        vec_t o;
#pragma unroll
        for (int e = 0; e < W; ++e) o[e] = (i + e > a.Nx / 2) ? (COPY ? vf[e] : s * vf[W - 1 - e]) : old[e];
        *reinterpret_cast<svec_t*>(lvl + sx * prow_ny + (i - 1)) = o;
    }
}

// ---- scalar kernel: any geometry / alignment ---------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_zipper_scalar(FieldTable ft, ZipArgs a)
{
    int item = blockIdx.x * blockDim.x + threadIdx.x;
    if (item >= ft.item0[ft.nfields]) return;
    const int f = find_field(ft, item);
    item -= ft.item0[f];
    const int xl = ft.xloc[f], yl = ft.yloc[f];
    int sgn = ft.sign[f];
    T* c = static_cast<T*>(ft.ptr[f]);
    const int nfix = (yl == TPG_CENTER) ? a.Nx - a.fix0 : 0;
    const int per_level = a.Hy * a.Nx + nfix;
    const int kk = item / per_level;
    int r = item - kk * per_level;
    int jrow, i;
    if (r < a.Hy * a.Nx) { jrow = r / a.Nx + 1; i = r - (jrow - 1) * a.Nx + 1; }
    else { jrow = 0; i = a.fix0 + (r - a.Hy * a.Nx) + 1; }
    const int k = a.kstart + kk;
    int ip = (xl == TPG_FACE) ? a.Nx - i + 2 : a.Nx - i + 1;
    if (ip > a.Nx) { sgn = sgn < 0 ? -sgn : sgn; ip -= a.Nx; }
    const int jdst = a.Ny + jrow;
    const int jsrc = (jrow == 0) ? a.Ny : ((yl == TPG_FACE) ? a.Ny - jrow + 1 : a.Ny - jrow);
    const long long kbase = a.plane * (k + a.Hz - 1);
    c[kbase + (long long)a.sx * (jdst + a.Hy - 1) + (i + a.Hx - 1)] =
        (T)sgn * c[kbase + (long long)a.sx * (jsrc + a.Hy - 1) + (ip + a.Hx - 1)];
}

// ---- periodic west/east halos over every row and level of the parent ---------------------------
struct PerArgs { int Nx, Hx, sx; long long nrows; int nfields; };
struct PtrTable { void* ptr[TPG_MAX_FIELDS]; };

template <typename T>
__global__ __launch_bounds__(256) void k_periodic_x(PtrTable pt, PerArgs a)
{
    // item = (row, h): copies c[Nx-Hx+1+h] -> c[1-Hx+h] and c[1+h] -> c[Nx+1+h]
    long long item = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    long long per_field = a.nrows * a.Hx;
    if (item >= per_field * a.nfields) return;
    int f = (int)(item / per_field);
    item -= (long long)f * per_field;
    long long row = item / a.Hx;
    int h = (int)(item - row * a.Hx);
    T* c = static_cast<T*>(pt.ptr[f]) + row * a.sx;
    c[h] = c[a.Nx + h];
    c[a.Hx + a.Nx + h] = c[a.Hx + h];
}

// 16-byte form (Hx and Nx multiples of the 16-B element count, 16-B aligned fields): one thread
// moves one 16-B chunk of the west halo and one of the east halo of a row; grid.y = field.
template <typename V>
__global__ __launch_bounds__(256) void k_periodic_x_vec(PtrTable pt, PerArgs a, int cpr /* chunks per side */, int epc /* elems per chunk */)
{
    long long item = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (item >= a.nrows * cpr) return;
    long long row = item / cpr;
    int v = (int)(item - row * cpr);
    char* base = static_cast<char*>(pt.ptr[blockIdx.y]);
    const size_t esz = 16 / epc;
    V* c = reinterpret_cast<V*>(base + (size_t)row * a.sx * esz);
    const int nxc = a.Nx / epc, hxc = a.Hx / epc;
    V w = c[nxc + v];            // interior column Nx-Hx+1.. (chunk units: Hx + Nx - Hx = Nx elements in)
    V e = c[hxc + v];            // interior column 1..
    c[v] = w;
    c[hxc + nxc + v] = e;
}

// ---- fused fill for small fields: zipper + periodic x in ONE launch ------------------------------
// A 2-D field (free surface, barotropic U, V: a few hundred KB) is filled in a few microseconds, so the
// two-launch sequence is pure launch latency -- and a split-explicit free surface does 3 such fills per
// substep, ~30 substeps per baroclinic step.  Every cell the sequence
// "fold north (zipper_boundary_condition.jl:70-138), then periodic west/east" writes is a function of
// ORIGINAL interior values only, so the two maps compose and one thread per written cell can apply
// them directly, race-free:
//   north halo row Ny+dj, any column i (corners included; iw = i wrapped into 1..Nx):
//        c[i, Ny+dj] = s' c[i'(iw), Ny-dj (+1 for y-Face)]
//   row Ny of y-Center fields, iw > Nx/2 (interior cell: the substitution; halo cell: its periodic copy):
//        c[i, Ny]    = s' c[i'(iw), Ny]
//   west / east halo cell of any other row (and of the z-halo levels, which the zipper skips):
//        c[i, j]     = c[iw, j]
// with i' = Nx-iw+1 (x-Center) or Nx-iw+2 (x-Face; iw = 1 wraps to i' = 1 with s' = |s|).  The only cell
// that is both read and written, the x-Face self-map i = Nx/2+1 of row Ny, is read by its own thread as
// long as it is no periodic source, i.e. Nx >= 2 Hx + 2; fold sources stay clear of written rows for
// Ny >= 2 Hy + 2.  Other geometries take the two-launch path.
struct FusedArgs { int Nx, Ny, Hx, Hy, Hz, Nz, sx, sy; long long plane; int per_level; };

template <typename T>
__global__ __launch_bounds__(256) void k_fill_fused(FieldTable ft, FusedArgs a)
{
    const int f = blockIdx.y;
    int item = blockIdx.x * blockDim.x + threadIdx.x;
    const int nlev = a.Nz + 2 * a.Hz;
    if (item >= a.per_level * nlev) return;
    const int lev = item / a.per_level;
    item -= lev * a.per_level;
    const int xl = ft.xloc[f], yl = ft.yloc[f];
    int sgn = ft.sign[f];
    T* c = static_cast<T*>(ft.ptr[f]) + a.plane * lev;
    const bool zipped = lev >= a.Hz && lev < a.Hz + a.Nz;
    // candidate cells of a level: (Hy+1) full rows from row Ny up, then the 2 Hx halo columns of the rows below
    int ii, jj;
    const int top = (a.Hy + 1) * a.sx;
    if (item < top) { jj = a.Ny + a.Hy - 1 + item / a.sx; ii = item % a.sx; }
    else { item -= top; jj = item / (2 * a.Hx); const int h = item - jj * 2 * a.Hx; ii = h < a.Hx ? h : a.Nx + h; }
    const int i = ii - a.Hx + 1, j = jj - a.Hy + 1;
    const int iw = i < 1 ? i + a.Nx : (i > a.Nx ? i - a.Nx : i);
    int ip = (xl == TPG_FACE) ? a.Nx - iw + 2 : a.Nx - iw + 1;
    if (ip > a.Nx) { sgn = sgn < 0 ? -sgn : sgn; ip -= a.Nx; }
    T v;
    if (zipped && j > a.Ny) {
        const int dj = j - a.Ny;
        const int jsrc = (yl == TPG_FACE) ? a.Ny - dj + 1 : a.Ny - dj;
        v = (T)sgn * c[(long long)a.sx * (jsrc + a.Hy - 1) + (ip + a.Hx - 1)];
    } else if (zipped && j == a.Ny && yl == TPG_CENTER && iw > a.Nx / 2) {
        v = (T)sgn * c[(long long)a.sx * jj + (ip + a.Hx - 1)];
    } else if (i != iw) {
        v = c[(long long)a.sx * jj + (iw + a.Hx - 1)];
    } else {
        return;                                   // interior cell that no fill touches
    }
    c[(long long)a.sx * jj + ii] = v;
}


// 16-byte form of the fused fill (rows chunkable: Hx and Nx multiples of W, 16-B aligned fields): one thread per 16-B chunk of
// a written row instead of one per cell -- half (f64) / a quarter (f32) of the threads and index arithmetic, whole-chunk loads
// and stores.  Same composed map as k_fill_fused; a chunk never straddles the interior / x-halo boundary (Hx % W == 0), so its
// W elements share one wrapped base column iw0 and its mirrored source is ONE contiguous, reversed window.
//   items of a level:  A = (Hy+1) rows from row Ny up  x  sx/W chunks (all columns, corners included)
//                      B = the Ny+Hy-1 rows below       x  2Hx/W x-halo chunks (plain periodic copies)
// GEN = true (see the note on GEN above): the chunks of A start at parent column r = Hx mod W (cpr = 2 (Hx / W) + Nx / W of them, all whole)
// and are stored element-aligned; the 2 r outermost halo columns of the (Hy+1) rows are S items, itemsA <= item < itemsA + itemsS, one
// element each through the scalar composed map; B has hc = Hx items per row, one element of each side per item.
struct FusedVecArgs { int Nx, Ny, Hx, Hy, Hz, Nz, sx; long long plane; int cpr, hc, itemsA, per_level; int r, itemsS; };

template <typename T, int W, bool GEN = false>
__global__ __launch_bounds__(256) void k_fill_fused_vec(FieldTable ft, FusedVecArgs a)
{
    typedef typename Vec<T, W>::aligned_t vec_t;
    typedef typename Vec<T, W>::loose_t lvec_t;
    typedef typename std::conditional<GEN, lvec_t, vec_t>::type svec_t;   // destination chunks and their periodic sources: aligned, or (GEN) element-aligned
    const int f = blockIdx.y;
    int item = blockIdx.x * blockDim.x + threadIdx.x;
    const int nlev = a.Nz + 2 * a.Hz;
    if (item >= a.per_level * nlev) return;
    const int lev = item / a.per_level;
    item -= lev * a.per_level;
    T* c = static_cast<T*>(ft.ptr[f]) + a.plane * lev;
    if constexpr (GEN) {
        if (item >= a.itemsA && item < a.itemsA + a.itemsS) {
            // ---- S: one of the r outermost columns of the west / east halo, rows Ny .. Ny+Hy: the composed map for one cell
            item -= a.itemsA;
            const int jr = item / (2 * a.r), q = item - jr * 2 * a.r;
            const int ii = q < a.r ? q : a.sx - 2 * a.r + q;          // west: columns 0 .. r-1; east: sx-r .. sx-1
            const int xl = ft.xloc[f], yl = ft.yloc[f], sgn = ft.sign[f];
            const bool zipped = lev >= a.Hz && lev < a.Hz + a.Nz;
            T* rowNy = c + (long long)a.sx * (a.Ny + a.Hy - 1);
            T* drow = rowNy + (long long)a.sx * jr;
            int iw, ip; T se;
            fold_column(ii, a.Nx, a.Hx, xl, (T)sgn, (T)(sgn < 0 ? -sgn : sgn), iw, ip, se);
            if (zipped && jr > 0) drow[ii] = se * rowNy[(ip + a.Hx - 1) - (long long)a.sx * (jr - (yl == TPG_FACE ? 1 : 0))];
            else if (zipped && yl == TPG_CENTER && iw > a.Nx / 2) drow[ii] = se * rowNy[ip + a.Hx - 1];            // row Ny (:102, :135)
            else drow[ii] = drow[iw + a.Hx - 1];                                                                  // plain periodic copy
            return;
        }
        if (item >= a.itemsA) item -= a.itemsS;
    }
    if (item >= a.itemsA) {
        // ---- B: x-halo chunk of a row below row Ny: west halo <- east interior columns, east halo <- west interior
        item -= a.itemsA;
        const int jj = item / a.hc, q = item - jj * a.hc;
        if constexpr (GEN) {                                        // one element of each side per item: hc = Hx
            T* r = c + (long long)a.sx * jj;
            const T w = r[a.Nx + q], e = r[a.Hx + q];
            r[q] = w;
            r[a.Hx + a.Nx + q] = e;
            return;
        }
        const int hw = a.hc >> 1;                                   // chunks per halo side
        T* row = c + (long long)a.sx * jj;
        const int dst = q < hw ? q * W : a.Hx + a.Nx + (q - hw) * W;
        const int src = q < hw ? a.Nx + q * W : a.Hx + (q - hw) * W;
        *reinterpret_cast<vec_t*>(row + dst) = *reinterpret_cast<const vec_t*>(row + src);
        return;
    }
    // ---- A: rows Ny .. Ny+Hy, every chunk of the padded row
    const int xl = ft.xloc[f], yl = ft.yloc[f];
    const int sgn = ft.sign[f];
    const int jr = item / a.cpr, ch = item - jr * a.cpr;            // jr = 0: row Ny, 1..Hy: halo rows
    const int ii0 = (GEN ? a.r : 0) + ch * W;                       // parent column of the chunk's first element
    const int i = ii0 - a.Hx + 1;                                   // its logical column
    const bool west = ii0 < a.Hx, east = ii0 >= a.Hx + a.Nx;
    const int iw0 = west ? i + a.Nx : (east ? i - a.Nx : i);        // wrapped into 1..Nx (whole chunk: Hx % W == 0)
    const bool halo = west || east;
    const bool zipped = lev >= a.Hz && lev < a.Hz + a.Nz;
    T* rowNy = c + (long long)a.sx * (a.Ny + a.Hy - 1);
    T* drow = rowNy + (long long)a.sx * jr;
    const T s = (T)sgn, as = (T)(sgn < 0 ? -sgn : sgn);
    // mirrored window of columns iw0 .. iw0+W-1: x-Center i' = Nx-iw+1, x-Face i' = Nx-iw+2 (descending in iw)
    const int wlo = a.Nx - iw0 - W + 1 + (xl == TPG_FACE ? 1 : 0) + a.Hx;        // parent column of the window's lowest element
    const bool wrap = (xl == TPG_FACE) && iw0 == 1;                 // i' = Nx+1 -> 1 with |sign| (:73-75, :90-92)
    vec_t o;
    if (zipped && jr > 0) {
        const T* srow = rowNy - (long long)a.sx * (jr - (yl == TPG_FACE ? 1 : 0));   // row Ny-j (y-Center) / Ny-j+1 (y-Face)
        // the wrap chunk's mirrored window would start one element into the east halo (a cell this launch writes):
        // it is read one element lower instead, i.e. wholly inside the interior, and indexed accordingly
        const vec_t v = *reinterpret_cast<const lvec_t*>(srow + wlo - (wrap ? 1 : 0));
        o[0] = wrap ? as * srow[a.Hx] : s * v[W - 1];
#pragma unroll
        for (int e = 1; e < W; ++e) o[e] = s * (wrap ? v[W - e] : v[W - 1 - e]);
    } else if (zipped && yl == TPG_CENTER) {
        // row Ny of a y-Center field: columns iw > Nx/2 take the substitution (interior) or its periodic image (west halo);
        // columns iw <= Nx/2 are untouched (interior) or a plain periodic copy (east halo)      (:102, :135)
        const bool any_hi = iw0 + W - 1 > a.Nx / 2, any_lo = iw0 <= a.Nx / 2;
        if (!any_hi && !halo) return;
        vec_t v = {}, pl = {};
        if (any_hi) v = *reinterpret_cast<const lvec_t*>(rowNy + wlo);
        if (any_lo) pl = *reinterpret_cast<const svec_t*>(rowNy + (iw0 + a.Hx - 1));
#pragma unroll
        for (int e = 0; e < W; ++e) o[e] = (iw0 + e > a.Nx / 2) ? s * v[W - 1 - e] : pl[e];
    } else {
        if (!halo) return;                                          // z-halo level, or row Ny of a y-Face field: periodic x only
        o = *reinterpret_cast<const svec_t*>(drow + (iw0 + a.Hx - 1));
    }
    *reinterpret_cast<svec_t*>(drow + ii0) = o;
}


// ---- merged fill for large fields: the whole fill_halo_regions! (zipper -> periodic x) in ONE launch -----------------
// Blocks [0, blocksA): the column kernel over the FULL padded width -- a thread owns one 16-B column chunk of one
// (field, level), x-halo chunks included, and writes rows Ny .. Ny+Hy of it: halo rows through the composed map
// (wrapped base column iw0, mirrored source window: the cell the periodic pass would have copied from the folded row),
// row Ny as the substitution / its periodic image (y-Center) or the plain periodic copy of its x halos (y-Face).
// Blocks [blocksA, ..): the periodic pass of every other (level, row): all rows of the z-halo levels, rows below row Ny of
// the folded levels.  The two parts touch disjoint cells and read only interior cells nobody writes, so they need no order: one kernel boundary less per fill, and the fold's launch
// ramp and tail hide under the periodic pass's stream.  Same geometry conditions as k_fill_fused.
// GEN = true (see the note on GEN above; the reference's model halo (5, 5, 5)): the chunks of A start at parent column r = Hx mod W
// (cprA = 2 (Hx / W) + Nx / W per level, all whole) and are stored element-aligned; blocks [blocksA, blocksA + blocksS) are the S items --
// one thread per (level, one of the 2 r outermost halo columns), rows Ny .. Ny+Hy through the scalar composed map --; in B a.hw = Hx items
// per row, one element of each side per item.
struct MergedArgs { int Nx, Ny, Hx, Hy, Hz, Nz, sx, sy; long long plane; int cprA, hw; unsigned blocksA; long long rowsB; int r; unsigned blocksS; };

template <typename T, int W, int HY, bool GEN = false>
__global__ __launch_bounds__(256) void k_fill_merged(FieldTable ft, MergedArgs a)
{
    typedef typename Vec<T, W>::aligned_t vec_t;
    typedef typename Vec<T, W>::loose_t lvec_t;
    typedef typename std::conditional<GEN, lvec_t, vec_t>::type svec_t;   // destination chunks and their periodic sources: aligned, or (GEN) element-aligned
    const int f = blockIdx.y;
    T* __restrict__ c = static_cast<T*>(ft.ptr[f]);
    if constexpr (GEN) {
        if (blockIdx.x >= a.blocksA && blockIdx.x < a.blocksA + a.blocksS) {
            // ---- S: one of the r outermost columns of the west / east halo of one level, rows Ny .. Ny+Hy
            const int item = (blockIdx.x - a.blocksA) * 256 + threadIdx.x;
            if (item >= a.Nz * 2 * a.r) return;
            const int kk = item / (2 * a.r), q = item - kk * 2 * a.r;
            const int ii = q < a.r ? q : a.sx - 2 * a.r + q;          // west: columns 0 .. r-1; east: sx-r .. sx-1
            const int xl = ft.xloc[f], yl = ft.yloc[f], sgn = ft.sign[f];
            const long long sx = a.sx;
            T* rowNy = c + a.plane * (kk + a.Hz) + sx * (a.Ny + a.Hy - 1);
            const int ysh = (yl == TPG_FACE) ? 1 : 0;
            int iw, ip; T se;
            fold_column(ii, a.Nx, a.Hx, xl, (T)sgn, (T)(sgn < 0 ? -sgn : sgn), iw, ip, se);
            T q0, qv[HY];
#pragma unroll
            for (int jr = 1; jr <= HY; ++jr) qv[jr - 1] = rowNy[(ip + a.Hx - 1) - sx * (jr - ysh)];
            const bool fixed = yl == TPG_CENTER && iw > a.Nx / 2;      // (:102, :135): the periodic image of the substituted cell
            q0 = fixed ? rowNy[ip + a.Hx - 1] : rowNy[iw + a.Hx - 1];
#pragma unroll
            for (int jr = 1; jr <= HY; ++jr) rowNy[sx * jr + ii] = se * qv[jr - 1];
            rowNy[ii] = fixed ? se * q0 : q0;
            return;
        }
    }
    if (blockIdx.x >= a.blocksA) {
        // ---- B: periodic x of one (level, row): chunk v of the west halo and of the east halo
        const long long item = (long long)(blockIdx.x - a.blocksA - (GEN ? a.blocksS : 0u)) * 256 + threadIdx.x;
        if (item >= a.rowsB * a.hw) return;
        const long long row = item / a.hw;                          // over (level, parent row)
        const int v = (int)(item - row * a.hw);
        const int lev = (int)(row / a.sy), jj = (int)(row - (long long)lev * a.sy);
        if (lev >= a.Hz && lev < a.Hz + a.Nz && jj >= a.Ny + a.Hy - 1) return;      // rows Ny.. of a folded level: part A
        if constexpr (GEN) {                                        // one element of each side per item: a.hw = Hx
            T* r = c + row * a.sx;
            const T w = r[a.Nx + v], e = r[a.Hx + v];
            r[v] = w;
            r[a.Hx + a.Nx + v] = e;
            return;
        }
        vec_t* r = reinterpret_cast<vec_t*>(c + row * a.sx);
        const int nxc = a.Nx / W, hxc = a.Hx / W;
        const vec_t w = r[nxc + v], e = r[hxc + v];
        r[v] = w;
        r[hxc + nxc + v] = e;
        return;
    }
    // ---- A
    const int item = blockIdx.x * 256 + threadIdx.x;
    if (item >= a.Nz * a.cprA) return;
    const int xl = ft.xloc[f], yl = ft.yloc[f];
    const int sgn = ft.sign[f];
    const int kk = item / a.cprA, ch = item - kk * a.cprA;
    const int ii0 = (GEN ? a.r : 0) + ch * W;                       // parent column of the chunk
    const int i = ii0 - a.Hx + 1;
    const bool west = ii0 < a.Hx, east = ii0 >= a.Hx + a.Nx;
    const int iw0 = west ? i + a.Nx : (east ? i - a.Nx : i);        // wrapped into 1..Nx
    const bool halo = west || east;
    T* lvl = c + a.plane * (kk + a.Hz);                             // parent (column 0, row 0) of level k = kk+1
    const long long sx = a.sx;
    const int prow_ny = a.Ny + a.Hy - 1;
    const int ysh = (yl == TPG_FACE) ? 1 : 0;
    const int wlo = a.Nx - iw0 - W + 1 + (xl == TPG_FACE ? 1 : 0) + a.Hx;     // parent column of the mirrored window
    const bool wrap = (xl == TPG_FACE) && iw0 == 1;
    T* rowNy = lvl + sx * prow_ny;
    vec_t v[HY];
    T w0[HY];
#pragma unroll
    for (int jr = 1; jr <= HY; ++jr) {
        const T* row = lvl + sx * (prow_ny - jr + ysh);
        // wrap chunk: the mirrored window would start one element into the east halo, a cell the periodic blocks of this
        // launch write; it is read one element lower (wholly interior) and indexed accordingly below
        v[jr - 1] = __builtin_nontemporal_load(reinterpret_cast<const lvec_t*>(row + wlo - (wrap ? 1 : 0)));
        w0[jr - 1] = wrap ? row[a.Hx] : (T)0;
    }
    const bool any_hi = (yl == TPG_CENTER) && (iw0 + W - 1 > a.Nx / 2);
    const bool need_pl = (yl == TPG_CENTER) ? ((iw0 <= a.Nx / 2) && (halo || any_hi)) : halo;
    vec_t vf = {}, pl = {};
    if (any_hi) vf = *reinterpret_cast<const lvec_t*>(rowNy + wlo);
    if (need_pl) pl = *reinterpret_cast<const svec_t*>(rowNy + (iw0 + a.Hx - 1));
    const T s = (T)sgn, as = (T)(sgn < 0 ? -sgn : sgn);
#pragma unroll
    for (int jr = 1; jr <= HY; ++jr) {
        vec_t o;
        o[0] = wrap ? as * w0[jr - 1] : s * v[jr - 1][W - 1];
#pragma unroll
        for (int e = 1; e < W; ++e) o[e] = s * (wrap ? v[jr - 1][W - e] : v[jr - 1][W - 1 - e]);
        *reinterpret_cast<svec_t*>(rowNy + sx * jr + ii0) = o;
    }
    if (any_hi || need_pl) {
        vec_t o;
#pragma unroll
        for (int e = 0; e < W; ++e) o[e] = (any_hi && iw0 + e > a.Nx / 2) ? s * vf[W - 1 - e] : pl[e];
        *reinterpret_cast<svec_t*>(rowNy + ii0) = o;
    }
}

// ---- latitude-band message pack / unpack --------------------------------------------------------
struct PackArgs { int sx, sy, nlev, Hy, row0; long long plane; int nfields; int chunk_elems; };

template <typename V, bool PACK>
__global__ __launch_bounds__(256) void k_pack(PtrTable pt, V* buffer, PackArgs a)
{
    // message layout: [field][level][Hy rows][sx]; one item = one V (16 B, 8 B or one element: the widest the rows and bases are aligned to)
    long long item = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long per_level = (long long)a.Hy * a.sx / a.chunk_elems;
    const long long per_field = per_level * a.nlev;
    if (item >= per_field * a.nfields) return;
    int f = (int)(item / per_field);
    long long r = item - (long long)f * per_field;
    int lev = (int)(r / per_level);
    long long w = r - (long long)lev * per_level;                  // chunk inside the Hy x sx slab
    V* field = static_cast<V*>(pt.ptr[f]) + ((long long)a.plane * lev + (long long)a.sx * a.row0) / a.chunk_elems + w;
    if (PACK) buffer[item] = *field;
    else *field = buffer[item];
}

// Float32 rows whose length is no multiple of 16 B (sx = 2 mod 4: Nx = 3600 at the reference's model halo 5), or bases off the 16-B
// grid: the Hy x sx slab of one (field, level) is ONE contiguous run on both sides, so it moves as 16-B chunks counted from the slab's
// first element through element-aligned accesses (free on this device, see GEN above); the last chunk of a slab may be short.
template <typename T, int W, bool PACK>
__global__ __launch_bounds__(256) void k_pack_loose(PtrTable pt, T* buffer, PackArgs a)
{
    typedef typename Vec<T, W>::loose_t lvec_t;
    long long item = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int slab = a.Hy * a.sx;
    const int cps = (slab + W - 1) / W;                           // chunks per slab
    const long long per_field = (long long)cps * a.nlev;
    if (item >= per_field * a.nfields) return;
    const int f = (int)(item / per_field);
    const long long r = item - (long long)f * per_field;
    const int lev = (int)(r / cps);
    const int e0 = ((int)(r - (long long)lev * cps)) * W;         // first element of the chunk inside the slab
    T* field = static_cast<T*>(pt.ptr[f]) + a.plane * lev + (long long)a.sx * a.row0 + e0;
    T* msg = buffer + ((long long)f * a.nlev + lev) * slab + e0;
    T* dst = PACK ? msg : field;
    const T* src = PACK ? field : msg;
    if (e0 + W <= slab) {
        *reinterpret_cast<lvec_t*>(dst) = *reinterpret_cast<const lvec_t*>(src);
    } else {
        for (int v = 0; v < slab - e0; ++v) dst[v] = src[v];
    }
}

int check_fields(void* const fields[], int nfields)
{
    if (!fields || nfields < 1) { tpg::set_error("no fields"); return TPG_ERR_INVALID_ARGUMENT; }
    for (int f = 0; f < nfields; ++f)
        if (!fields[f]) { tpg::set_error("null field %d", f); return TPG_ERR_INVALID_ARGUMENT; }
    return TPG_OK;
}

#define TPG_LAUNCH(kernel, grid, block, stream, ...)                                                            \
    do {                                                                                                        \
        if (tpg::ev_start || tpg::ev_stop) {                                                                    \
            hipExtLaunchKernelGGL(kernel, grid, block, 0, stream, tpg::ev_start, tpg::ev_stop, 0, __VA_ARGS__); \
            tpg::ev_start = tpg::ev_stop = nullptr;                                                             \
        } else                                                                                                  \
            hipLaunchKernelGGL(kernel, grid, block, 0, stream, __VA_ARGS__);                                    \
    } while (0)

// Which instantiation of the chunked kernels serves this geometry and these pointers.  Plain (gen = false): Hx and Nx whole numbers of
// 16-B chunks and every field 16-B aligned -- the geometry of the defaults, halo (4, 4, 4).  GEN (see the note above) for everything else:
// an odd Hx (the reference's model halo (5, 5, 5)), 16-B-misaligned pointers, and -- with 8-B chunks, W = 2 -- Float32 rows with
// Nx = 2 mod 4.  Nx is even (tripolar_grid.jl:81-83), so W = 2 always divides it: every geometry has a chunked form.
struct ChunkPlan { int W; bool gen; };
template <typename T>
ChunkPlan chunk_plan(const Geom& g, void* const fields[], int n)
{
    constexpr int WMAX = 16 / (int)sizeof(T);
    bool plain = g.Hx % WMAX == 0 && g.Nx % WMAX == 0;
    for (int f = 0; f < n && plain; ++f) plain = ((uintptr_t)fields[f] % 16) == 0;
    if (plain) return { WMAX, false };
    return { g.Nx % WMAX == 0 ? WMAX : 2, true };
}

template <typename T, int W, bool COPY, bool GEN>
void launch_cols(int Hy, dim3 grid, hipStream_t s, const FieldTable& ft, const ZipArgs& a)
{
    switch (Hy) {
    case 1: TPG_LAUNCH((k_zipper_cols<T, W, 1, COPY, GEN>), grid, dim3(256), s, ft, a); break;
    case 2: TPG_LAUNCH((k_zipper_cols<T, W, 2, COPY, GEN>), grid, dim3(256), s, ft, a); break;
    case 3: TPG_LAUNCH((k_zipper_cols<T, W, 3, COPY, GEN>), grid, dim3(256), s, ft, a); break;
    case 4: TPG_LAUNCH((k_zipper_cols<T, W, 4, COPY, GEN>), grid, dim3(256), s, ft, a); break;
    case 5: TPG_LAUNCH((k_zipper_cols<T, W, 5, COPY, GEN>), grid, dim3(256), s, ft, a); break;
    case 6: TPG_LAUNCH((k_zipper_cols<T, W, 6, COPY, GEN>), grid, dim3(256), s, ft, a); break;
    case 7: TPG_LAUNCH((k_zipper_cols<T, W, 7, COPY, GEN>), grid, dim3(256), s, ft, a); break;
    default: TPG_LAUNCH((k_zipper_cols<T, W, 8, COPY, GEN>), grid, dim3(256), s, ft, a); break;
    }
}

// Kernel choice: column items (k_zipper_cols, plain or GEN: chunk_plan) for Hy <= 8; row items otherwise
// (k_zipper_vec for Hy > 8 -- e.g. the extended north halo of the split-explicit free surface -- on the plain geometry, k_zipper_scalar
// for Hy > 8 elsewhere and for Hy = 0).  TPG_ZIPPER_VARIANT=0 forces the row kernels everywhere (cross-check,
// tests/test_gpu_variants.py).  What was measured and dropped (tools/fillbench, profiles/r02/fillbench_ab.txt):
// plain loads (cold-dirty 26 vs 20 us), non-temporal stores (+2 us), write-through sc1 / sc0 sc1 buffer stores
// (-0.5 us cold-clean, +0 dirty), a persistent software-pipelined grid (1024 blocks, loads of item n+1 ahead of the
// stores of item n: -0.5 us), two half-row chunks per thread (one resident round of 4224 waves: +-0), 512 / 1024-thread
// blocks, two levels per thread (slower).  All of them, and same-shape pure copies, sit at 14.7-16.1 us cold:
// the 73 MB launch is at the copy ceiling of this access shape.
template <typename T, bool COPY = false>
int zipper_batch(void* const fields[], int n, const int8_t xloc[], const int8_t yloc[], const int32_t sign[],
                 const Geom& g, int kstart, int kcount, hipStream_t s)
{
    constexpr int WMAX = 16 / (int)sizeof(T);
    const ChunkPlan cp = chunk_plan<T>(g, fields, n);
    const int W = cp.W;
    const bool vec = !cp.gen;                                        // the row-item kernel k_zipper_vec exists in the plain form only
    const bool cols = g.Hy >= 1 && g.Hy <= 8 && (COPY ? vec : tpg::config().zipper_variant != 0);    // Hy = 0: only the row-Ny substitution remains (row kernels)
    if (COPY && !cols) { tpg::set_error("copy probe: geometry has no plain column kernel"); return TPG_ERR_UNSUPPORTED; }

    FieldTable ft;
    ZipArgs a;
    a.Nx = g.Nx; a.Ny = g.Ny; a.Hx = g.Hx; a.Hy = g.Hy; a.Hz = g.Hz; a.sx = g.sx; a.plane = g.plane;
    a.kstart = kstart; a.kcount = kcount;
    const bool chunked = cols || vec;
    a.nchunks = chunked ? g.Nx / W : g.Nx;
    a.fix0 = chunked ? (g.Nx / 2) / W : g.Nx / 2;      // first chunk / element (0-based) holding an i > Nx/2
    ft.nfields = n;
    long long total = 0;
    for (int f = 0; f < n; ++f) {
        ft.ptr[f] = fields[f]; ft.xloc[f] = xloc[f]; ft.yloc[f] = yloc[f]; ft.sign[f] = sign[f];
        ft.item0[f] = (int)total;
        long long per_level = cols ? a.nchunks
                                   : (long long)g.Hy * a.nchunks + (yloc[f] == TPG_CENTER ? a.nchunks - a.fix0 : 0);
        total += per_level * kcount;
        if (total >= (1ll << 31)) { tpg::set_error("zipper batch too large for 32-bit item index"); return TPG_ERR_UNSUPPORTED; }
    }
    ft.item0[n] = (int)total;
    if (total == 0) return TPG_OK;
    dim3 grid((unsigned)((total + 255) / 256));
    if (cols) {
        dim3 grid2((unsigned)(((long long)kcount * a.nchunks + 255) / 256), (unsigned)n);
        if constexpr (COPY) {
            launch_cols<T, WMAX, true, false>(g.Hy, grid2, s, ft, a);
        } else if constexpr (sizeof(T) == 8) {
            if (cp.gen) launch_cols<T, 2, false, true>(g.Hy, grid2, s, ft, a);
            else        launch_cols<T, 2, false, false>(g.Hy, grid2, s, ft, a);
        } else {
            if (W == 2)      launch_cols<T, 2, false, true>(g.Hy, grid2, s, ft, a);
            else if (cp.gen) launch_cols<T, 4, false, true>(g.Hy, grid2, s, ft, a);
            else             launch_cols<T, 4, false, false>(g.Hy, grid2, s, ft, a);
        }
    }
    else if (vec) TPG_LAUNCH((k_zipper_vec<T, WMAX>), grid, dim3(256), s, ft, a);
    else          TPG_LAUNCH((k_zipper_scalar<T>), grid, dim3(256), s, ft, a);
    return tpg::launch_status("k_zipper");
}

template <typename T, int W, bool GEN>
int merged_batch(const FieldTable& t, const MergedArgs& a, int n, int Hy, hipStream_t s)
{
    const long long itemsB = a.rowsB * a.hw;
    dim3 grid(a.blocksA + a.blocksS + (unsigned)((itemsB + 255) / 256), (unsigned)n);
    switch (Hy) {
    case 1: TPG_LAUNCH((k_fill_merged<T, W, 1, GEN>), grid, dim3(256), s, t, a); break;
    case 2: TPG_LAUNCH((k_fill_merged<T, W, 2, GEN>), grid, dim3(256), s, t, a); break;
    case 3: TPG_LAUNCH((k_fill_merged<T, W, 3, GEN>), grid, dim3(256), s, t, a); break;
    case 4: TPG_LAUNCH((k_fill_merged<T, W, 4, GEN>), grid, dim3(256), s, t, a); break;
    case 5: TPG_LAUNCH((k_fill_merged<T, W, 5, GEN>), grid, dim3(256), s, t, a); break;
    case 6: TPG_LAUNCH((k_fill_merged<T, W, 6, GEN>), grid, dim3(256), s, t, a); break;
    case 7: TPG_LAUNCH((k_fill_merged<T, W, 7, GEN>), grid, dim3(256), s, t, a); break;
    default: TPG_LAUNCH((k_fill_merged<T, W, 8, GEN>), grid, dim3(256), s, t, a); break;
    }
    return tpg::launch_status("k_fill_merged");
}

// the (T, W, GEN) instantiations that exist: Float64 16-B chunks plain / GEN; Float32 16-B chunks plain / GEN, 8-B chunks GEN only
template <typename T>
int merged_dispatch(const FieldTable& t, const MergedArgs& a, int n, int Hy, ChunkPlan cp, hipStream_t s)
{
    if constexpr (sizeof(T) == 8) return cp.gen ? merged_batch<T, 2, true>(t, a, n, Hy, s) : merged_batch<T, 2, false>(t, a, n, Hy, s);
    else if (cp.W == 2) return merged_batch<T, 2, true>(t, a, n, Hy, s);
    else return cp.gen ? merged_batch<T, 4, true>(t, a, n, Hy, s) : merged_batch<T, 4, false>(t, a, n, Hy, s);
}

template <typename T>
void fused_vec_dispatch(dim3 grid, hipStream_t s, const FieldTable& t, const FusedVecArgs& v, ChunkPlan cp)
{
    if constexpr (sizeof(T) == 8) {
        if (cp.gen) TPG_LAUNCH((k_fill_fused_vec<T, 2, true>), grid, dim3(256), s, t, v);
        else        TPG_LAUNCH((k_fill_fused_vec<T, 2, false>), grid, dim3(256), s, t, v);
    } else {
        if (cp.W == 2)   TPG_LAUNCH((k_fill_fused_vec<T, 2, true>), grid, dim3(256), s, t, v);
        else if (cp.gen) TPG_LAUNCH((k_fill_fused_vec<T, 4, true>), grid, dim3(256), s, t, v);
        else             TPG_LAUNCH((k_fill_fused_vec<T, 4, false>), grid, dim3(256), s, t, v);
    }
}

}  // namespace
