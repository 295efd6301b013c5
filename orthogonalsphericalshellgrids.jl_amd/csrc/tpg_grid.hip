// tpg_grid.hip -- TripolarGrid coordinate + staggered-metric precompute for gfx950.
//
// Replaces src/tripolar_grid.jl:73-328 of the reference (see include/tripolar_hip.h).  The
// reference runs ~40 full-array host passes (two CPU KernelAbstractions kernels, 8 circshifts,
// 20 Field set!/fill_halo_regions!/deepcopy round trips, 12 continue_south!, 20 map+H2D copies);
// here the final padded device arrays are produced directly by four launches on one stream:
//
//   K0 tables      : per-i  a*sind(lambda), a*cosd(lambda) (Face, Center; the Nlambda/4 circshift is
//                    folded into the index), per-j sinh(psi), cosh(psi) latitude-stretching tables
//                    (Face, Center), and the (Hy+1)-row lat-lon continuation table.  O(Nx+Ny) work.
//   K1 cells       : every interior cell (i, j) of the rank's row band: the Murray (1996) map at the 4
//                    staggered locations of the cell and of its stencil neighbours THROUGH the halo
//                    index maps (periodic x, zipper fold, row-Ny substitution, zero south halo),
//                    then the 8 haversine edge lengths, 2 spherical quadrilateral areas and 2
//                    product areas; stores 8 + 12 values.  Two forms with identical arithmetic
//                    (tests/test_gpu_variants.py), selected by TPG_CELLS_VARIANT:
//                      3  k_cells_tile   DEFAULT.  A block of 8 waves evaluates 8 point rows x 64
//                                        columns once (one point set per thread), parks them in LDS
//                                        and, after one barrier, every thread computes its cell from
//                                        its own registers + LDS neighbours.  <= 128 VGPRs: 4
//                                        waves/SIMD; transcendentals as straight-line batches
//                                        (tpg_batch.hpp).
//                      0  k_cells        one thread per cell, everything recomputed: the simple
//                                        reference form the tile kernel is checked against.
//                    (Round 1 also carried two register-marching forms -- waves of 62 columns marching
//                    north with the previous row in registers, 256 VGPRs, 2 waves/SIMD: 664-706 us vs the
//                    tile form's 499 us at 1/10 degree; removed, history in DESIGN.md 4.)
//   K2 halos       : compact pass over halo cells only (x-halo columns, north fold rows, zero
//                    south rows of the coordinates, row-Ny substitution of the y-Center metrics).
//                    Round 6 built the alternative -- the tile kernel PUSHES these cells itself (HaloPush below), K0 + K1
//                    is then the whole build -- and measured it against this sequence: it saves 0.3-2.8 us per build
//                    inside one binary and nothing against the round-5 binary (the 1/10 degree globe: + 1.7 %).  It is
//                    kept in the TEST library (TPG_CELLS_VARIANT=3, k_cells_tile_push) as a bit-identical variant; the
//                    product builds with K0 + K1 + K2.
//   K3 south       : lat-lon continuation rows j = 1-Hy..1 of the 12 metrics (south rank only).
//
// No inter-rank communication: seam halo rows are the neighbour's interior rows of the same
// analytic function.
#include "tpg_common.hpp"
#include "tpg_math.hpp"
#include "tpg_batch.hpp"

using namespace tpgm;

namespace {

constexpr double kC180Pi = 180.0 / kPi;
constexpr double kC360Pi = 360.0 / kPi;

struct GridK {
    int Nx, Ny, Hx, Hy;
    int shift;            // Nx / 4
    int jstart, jend;     // owned global rows
    int jm_lo, jm_hi;     // global rows whose interior cells this rank evaluates
    int sx, rows;         // local padded extents
    int ft;
    double fplp90;        // first_pole_longitude + 90
    double R;
    // workspace tables
    const double* ti;     // [4][Nx]: a*sind(lf), a*cosd(lf), a*sind(lc), a*cosd(lc), post-shift index
    const double* tj;     // [4][Ny]: sinh/cosh(psi) at Face rows, then at Center rows
    const double* ts;     // [5][Hy+1]: dx_c, dx_f, az_c, az_f, dy for j = 1-Hy..1
};

struct OutPtrs { void* p[TPG_NUM_ARRAYS]; };

struct TableArgs {
    int Nx, Ny, Hy, shift, ft;
    double south, npl, R;
    double* ti; double* tj; double* ts;
};

// ---- 1-D tables (src/tripolar_grid.jl:76-97) ------------------------------------------------
__device__ double lambda_face(int i, int N, int ft)
{
    long long num = 360ll * (i - 1) - 180ll * N;
    if (ft == TPG_F32) return (double)((float)num / (float)N);
    return (double)num / (double)N;
}
__device__ double lambda_center(int i, int N, int ft)
{
    long long num = 360ll * (2ll * i - 1) - 360ll * N;
    if (ft == TPG_F32) return (double)((float)num / (float)(2ll * N));
    return (double)num / (double)(2ll * N);
}
__device__ double phi_center(int j, int N, double south)
{
    if (N == 1) return south;
    if (south == __builtin_rint(south) && absD(south) < 1e6) {
        long long s = (long long)south;
        long long num = s * (N - 1) + (90 - s) * (long long)(j - 1);
        return (double)num / (double)(N - 1);
    }
    dd step = dd_div(two_sum(90.0, -south), dd{ (double)(N - 1), 0.0 });
    dd v = dd_add(dd{ south, 0.0 }, dd_mul_d(step, (double)(j - 1)));
    return v.hi + v.lo;
}

__global__ __launch_bounds__(64) void k_tables(TableArgs t)
{
    int tid = blockIdx.x * blockDim.x + threadIdx.x;
    const double a = tand((90.0 - t.npl) / 2);               // focal_distance, :76
    if (tid < t.Nx) {
        int is = tid + 1;
        int i0 = is - t.shift; if (i0 < 1) i0 += t.Nx;        // circshift folded into the index (:121-130)
        double lf = lambda_face(i0, t.Nx, t.ft), lc = lambda_center(i0, t.Nx, t.ft);
        double slf = sind(lf), clf = cosd(lf), slc = sind(lc), clc = cosd(lc);
        if (t.ft == TPG_F32) {   // sind(::Float32) returns Float32 in the reference
            slf = (double)(float)slf; clf = (double)(float)clf; slc = (double)(float)slc; clc = (double)(float)clc;
        }
        t.ti[0 * t.Nx + tid] = a * slf;
        t.ti[1 * t.Nx + tid] = a * clf;
        t.ti[2 * t.Nx + tid] = a * slc;
        t.ti[3 * t.Nx + tid] = a * clc;
        return;
    }
    tid -= t.Nx;
    if (tid < 2 * t.Ny) {
        int face = tid < t.Ny;
        int j = (face ? tid : tid - t.Ny) + 1;
        double pc = phi_center(j, t.Ny, t.south);
        double phi = pc;
        if (face) {
            double dphi = t.Ny > 1 ? phi_center(2, t.Ny, t.south) - phi_center(1, t.Ny, t.south) : 0.0;   // :96
            phi = pc - dphi / 2;                                                                          // :97
        }
        double psi = asinhD(tand((90.0 - phi) / 2) / a);      // generate_tripolar_coordinates.jl:66
        double sh, ch;
        sinh_cosh(psi, sh, ch);
        int base = face ? 0 : 2;
        t.tj[(base + 0) * t.Ny + (j - 1)] = sh;
        t.tj[(base + 1) * t.Ny + (j - 1)] = ch;
        return;
    }
    tid -= 2 * t.Ny;
    if (tid <= t.Hy) {
        // lat-lon continuation rows (src/tripolar_grid.jl:277-300); Oceananigans
        // LatitudeLongitudeGrid metric formulas, SURVEY.md Appendix A-8
        int j = 1 - t.Hy + tid;
        const double south = t.south, R = t.R;
        const double dlam = 360.0 / (double)t.Nx;
        const double dphiL = (90.0 - south) / (double)t.Ny;
        double pf, pfn, pc, pcm;
        if (south == __builtin_rint(south)) {
            long long s = (long long)south, Ny = t.Ny;
            pf  = (double)(s * Ny + (90 - s) * (long long)(j - 1)) / (double)Ny;
            pfn = (double)(s * Ny + (90 - s) * (long long)j) / (double)Ny;
            pc  = (double)(2 * s * Ny + (90 - s) * (long long)(2 * j - 1)) / (double)(2 * Ny);
            pcm = (double)(2 * s * Ny + (90 - s) * (long long)(2 * j - 3)) / (double)(2 * Ny);
        } else {
            pf  = south + (double)(j - 1) * dphiL;
            pfn = south + (double)j * dphiL;
            pc  = south + ((double)(2 * j - 1) * (90.0 - south)) / (double)(2 * t.Ny);
            pcm = south + ((double)(2 * j - 3) * (90.0 - south)) / (double)(2 * t.Ny);
        }
        int n = t.Hy + 1;
        t.ts[0 * n + tid] = R * (dlam * kDeg2Rad) * cosD(kPi * pc / 180);
        t.ts[1 * n + tid] = R * (dlam * kDeg2Rad) * cosD(kPi * pf / 180);
        t.ts[2 * n + tid] = R * R * (dlam * kDeg2Rad) * (sinD(kPi * pfn / 180) - sinD(kPi * pf / 180));
        t.ts[3 * n + tid] = R * R * (dlam * kDeg2Rad) * (sinD(kPi * pc / 180) - sinD(kPi * pcm / 180));
        t.ts[4 * n + tid] = R * (dphiL * kDeg2Rad);
    }
}

// ---- the Murray (1996) map at one staggered point (generate_tripolar_coordinates.jl:66-87) ----
struct LP { double lam, phi; };

__device__ __forceinline__ LP analytic(const GridK& g, int xl, int yl, int is, int js)
{
    int i0 = is - g.shift; if (i0 < 1) i0 += g.Nx;           // pre-shift index drives the :75/:82 branches
    const double* ti = g.ti + (xl == TPG_FACE ? 0 : 2) * g.Nx;
    const double* tj = g.tj + (yl == TPG_FACE ? 0 : 2) * g.Ny;
    double asl = ti[is - 1], acl = ti[g.Nx + is - 1];
    double sh = tj[js - 1], ch = tj[g.Ny + js - 1];
    double x = asl * ch;                                     // :67
    double y = acl * sh;                                     // :68
    bool pole = (x == 0.0) & (y == 0.0);                     // :74
    double l = pole ? (i0 == 1 ? -90.0 : 90.0) : -kC180Pi * atanD(y / x);   // :75-77
    LP r;
    r.phi = 90.0 - kC360Pi * atanD(sqrt(y * y + x * x));     // :78
    l += (i0 <= g.Nx / 2) ? -90.0 : 90.0;                    // :82
    l += g.fplp90;                                           // :86
    r.lam = fmod360(fmod360(l) + 360.0);                     // :87, OrthogonalSphericalShellGrids.jl:24
    return r;
}

// x index of the fold partner (zipper_boundary_condition.jl:73-75, :90-92, :110, :125)
__device__ __forceinline__ int fold_partner(int xl, int i, int Nx)
{
    int ip = (xl == TPG_FACE) ? Nx - i + 2 : Nx - i + 1;
    return ip > Nx ? ip - Nx : ip;
}

// Value of the halo-filled coordinate field of location (xl,yl) at logical (i,j), i in 0..Nx+1,
// j in 0..Ny+1: the composition of periodic x, north fold (sign +1, src/tripolar_grid.jl:147),
// row-Ny substitution (zipper_boundary_condition.jl:102,135) and the zero south halo (:148).
__device__ __forceinline__ LP coord(const GridK& g, int xl, int yl, int i, int j)
{
    if (j < 1) return LP{ 0.0, 0.0 };
    int iw = i < 1 ? i + g.Nx : (i > g.Nx ? i - g.Nx : i);
    int js = j;
    if (j > g.Ny) {
        int dj = j - g.Ny;
        js = (yl == TPG_FACE) ? g.Ny - dj + 1 : g.Ny - dj;
        iw = fold_partner(xl, iw, g.Nx);
        if (js < 1) return LP{ 0.0, 0.0 };
    } else if (j == g.Ny && yl == TPG_CENTER && iw > g.Nx / 2) {
        iw = fold_partner(xl, iw, g.Nx);
    }
    return analytic(g, xl, yl, iw, js);
}

// Distances.haversine((l1,p1),(l2,p2),R), call sites src/tripolar_grid_utils.jl:13-21
__device__ __forceinline__ double haversine(LP a, LP b, double R)
{
    double dl = (b.lam - a.lam) * kDeg2Rad;
    double a1 = a.phi * kDeg2Rad;
    double a2 = b.phi * kDeg2Rad;
    double dp = a2 - a1;
    double s1 = sinD(dp / 2), s2 = sinD(dl / 2);
    double h = s1 * s1 + cosD(a1) * cosD(a2) * (s2 * s2);
    double r = sqrt(h);
    return 2 * (R * asinD(r != r ? r : (r < 1.0 ? r : 1.0)));
}

struct V3 { double x, y, z; };
// Oceananigans lat_lon_to_cartesian(phi, lambda, 1)
__device__ __forceinline__ V3 cartesian(LP p)
{
    double cl = cosd(p.phi);
    return V3{ cosd(p.lam) * cl, sind(p.lam) * cl, sind(p.phi) };
}
__device__ __forceinline__ double dot3(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 cross3(V3 a, V3 b)
{
    return V3{ a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x };
}
// Oceananigans spherical_area_triangle / spherical_area_quadrilateral
__device__ __forceinline__ double tri_area(V3 a, V3 b, V3 c)
{
    double t = absD(dot3(a, cross3(b, c)));
    t /= 1 + dot3(a, b) + dot3(b, c) + dot3(a, c);
    return 2 * atanD(t);
}
__device__ __forceinline__ double quad_area(V3 a, V3 b, V3 c, V3 d)
{
    double A = tri_area(a, b, c);
    A += tri_area(a, b, d);
    A += tri_area(a, c, d);
    A += tri_area(b, c, d);
    return A / 2;
}

// tan(E/2) of one spherical triangle through the unscaled division (tpgm::div_nr).  The denominator
// 1 + a.b + b.c + a.c of unit vectors is 0 or at least ~1e-16 in magnitude and at most 4; a zero gives
// NaN here (rcp(0) = inf), which the callers' atan_small_b reports as `rare`, and their fallback
// recomputes the four tangents with IEEE `/` (tri_tan).
__device__ __forceinline__ double tri_tan_nr(V3 a, V3 b, V3 c)
{
    return div_nr(absD(dot3(a, cross3(b, c))), 1 + dot3(a, b) + dot3(b, c) + dot3(a, c));
}
__device__ __forceinline__ double tri_tan(V3 a, V3 b, V3 c)
{
    return absD(dot3(a, cross3(b, c))) / (1 + dot3(a, b) + dot3(b, c) + dot3(a, c));
}

// NT = streaming store: the 20 output arrays (1 GB at 1/10 deg) are written once and not re-read by
// this launch sequence; a plain store would park them as dirty lines in L2 / Infinity Cache and
// the NEXT kernel on the stream (typically a halo fill) would pay for their eviction.
template <typename T, bool NT = false>
__device__ __forceinline__ void put(const OutPtrs& o, int q, long long off, double v)
{
    T* p = static_cast<T*>(o.p[q]) + off;
    if (NT) __builtin_nontemporal_store((T)v, p); else *p = (T)v;
}

// The tile kernel addresses its stores as (uniform base pointer) + (32-bit BYTE offset in a VGPR): the saddr form of
// global_store takes exactly that, so the 20 stores of a cell share two offset registers instead of each paying a 64-bit
// v_lshl_add_u64 for its own address (round 3).  Valid while one array is smaller than 4 GiB -- the launcher checks.
template <typename T, bool NT = false>
__device__ __forceinline__ void put32(const OutPtrs& o, int q, unsigned boff, double v)
{
    // the empty asm keeps the zero-extension of the offset next to its store: hoisted and shared as a 64-bit pair, it would no
    // longer match the saddr addressing pattern and every store would pay a 64-bit add again
    asm volatile("" : "+v"(boff));
    T* p = reinterpret_cast<T*>(static_cast<char*>(o.p[q]) + boff);
    if (NT) __builtin_nontemporal_store((T)v, p); else *p = (T)v;
}

// ---- halo images pushed by the producer (round 6) ------------------------------------------------------------------------
// K2 (k_halos) copies every halo cell of the 20 arrays from the interior cell that defines it: a second launch (6.8 us + a kernel
// boundary: 3 % of the 1/10 degree build, 7-9 % of a 1/4 degree build or of a 225-row band of BASELINE config 4).  The same cells can
// be written by the thread that PRODUCES the value, because every one of K2's index maps is a bijection from sources to destinations:
//   periodic x        the x-halo columns i = 1-Hx .. 0 and Nx+1 .. Nx+Hx are simply CELLS of the tile grid (it spans 1-Hx .. Nx+Hx): the
//                     wrapped column feeds the same arithmetic, so the same bits come out, and the stores stay whole coalesced rows.
//                     (Pushing them as images from the first / last interior columns cost what K2 saved: + 3.7 us of partial-wave
//                     stores in the two edge tile columns; profiles/r06/build_push_ab.txt)                          [src/tripolar_grid.jl:149-150]
//   north fold, +1    (i, j) with dj = Ny - j (+1 for y-Face arrays) in 1 .. rows present: at (i*, Ny + dj), i* = Nx - i + 1 (x-Center) or
//                     Nx - i + 2 (x-Face; i = 1 -> 1), and at the periodic images of THAT cell (the corners)       [:147; zipper_boundary_condition.jl:70-138]
//   row-Ny substitution of the y-Center METRICS: (i, Ny) with partner i' = Nx - i + 1 / Nx - i + 2 in Nx/2+1 .. Nx also lands at (i', Ny) and
//                     its west image; the thread that owns (i', Ny) does not store there                            [zipper_boundary_condition.jl:102,135]
//                     (the coordinates were evaluated through the substitution already: coord())
//   south             rows j < 1 of the coordinates are 0.0 (src/tripolar_grid.jl:148), rows j <= 1 of the metrics the lat-lon
//                     continuation (:277-300, continue_south!): constants per row, written for their 62 columns (+ x-halo images) by the
//                     apron wave of the southernmost tiles; the cells of row 1 do not store their metrics.
// Only EDGE tiles do any of this (block-uniform test): the tile rows that reach row Ny - Hy and the southernmost tile row.  Valid where source and images are distinct cells -- Ny > 2 Hy + 2 and Nx >= 2 Hx + 2; other grids keep K2 / K3.
struct HaloPush {
    int on;       // 1: the tile kernel writes every halo cell itself
    int jn_hi;    // last north halo row present in the band: min(jend + Hy, Ny + Hy); Ny = none
    int south;    // 1: the band holds rows j <= 1 whose metrics are the lat-lon continuation (south_in_band)
    int j_lo;     // first row of the band's parent: jstart - Hy
};

// value v of array q (location xl, yl; metric or coordinate; all four compile-time constants at the call sites) produced at the interior cell
// (i, j), whose own cell is at byte offset boff: the own cell and every image of it, as saddr stores at offsets derived from boff.  The images
// carry the SAME streaming hint as the own-cell stores: where the compiler sinks the stores of the edge and the non-edge branch of a site
// into one instruction it keeps a hint only if both had it, and a first version with plain image stores silently turned 11 of the 21 stores
// of the NON-edge path into plain ones -- the 1 GB then stayed behind as dirty lines and the next kernel on the stream (the halo fill of a
// bench step) ran 67 instead of 47 us (profiles/r06/build_push_ab.txt; tests/test_abi.py now counts the plain stores of the kernel).  Inlined:
// a few integer operations and predicated stores per site, in edge tiles only (a first version behind a noinline call, with g / o copied to
// scratch for it, ran the 1/4 degree build at 344 us instead of 102: profiles/r06/build_push_ab.txt).
template <typename T, bool NT>
__device__ __forceinline__ void emit_edge(const GridK& g, const HaloPush& hp, const OutPtrs& o, int q, int xl, int yl, bool metric, int i, int j,
                                          unsigned boff, double v)
{
    constexpr int sz = (int)sizeof(T);
    const int Nx = g.Nx, Ny = g.Ny, Hx = g.Hx;
    const int iw = i < 1 ? i + Nx : (i > Nx ? i - Nx : i);                   // an x-halo column is a cell of its own: same value as column iw
    const int ipart = Nx - iw + 1 + xl;                                      // un-wrapped fold partner (x-Face iw = 1: Nx + 1)
    const unsigned pw = (unsigned)(Nx * sz);                                 // byte distance of a periodic image
    const bool rowNyC = metric && yl == TPG_CENTER && j == Ny;
    const bool dest = rowNyC && iw > Nx / 2 && ipart != iw;                   // a substitution destination (or its west image): the partner writes it
    const bool south_row = metric && hp.south && j <= 1;                      // a continuation row: the south tiles' apron waves write it
    if (!dest && !south_row) put32<T, NT>(o, q, boff, v);
    if (i < 1 || i > Nx) return;                                              // images are pushed from the interior instance of a column only
    if (rowNyC && ipart > Nx / 2 && ipart <= Nx && ipart != i) {
        const unsigned b2 = boff + (unsigned)((ipart - i) * sz);
        put32<T, NT>(o, q, b2, v);
        if (ipart > Nx - Hx) put32<T, NT>(o, q, b2 - pw, v);
    }
    const int dj = Ny - j + yl;                                              // fold image row Ny + dj: (Ny + dj) - j = 2 dj - yl rows up
    if (dj >= 1 && Ny + dj <= hp.jn_hi) {
        const int iwd = ipart > Nx ? ipart - Nx : ipart;
        const unsigned b3 = boff + (unsigned)(((iwd - i) + g.sx * (2 * dj - yl)) * sz);
        put32<T, NT>(o, q, b3, v);
        if (iwd <= Hx) put32<T, NT>(o, q, b3 + pw, v);
        if (iwd > Nx - Hx) put32<T, NT>(o, q, b3 - pw, v);
    }
}

// the south rows of column i (interior or x halo): zeros below row 1 of the 8 coordinate arrays, the continuation rows j <= 1 of the 12 metrics;
// boff1 = byte offset of (i, 1)
template <typename T, bool NT>
__device__ __forceinline__ void emit_south(const GridK& g, const HaloPush& hp, const OutPtrs& o, unsigned boff1)
{
    constexpr int sz = (int)sizeof(T);
    const int n = g.Hy + 1;
    const unsigned rw = (unsigned)(g.sx * sz);
#pragma unroll 1
    for (int j = hp.j_lo; j <= 1; ++j) {
        const int rt = j - (1 - g.Hy);
        const unsigned b = boff1 - rw * (unsigned)(1 - j);
        double tv[5];                                                       // the row's five continuation values, loaded before the first store
#pragma unroll
        for (int c = 0; c < 5; ++c) tv[c] = g.ts[c * n + rt];
#pragma unroll
        for (int q = 0; q < TPG_NUM_ARRAYS; ++q) {
            double v = 0.0;
            if (q >= TPG_DX_CC) {
                const int yl = (q >= TPG_DY_CC && q <= TPG_DY_FF) ? (q & 1) : ((q & 3) >> 1);          // array_loc(): dy is cc, cf, fc, ff
                const int col = (q >= TPG_DY_CC && q <= TPG_DY_FF) ? 4 : ((q >= TPG_AZ_CC ? 2 : 0) + (yl == TPG_FACE ? 1 : 0));
                v = tv[col];
            } else if (j >= 1) {
                continue;                                                   // row 1 of the coordinates is the cells'
            }
            put32<T, NT>(o, q, b, v);
        }
    }
}

// ---- K1: interior cells ------------------------------------------------------------------------
template <typename T, bool NT>
__global__ __launch_bounds__(256) void k_cells(GridK g, OutPtrs o)
{
    const int nbx = (g.Nx + 255) / 256;
    int rb = blockIdx.x / nbx;
    int i = (blockIdx.x - rb * nbx) * 256 + threadIdx.x + 1;
    int j = g.jm_lo + rb;
    if (i > g.Nx) return;

    LP cc = coord(g, 0, 0, i, j), fc = coord(g, 1, 0, i, j), cf = coord(g, 0, 1, i, j), ff = coord(g, 1, 1, i, j);
    LP fc_e = coord(g, 1, 0, i + 1, j), fc_s = coord(g, 1, 0, i, j - 1);
    LP cc_w = coord(g, 0, 0, i - 1, j), cc_s = coord(g, 0, 0, i, j - 1), cc_sw = coord(g, 0, 0, i - 1, j - 1);
    LP ff_e = coord(g, 1, 1, i + 1, j), ff_n = coord(g, 1, 1, i, j + 1), ff_ne = coord(g, 1, 1, i + 1, j + 1);
    LP cf_w = coord(g, 0, 1, i - 1, j), cf_n = coord(g, 0, 1, i, j + 1);

    const double R = g.R;
    double dxcc = haversine(fc_e, fc, R);      // tripolar_grid_utils.jl:13
    double dxfc = haversine(cc, cc_w, R);      // :14
    double dxcf = haversine(ff_e, ff, R);      // :15
    double dxff = haversine(cf, cf_w, R);      // :16
    double dycc = haversine(cf_n, cf, R);      // :18
    double dyfc = haversine(ff_n, ff, R);      // :19
    double dycf = haversine(cc, cc_s, R);      // :20
    double dyff = haversine(fc, fc_s, R);      // :21
    double azcc = quad_area(cartesian(ff), cartesian(ff_e), cartesian(ff_ne), cartesian(ff_n)) * (R * R);   // :23-28
    double azfc = dyfc * dxfc;                 // :34
    double azcf = dycf * dxcf;                 // :35
    double azff = quad_area(cartesian(cc_sw), cartesian(cc_s), cartesian(cc), cartesian(cc_w)) * (R * R);   // :38-43

    long long off = (long long)(i + g.Hx - 1) + (long long)g.sx * (j - g.jstart + g.Hy);
    put<T, NT>(o, TPG_LAMBDA_CC, off, cc.lam); put<T, NT>(o, TPG_LAMBDA_FC, off, fc.lam);
    put<T, NT>(o, TPG_LAMBDA_CF, off, cf.lam); put<T, NT>(o, TPG_LAMBDA_FF, off, ff.lam);
    put<T, NT>(o, TPG_PHI_CC, off, cc.phi); put<T, NT>(o, TPG_PHI_FC, off, fc.phi);
    put<T, NT>(o, TPG_PHI_CF, off, cf.phi); put<T, NT>(o, TPG_PHI_FF, off, ff.phi);
    put<T, NT>(o, TPG_DX_CC, off, dxcc); put<T, NT>(o, TPG_DX_FC, off, dxfc);
    put<T, NT>(o, TPG_DX_CF, off, dxcf); put<T, NT>(o, TPG_DX_FF, off, dxff);
    put<T, NT>(o, TPG_DY_CC, off, dycc); put<T, NT>(o, TPG_DY_CF, off, dycf);
    put<T, NT>(o, TPG_DY_FC, off, dyfc); put<T, NT>(o, TPG_DY_FF, off, dyff);
    put<T, NT>(o, TPG_AZ_CC, off, azcc); put<T, NT>(o, TPG_AZ_FC, off, azfc);
    put<T, NT>(o, TPG_AZ_CF, off, azcf); put<T, NT>(o, TPG_AZ_FF, off, azff);
}

// ---- shared point records of the work-sharing form (k_cells_tile) ------------------------------------
// Every staggered point (lambda, phi) and what the metrics derive from it -- a = deg2rad(phi),
// cos(a) for the haversines, the unit vector for the two quadrilateral areas -- is evaluated ONCE
// by the thread that owns it and shared with the cells around it: ~65 instead of ~140 transcendental
// calls per cell.  The arithmetic of every value is unchanged (same operation sequence), so results
// are bit-identical to k_cells.
struct Pt  { double lam, phi, a, ca; };                 // FC / CF points
struct PtX { double lam, phi, a, ca, X, Y, Z; };        // CC / FF points (+ unit vector)
struct Nb  { double lam, a, ca; };                      // what a haversine needs of a neighbour
struct NbX { double lam, a, ca, X, Y, Z; };

__device__ __forceinline__ Pt make_pt(const GridK& g, int xl, int yl, int i, int j)
{
    LP p = coord(g, xl, yl, i, j);
    Pt r; r.lam = p.lam; r.phi = p.phi; r.a = p.phi * kDeg2Rad; r.ca = cosD(r.a);
    return r;
}
__device__ __forceinline__ PtX make_ptx(const GridK& g, int xl, int yl, int i, int j)
{
    LP p = coord(g, xl, yl, i, j);
    PtX r; r.lam = p.lam; r.phi = p.phi; r.a = p.phi * kDeg2Rad; r.ca = cosD(r.a);
    double sl, cl, sp, cp;
    sincosd(p.lam, sl, cl);
    sincosd(p.phi, sp, cp);
    r.X = cl * cp; r.Y = sl * cp; r.Z = sp;             // lat_lon_to_cartesian(phi, lambda, 1)
    return r;
}
__device__ __forceinline__ Nb nb_of(const Pt& p) { return Nb{ p.lam, p.a, p.ca }; }
__device__ __forceinline__ Nb nb_of(const PtX& p) { return Nb{ p.lam, p.a, p.ca }; }
__device__ __forceinline__ NbX nbx_of(const PtX& p) { return NbX{ p.lam, p.a, p.ca, p.X, p.Y, p.Z }; }
__device__ __forceinline__ V3 v3_of(const PtX& p) { return V3{ p.X, p.Y, p.Z }; }
__device__ __forceinline__ V3 v3_of(const NbX& p) { return V3{ p.X, p.Y, p.Z }; }

template <int DIR> __device__ __forceinline__ double shf(double v)
{
    return DIR > 0 ? __shfl_down(v, 1, 64) : __shfl_up(v, 1, 64);   // DIR>0: value of lane+1
}
template <int DIR> __device__ __forceinline__ Nb shf_nb(const Pt& p)
{
    return Nb{ shf<DIR>(p.lam), shf<DIR>(p.a), shf<DIR>(p.ca) };
}
template <int DIR> __device__ __forceinline__ NbX shf_nbx(const PtX& p)
{
    return NbX{ shf<DIR>(p.lam), shf<DIR>(p.a), shf<DIR>(p.ca), shf<DIR>(p.X), shf<DIR>(p.Y), shf<DIR>(p.Z) };
}

// haversine(x, y, R) with the per-point parts precomputed: x = first argument, y = second
__device__ __forceinline__ double hav(double xlam, double xa, double xca, double ylam, double ya, double yca, double R)
{
    double dl = (ylam - xlam) * kDeg2Rad;
    double dp = ya - xa;
    double s1 = sinD(dp / 2), s2 = sinD(dl / 2);
    double h = s1 * s1 + xca * yca * (s2 * s2);
    double r = sqrt(h);
    return 2 * (R * asinD(r != r ? r : (r < 1.0 ? r : 1.0)));
}
#define HAV(P, Q) hav((P).lam, (P).a, (P).ca, (Q).lam, (Q).a, (Q).ca, R)


// ---- one point set = the 4 points a tile thread creates, on the straight-line batch forms ----------
//  * rows j < Ny need no fold / substitution / pole logic, so a lane's lambda-table values
//    (a sind, a cosd at its Face and Center column) and the per-row psi-table values (wave-uniform)
//    feed the Murray map directly; the index maps of coord() are left to the special rows;
//  * all transcendentals are evaluated by the straight-line batch forms of tpg_batch.hpp
//    (independent chains per basic block), with rare cases patched afterwards.
struct Step4 {           // the 4 points a step creates: k = 0 FC(jc), 1 CC(jc), 2 FF(jf), 3 CF(jf)
    double lam[4], phi[4], a[4], ca[4];
    double X[2], Y[2], Z[2];     // 0: CC, 1: FF
};

struct LaneConst { double aslF, aclF, aslC, aclC, hemi; };

struct RowTab { double shC, chC, shF, chF; };    // sinh/cosh(psi) of Center row jc and Face row jf: wave-uniform

// Scalar (s_load) read of the psi tables: the row index is wave-uniform and the tables were written by
// the previous kernel, so they can be read through the constant address space into SGPRs -- no VGPRs,
// and issued one row ahead so the latency never sits on the critical path.
typedef const double __attribute__((address_space(4))) kconst_double;
__device__ __forceinline__ RowTab load_rowtab(const GridK& g, int jc, int jf)
{
    kconst_double* tj = (kconst_double*)g.tj;
    jc = __builtin_amdgcn_readfirstlane(jc < 1 ? 1 : (jc > g.Ny ? g.Ny : jc));
    jf = __builtin_amdgcn_readfirstlane(jf < 1 ? 1 : (jf > g.Ny ? g.Ny : jf));
    return RowTab{ tj[2 * g.Ny + jc - 1], tj[3 * g.Ny + jc - 1], tj[0 * g.Ny + jf - 1], tj[1 * g.Ny + jf - 1] };
}

__device__ __forceinline__ void points_fast(const GridK& g, const LaneConst& lc, const RowTab& rt, Step4& s, const double* atab)
{
    const double shC = rt.shC, chC = rt.chC, shF = rt.shF, chF = rt.chF;
    double x[4] = { lc.aslF * chC, lc.aslC * chC, lc.aslF * chF, lc.aslC * chF };          // :67
    double y[4] = { lc.aclF * shC, lc.aclC * shC, lc.aclF * shF, lc.aclC * shF };          // :68
    double q[4], rr[4], at1[4], at2[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { q[k] = y[k] / x[k]; rr[k] = sqrt_nr<true>(y[k] * y[k] + x[k] * x[k]);   /* > 0: no pole below row Ny */ }
    tpgb::atan_tab_b<4>(q, at1, atab);
    tpgb::atan_tab_b<4, true>(rr, at2, atab);      // rr = sqrt(...) of finite table products
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        double l = -kC180Pi * at1[k];                          // :77 (no pole below row Ny)
        s.phi[k] = 90.0 - kC360Pi * at2[k];                    // :78
        l += lc.hemi;                                          // :82
        l += g.fplp90;                                         // :86
        s.lam[k] = tpgb::fmod360_pos(tpgb::fmod360_small(l) + 360.0);     // :87
        s.a[k] = s.phi[k] * kDeg2Rad;
    }
    if (tpgb::cos_b<4>(s.a, s.ca)) {
#pragma unroll
        for (int k = 0; k < 4; ++k) s.ca[k] = cosD(s.a[k]);
    }
    double ang[4] = { s.lam[1], s.phi[1], s.lam[2], s.phi[2] }, sn[4], cs[4];
    tpgb::sincosd_b<4>(ang, sn, cs);
    s.X[0] = cs[0] * cs[1]; s.Y[0] = sn[0] * cs[1]; s.Z[0] = sn[1];      // CC
    s.X[1] = cs[2] * cs[3]; s.Y[1] = sn[2] * cs[3]; s.Z[1] = sn[3];      // FF
}

// same arithmetic in batches of 2 (for kernels that run 4 waves/SIMD and must stay under 128 VGPRs)
__device__ __forceinline__ void points_fast2(const GridK& g, const LaneConst& lc, const RowTab& rt, Step4& s, const double* atab)
{
    const double shC = rt.shC, chC = rt.chC, shF = rt.shF, chF = rt.chF;
    double x[4] = { lc.aslF * chC, lc.aslC * chC, lc.aslF * chF, lc.aslC * chF };          // :67
    double y[4] = { lc.aclF * shC, lc.aclC * shC, lc.aclF * shF, lc.aclC * shF };          // :68
    double q[4], rr[4], at1[4], at2[4];
    // y / x through the unscaled division: both are products of finite table entries of moderate size (|x| is 0 on the two pole
    // meridians, else >= ~1e-17; tests/test_gpu_math.py checks div_nr against IEEE on that range).  A zero denominator is the one case
    // that needs IEEE semantics (y / +-0 = +-Inf, and atan(+-Inf) = +-pi/2): those lanes are patched after the table atan, which can
    // then take the FINITE form for every lane (no clamp of the argument; a NaN from div_nr(y, 0) just flows through the discarded lane).
    bool zerox = false;
#pragma unroll
    for (int k = 0; k < 4; ++k) { q[k] = div_nr(y[k], x[k]); zerox |= x[k] == 0.0; rr[k] = sqrt_nr<true>(y[k] * y[k] + x[k] * x[k]);   /* > 0: no pole below row Ny */ }
#pragma unroll
    for (int h = 0; h < 4; h += 2) {
        double qa[2] = { q[h], q[h + 1] }, ra[2] = { rr[h], rr[h + 1] }, o1[2], o2[2];
        tpgb::atan_tab_b<2, true>(qa, o1, atab);
        tpgb::atan_tab_b<2, true>(ra, o2, atab);      // ra = sqrt(...) of finite table products
        at1[h] = o1[0]; at1[h + 1] = o1[1]; at2[h] = o2[0]; at2[h + 1] = o2[1];
    }
    if (zerox) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (x[k] == 0.0) {
                // msun: atan(+-Inf) = +-(atanhi[3] + atanlo[3]) = +-pi/2 (the table row gives hi - ((-0 - lo) - t) = RN(hi + lo) = hi as well);
                // 0 / 0 stays NaN
                const double qq = y[k] / x[k];
                at1[k] = (qq != qq) ? qq : csign(kPio2Hi, qq);
            }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        double l = -kC180Pi * at1[k];                          // :77 (no pole below row Ny)
        s.phi[k] = 90.0 - kC360Pi * at2[k];                    // :78
        l += lc.hemi;                                          // :82
        l += g.fplp90;                                         // :86
        s.lam[k] = tpgb::fmod360_pos(tpgb::fmod360_small(l) + 360.0);     // :87
        s.a[k] = s.phi[k] * kDeg2Rad;
    }
    double sn[4], cs[4];
#pragma unroll
    for (int h = 0; h < 4; h += 2) {
        double aa[2] = { s.a[h], s.a[h + 1] }, cc2[2];
        if (tpgb::cos_lat_b<2>(aa, cc2)) { cc2[0] = cosD(aa[0]); cc2[1] = cosD(aa[1]); }      // a = deg2rad(latitude)
        s.ca[h] = cc2[0]; s.ca[h + 1] = cc2[1];
    }
    {   // unit vectors of CC (lam[1], phi[1]) and FF (lam[2], phi[2]): the two longitudes through the general form, the two latitudes
        // through the |x| <= 90 form; sn / cs index = 2 * point + (0 longitude, 1 latitude)
        double lon[2] = { s.lam[1], s.lam[2] }, lat[2] = { s.phi[1], s.phi[2] }, sl[2], cl[2], sp[2], cp[2];
        tpgb::sincosd_b<2>(lon, sl, cl);
        if (tpgb::sincosd_lat_b<2>(lat, sp, cp)) tpgb::sincosd_b<2>(lat, sp, cp);
        sn[0] = sl[0]; cs[0] = cl[0]; sn[1] = sp[0]; cs[1] = cp[0];
        sn[2] = sl[1]; cs[2] = cl[1]; sn[3] = sp[1]; cs[3] = cp[1];
    }
    s.X[0] = cs[0] * cs[1]; s.Y[0] = sn[0] * cs[1]; s.Z[0] = sn[1];      // CC
    s.X[1] = cs[2] * cs[3]; s.Y[1] = sn[2] * cs[3]; s.Z[1] = sn[3];      // FF
}

// general rows (row Ny: fold, substitution, pole; row 0: zero south halo) through coord()
__device__ __noinline__ void points_general(const GridK& g, int i, int jc, int jf, Step4& s)
{
    Pt fc = make_pt(g, 1, 0, i, jc), cf = make_pt(g, 0, 1, i, jf);
    PtX cc = make_ptx(g, 0, 0, i, jc), ff = make_ptx(g, 1, 1, i, jf);
    s.lam[0] = fc.lam; s.phi[0] = fc.phi; s.a[0] = fc.a; s.ca[0] = fc.ca;
    s.lam[1] = cc.lam; s.phi[1] = cc.phi; s.a[1] = cc.a; s.ca[1] = cc.ca;
    s.lam[2] = ff.lam; s.phi[2] = ff.phi; s.a[2] = ff.a; s.ca[2] = ff.ca;
    s.lam[3] = cf.lam; s.phi[3] = cf.phi; s.a[3] = cf.a; s.ca[3] = cf.ca;
    s.X[0] = cc.X; s.Y[0] = cc.Y; s.Z[0] = cc.Z;
    s.X[1] = ff.X; s.Y[1] = ff.Y; s.Z[1] = ff.Z;
}

// ---- K1 (tile form): 64 x R point sets per block through LDS, 4 waves per SIMD -------------------
// Sharing points through registers (a wave marching north with the previous row live) needs ~70 live doubles
// per lane: 2 waves/SIMD, where dependent FP64 chains leave the VALU idle ~28 % of the time (PMC, round 1); the
// plain thread-per-cell kernel at 4 waves/SIMD is 95 % busy.  This form keeps the work-sharing but not the
// register state: a block of 64 x R threads evaluates one point set per thread (step s = FC(s), CC(s),
// FF(s+1), CF(s+1)), parks {lambda, a, cos a[, X, Y, Z]} in LDS
// (18 doubles x 64 x R), and after ONE barrier every thread with a south and east/west neighbour
// inside the tile computes its cell from its own registers plus 48 LDS reads.  Row p = 0 and lanes
// 0 / 63 are aprons (tiles overlap by one point row / two columns): (R-1)/R x 62/64 of the lanes emit;
// the apron wave, which has no cell row, spends phase 2 on one of the eight haversines (Dy_ff) of all rows.
// A wave owns one point row, so the special rows (0, Ny) are a wave-uniform branch to coord().
// Same arithmetic as k_cells: bit-identical results.
template <int R> struct TileLds { double v[18][R][64]; };
enum { L_FC = 0, L_CC = 3, L_FF = 9, L_CF = 15 };    // field bases: FC(lam,a,ca) CC(lam,a,ca,X,Y,Z) FF(6) CF(3)

// N haversines at once: x = first argument, y = second, each {lambda, deg2rad(phi), cos}; the same
// operation sequence as hav() with the batch forms (rare arguments fall back to the scalar functions)
template <int N>
__device__ __forceinline__ void hav_batch(const Nb (&X)[N], const Nb (&Y)[N], double Rad, double (&d)[N])
{
    double hp[N], hl[N], s1[N], s2[N], hh[N], rt[N], rm[N], as[N];
    bool zero = false;
#pragma unroll
    for (int e = 0; e < N; ++e) {
        // (d lambda * deg2rad) / 2 as ONE product with deg2rad / 2: halving is exact and rounding is scale-invariant (no underflow:
        // a difference of two longitudes is 0 or >= ~1e-14), so RN(x c) / 2 = RN(x (c / 2)) bit for bit
        double dp = Y[e].a - X[e].a;
        hp[e] = dp / 2; hl[e] = (Y[e].lam - X[e].lam) * (kDeg2Rad * 0.5);
    }
    if (tpgb::sin_small_b<N>(hp, s1)) {
#pragma unroll
        for (int e = 0; e < N; ++e) s1[e] = sinD(hp[e]);
    }
    if (tpgb::sin_small_b<N>(hl, s2)) {
#pragma unroll
        for (int e = 0; e < N; ++e) s2[e] = sinD(hl[e]);
    }
#pragma unroll
    for (int e = 0; e < N; ++e) {
        double h = s1[e] * s1[e] + X[e].ca * Y[e].ca * (s2[e] * s2[e]);
        hh[e] = h;
        zero |= h == 0.0;
        rt[e] = sqrt_nr<true>(h);                 // h = 0 or >= ~1e-34 (squares of half-differences of O(1) doubles); h = 0 patched below
    }
    if (zero) {                                   // coincident points (pole-adjacent edges): the unscaled square root needs x > 0
#pragma unroll
        for (int e = 0; e < N; ++e) rt[e] = sqrt_nr(hh[e]);
    }
#pragma unroll
    for (int e = 0; e < N; ++e) {
        // min(r, 1) as ONE v_min_f64.  The reference's min keeps a NaN; minnum drops it (-> 1.0), but 1.0 is outside the fast domain
        // of asin_small_b (|x| < 0.5), so such a lane raises `rare` and the fallback below recomputes the NaN-keeping min from rt
        rm[e] = __builtin_fmin(rt[e], 1.0);
    }
    if (tpgb::asin_small_b<N>(rm, as)) {
#pragma unroll
        for (int e = 0; e < N; ++e) as[e] = asinD(!(rt[e] >= 1.0) ? rt[e] : 1.0);
    }
#pragma unroll
    for (int e = 0; e < N; ++e) d[e] = (2 * Rad) * as[e];      // 2 (R asin) = (2 R) asin bit for bit: doubling is exact (2 R: wave-uniform)
}

template <typename T, bool NT, int R>
__global__ __launch_bounds__(64 * R, 4) void k_cells_tile(GridK g, OutPtrs o, int tiles_x)
{
    __shared__ __attribute__((aligned(16))) double atabs[R][TPG_ATAN_TABLE_DOUBLES];
    __shared__ TileLds<R> lds;
    const int lane = threadIdx.x & 63;
    const int p = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);         // point row of this wave
    // one copy of the atan interval table per wave: its load overlaps the lane's other table loads and
    // needs no block barrier (LDS operations of one wave execute in order)
    double* atab = atabs[p];
    tpgb::atan_table_init(atab, lane);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // Tile rows are dispatched in blockIdx.y order: NORTH to SOUTH, so that the one slow row of a launch -- row Ny, whose wave takes the
    // general (scalar) path through coord(): +2.2 us of block latency -- starts first instead of ending the launch.  Worth ~0.3 % at
    // 1/10 degree (in-process A/B, round 4: 505.7 -> 504.2 us per build, i.e. inside the noise); kept because it costs nothing.
    const int ty = (int)gridDim.y - 1 - (int)blockIdx.y;
    const int tx = blockIdx.x;
    const int s0 = g.jm_lo - 1 + ty * (R - 1);
    const int s = s0 + p;                                                    // this wave's step
    int i = tx * 62 + lane;
    const bool col_emit = lane >= 1 && lane <= 62 && i <= g.Nx;
    if (i > g.Nx + 1) i = g.Nx + 1;
    const double Rad = g.R;
    const bool active_row = s <= g.jm_hi;                                    // rows past the band: idle waves
    const unsigned col = (unsigned)(i + g.Hx - 1);
    auto rowoff = [&](int j) -> unsigned { return (col + (unsigned)g.sx * (unsigned)(j - g.jstart + g.Hy)) * (unsigned)sizeof(T); };   // bytes

    // ---- phase 1: one point set per thread
    Step4 q;
    if (active_row) {
        const bool fast = s >= 1 && s < g.Ny && absD(g.fplp90) <= 360.0;     // wave-uniform
        if (fast) {
            LaneConst lc;
            const int iw = i < 1 ? i + g.Nx : (i > g.Nx ? i - g.Nx : i);
            int i0 = iw - g.shift; if (i0 < 1) i0 += g.Nx;
            lc.aslF = g.ti[0 * g.Nx + iw - 1]; lc.aclF = g.ti[1 * g.Nx + iw - 1];
            lc.aslC = g.ti[2 * g.Nx + iw - 1]; lc.aclC = g.ti[3 * g.Nx + iw - 1];
            lc.hemi = (i0 <= g.Nx / 2) ? -90.0 : 90.0;
            points_fast2(g, lc, load_rowtab(g, s, s + 1), q, atab);
        } else {
            Step4 tmp; GridK gc = g; points_general(gc, i, s, s + 1, tmp); q = tmp;
        }
        // coordinates: rows p >= 1 emit CC/FC(s) and FF/CF(s+1); the first tile's apron row emits FF/CF(jm_lo)
        if (col_emit) {
            if (p >= 1 && s >= g.jm_lo) {
                unsigned off = rowoff(s);
                put32<T, NT>(o, TPG_LAMBDA_FC, off, q.lam[0]); put32<T, NT>(o, TPG_PHI_FC, off, q.phi[0]);
                put32<T, NT>(o, TPG_LAMBDA_CC, off, q.lam[1]); put32<T, NT>(o, TPG_PHI_CC, off, q.phi[1]);
            }
            if ((p >= 1 || ty == 0) && s + 1 >= g.jm_lo && s + 1 <= g.jm_hi) {
                unsigned off1 = rowoff(s + 1);
                put32<T, NT>(o, TPG_LAMBDA_FF, off1, q.lam[2]); put32<T, NT>(o, TPG_PHI_FF, off1, q.phi[2]);
                put32<T, NT>(o, TPG_LAMBDA_CF, off1, q.lam[3]); put32<T, NT>(o, TPG_PHI_CF, off1, q.phi[3]);
            }
        }
        double (*L)[R][64] = lds.v;
        L[L_FC + 0][p][lane] = q.lam[0]; L[L_FC + 1][p][lane] = q.a[0]; L[L_FC + 2][p][lane] = q.ca[0];
        L[L_CC + 0][p][lane] = q.lam[1]; L[L_CC + 1][p][lane] = q.a[1]; L[L_CC + 2][p][lane] = q.ca[1];
        L[L_CC + 3][p][lane] = q.X[0];   L[L_CC + 4][p][lane] = q.Y[0]; L[L_CC + 5][p][lane] = q.Z[0];
        L[L_FF + 0][p][lane] = q.lam[2]; L[L_FF + 1][p][lane] = q.a[2]; L[L_FF + 2][p][lane] = q.ca[2];
        L[L_FF + 3][p][lane] = q.X[1];   L[L_FF + 4][p][lane] = q.Y[1]; L[L_FF + 5][p][lane] = q.Z[1];
        L[L_CF + 0][p][lane] = q.lam[3]; L[L_CF + 1][p][lane] = q.a[3]; L[L_CF + 2][p][lane] = q.ca[3];
    }
    __syncthreads();                                                         // the only barrier: point sets in LDS
    if (p == 0) {
        // The apron wave has no cell row of its own: instead of idling through phase 2 it computes one of the
        // eight haversines, Dy_ff = hav(FC(i,s), FC(i,s-1)), for every row of the tile from LDS (two rows at a time)
        if (!col_emit) return;
        double (*L)[R][64] = lds.v;
#pragma unroll 1
        for (int r = 1; r < R; r += 2) {
            const int sa = s0 + r, sb = sa + 1;
            if (sa > g.jm_hi) break;
            const bool two = (r + 1 < R) && sb <= g.jm_hi;
            const int rb = two ? r + 1 : r;
            Nb X[2] = { Nb{ L[L_FC + 0][r][lane], L[L_FC + 1][r][lane], L[L_FC + 2][r][lane] },
                        Nb{ L[L_FC + 0][rb][lane], L[L_FC + 1][rb][lane], L[L_FC + 2][rb][lane] } };
            Nb Y[2] = { Nb{ L[L_FC + 0][r - 1][lane], L[L_FC + 1][r - 1][lane], L[L_FC + 2][r - 1][lane] },
                        Nb{ L[L_FC + 0][rb - 1][lane], L[L_FC + 1][rb - 1][lane], L[L_FC + 2][rb - 1][lane] } };
            double dd2[2];
            hav_batch<2>(X, Y, Rad, dd2);
            if (sa >= g.jm_lo) put32<T, NT>(o, TPG_DY_FF, rowoff(sa), dd2[0]);
            if (two && sb >= g.jm_lo) put32<T, NT>(o, TPG_DY_FF, rowoff(sb), dd2[1]);
        }
        return;
    }
    if (!active_row || s < g.jm_lo || !col_emit) return;                    // idle rows and apron lanes are done

    // ---- phase 2: the cell (i, s) from own registers + LDS neighbours, loaded just in time so that the
    //      live set stays under 128 VGPRs (4 waves/SIMD supply the ILP; batches of 2 suffice)
    double (*L)[R][64] = lds.v;
    const int pm = p - 1, le = lane + 1, lw = lane - 1;
    const unsigned off = rowoff(s);

    // 2 spherical quadrilaterals first (unit vectors only): Az_cc from FF points, Az_ff from CC points
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        V3 a, b, c, dd;
        if (k == 0) {   // ffP = FF(i,s), ffEP = FF(i+1,s), ffE = FF(i+1,s+1), ff = FF(i,s+1)
            a = V3{ L[L_FF + 3][pm][lane], L[L_FF + 4][pm][lane], L[L_FF + 5][pm][lane] };
            b = V3{ L[L_FF + 3][pm][le], L[L_FF + 4][pm][le], L[L_FF + 5][pm][le] };
            c = V3{ L[L_FF + 3][p][le], L[L_FF + 4][p][le], L[L_FF + 5][p][le] };
            dd = V3{ q.X[1], q.Y[1], q.Z[1] };
        } else {        // ccWP = CC(i-1,s-1), ccP = CC(i,s-1), cc = CC(i,s), ccW = CC(i-1,s)
            a = V3{ L[L_CC + 3][pm][lw], L[L_CC + 4][pm][lw], L[L_CC + 5][pm][lw] };
            b = V3{ L[L_CC + 3][pm][lane], L[L_CC + 4][pm][lane], L[L_CC + 5][pm][lane] };
            c = V3{ q.X[0], q.Y[0], q.Z[0] };
            dd = V3{ L[L_CC + 3][p][lw], L[L_CC + 4][p][lw], L[L_CC + 5][p][lw] };
        }
        double tt[4], at[4];
        tt[0] = tri_tan_nr(a, b, c); tt[1] = tri_tan_nr(a, b, dd);
        tt[2] = tri_tan_nr(a, c, dd); tt[3] = tri_tan_nr(b, c, dd);
        if (__any(tpgb::atan_small_b<4>(tt, at))) {                 // wave-uniform fallback (large or degenerate triangles)
            tt[0] = tri_tan(a, b, c); tt[1] = tri_tan(a, b, dd); tt[2] = tri_tan(a, c, dd); tt[3] = tri_tan(b, c, dd);
            tpgb::atan_b<4>(tt, at);
        }
        // (2 t0 + 2 t1 + 2 t2 + 2 t3) / 2, summed left to right, is t0 + t1 + t2 + t3 summed left to right: doubling and halving are
        // exact and commute with every rounding (no overflow / underflow at these magnitudes) -- 5 multiplications less per quadrilateral
        double A = at[0];
        A += at[1];
        A += at[2];
        A += at[3];
        put32<T, NT>(o, k == 0 ? TPG_AZ_CC : TPG_AZ_FF, off, A * (Rad * Rad));
    }

    // 8 haversines in pairs; operand e = (x point, y point), each {lam, a, ca}
    //   0 dxcc(fcE,fc) 1 dxfc(cc,ccW) 2 dxcf(ffEP,ffP) 3 dxff(cfP,cfWP) 4 dycc(cf,cfP) 5 dyfc(ff,ffP) 6 dycf(cc,ccP) 7 dyff(fc,fcP)
    auto ld = [&](int f, int pp, int ll) -> Nb { return Nb{ L[f + 0][pp][ll], L[f + 1][pp][ll], L[f + 2][pp][ll] }; };
    const Nb own[4] = { Nb{ q.lam[0], q.a[0], q.ca[0] }, Nb{ q.lam[1], q.a[1], q.ca[1] },
                        Nb{ q.lam[2], q.a[2], q.ca[2] }, Nb{ q.lam[3], q.a[3], q.ca[3] } };   // fc cc ff cf
    double d[8];
#pragma unroll
    for (int hb = 0; hb < 6; hb += 2) {
        Nb X[2], Y[2];
        if (hb == 0)      { X[0] = ld(L_FC, p, le); Y[0] = own[0];            X[1] = own[1];            Y[1] = ld(L_CC, p, lw); }
        else if (hb == 2) { X[0] = ld(L_FF, pm, le); Y[0] = ld(L_FF, pm, lane); X[1] = ld(L_CF, pm, lane); Y[1] = ld(L_CF, pm, lw); }
        else              { X[0] = own[3];           Y[0] = ld(L_CF, pm, lane); X[1] = own[2];            Y[1] = ld(L_FF, pm, lane); }
        double dd2[2];
        hav_batch<2>(X, Y, Rad, dd2);
        d[hb] = dd2[0]; d[hb + 1] = dd2[1];
    }
    {   // Dy_cf here; Dy_ff is the apron wave's
        Nb X[1] = { own[1] }, Y[1] = { ld(L_CC, pm, lane) };
        double dd1[1];
        hav_batch<1>(X, Y, Rad, dd1);
        d[6] = dd1[0];
    }
    put32<T, NT>(o, TPG_DX_CC, off, d[0]); put32<T, NT>(o, TPG_DX_FC, off, d[1]);
    put32<T, NT>(o, TPG_DX_CF, off, d[2]); put32<T, NT>(o, TPG_DX_FF, off, d[3]);
    put32<T, NT>(o, TPG_DY_CC, off, d[4]); put32<T, NT>(o, TPG_DY_FC, off, d[5]);
    put32<T, NT>(o, TPG_DY_CF, off, d[6]);
    put32<T, NT>(o, TPG_AZ_FC, off, d[5] * d[1]);
    put32<T, NT>(o, TPG_AZ_CF, off, d[6] * d[2]);
}

#ifdef TPG_TEST_ABI
// The tile body of the halo-push form (k_cells_tile_push below, TEST library only): the tile kernel k_cells_tile, with the tile grid spanning
// the x halos; MODE 1 = the ordinary tiles, MODE 2 = the EDGE tile rows (north: fold images + row-Ny substitution; south: zero /
// continuation rows) -- chosen once per block, at the top of the kernel, from block-uniform values.  A COPY of k_cells_tile's body on
// purpose: sharing one templated body with the product's kernel cost that kernel 1.7-2 % on the 1/10 degree build (same-box A/B against
// the round-5 binary, profiles/r06/build_push_ab.txt), so the product's kernel below is left exactly as it was.  The structs come BY VALUE:
// by reference the kernel arguments left the SGPRs (+19 %).
template <typename T, bool NT, int R, int MODE>
__device__ __forceinline__ void cells_tile_body(const GridK g, const OutPtrs o, const int tiles_x, const HaloPush hp, const int ty,
                                                double (&atabs)[R][TPG_ATAN_TABLE_DOUBLES], TileLds<R>& lds)
{
    constexpr bool PUSH = MODE != 0, EDGE = MODE == 2;
    const int lane = threadIdx.x & 63;
    const int p = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);         // point row of this wave
    // one copy of the atan interval table per wave: its load overlaps the lane's other table loads and
    // needs no block barrier (LDS operations of one wave execute in order)
    double* atab = atabs[p];
    tpgb::atan_table_init(atab, lane);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int tx = blockIdx.x;
    const int s0 = g.jm_lo - 1 + ty * (R - 1);
    const int s = s0 + p;                                                    // this wave's step
    // columns: 1 .. Nx -- or, with the halo push, 1-Hx .. Nx+Hx: the x-halo columns are cells of the tile grid (wrapped below)
    const int i_hi = PUSH ? g.Nx + g.Hx : g.Nx;
    int i = tx * 62 + lane - (PUSH ? g.Hx : 0);
    const bool col_emit = lane >= 1 && lane <= 62 && i <= i_hi;
    if (i > i_hi + 1) i = i_hi + 1;
    const double Rad = g.R;
    const bool active_row = s <= g.jm_hi;                                    // rows past the band: idle waves
    const unsigned col = (unsigned)(i + g.Hx - 1);
    auto rowoff = [&](int j) -> unsigned { return (col + (unsigned)g.sx * (unsigned)(j - g.jstart + g.Hy)) * (unsigned)sizeof(T); };   // bytes
    // emit(): array q at location (xl, yl), row j -- the plain store, or (EDGE tiles) own cell + halo images (HaloPush above)
    auto emit = [&](int q, int xl, int yl, bool metric, int j, unsigned boff, double v) {
        if constexpr (EDGE) emit_edge<T, NT>(g, hp, o, q, xl, yl, metric, i, j, boff, v);
        else put32<T, NT>(o, q, boff, v);
    };

    // ---- phase 1: one point set per thread
    Step4 q;
    if (active_row) {
        const bool fast = s >= 1 && s < g.Ny && absD(g.fplp90) <= 360.0;     // wave-uniform
        if (fast) {
            LaneConst lc;
            const int iw = i < 1 ? i + g.Nx : (i > g.Nx ? i - g.Nx : i);
            int i0 = iw - g.shift; if (i0 < 1) i0 += g.Nx;
            lc.aslF = g.ti[0 * g.Nx + iw - 1]; lc.aclF = g.ti[1 * g.Nx + iw - 1];
            lc.aslC = g.ti[2 * g.Nx + iw - 1]; lc.aclC = g.ti[3 * g.Nx + iw - 1];
            lc.hemi = (i0 <= g.Nx / 2) ? -90.0 : 90.0;
            points_fast2(g, lc, load_rowtab(g, s, s + 1), q, atab);
        } else {
            Step4 tmp; GridK gc = g; points_general(gc, i, s, s + 1, tmp); q = tmp;
        }
        // coordinates: rows p >= 1 emit CC/FC(s) and FF/CF(s+1); the first tile's apron row emits FF/CF(jm_lo)
        if (col_emit) {
            if (p >= 1 && s >= g.jm_lo) {
                unsigned off = rowoff(s);
                emit(TPG_LAMBDA_FC, 1, 0, false, s, off, q.lam[0]); emit(TPG_PHI_FC, 1, 0, false, s, off, q.phi[0]);
                emit(TPG_LAMBDA_CC, 0, 0, false, s, off, q.lam[1]); emit(TPG_PHI_CC, 0, 0, false, s, off, q.phi[1]);
            }
            if ((p >= 1 || ty == 0) && s + 1 >= g.jm_lo && s + 1 <= g.jm_hi) {
                unsigned off1 = rowoff(s + 1);
                emit(TPG_LAMBDA_FF, 1, 1, false, s + 1, off1, q.lam[2]); emit(TPG_PHI_FF, 1, 1, false, s + 1, off1, q.phi[2]);
                emit(TPG_LAMBDA_CF, 0, 1, false, s + 1, off1, q.lam[3]); emit(TPG_PHI_CF, 0, 1, false, s + 1, off1, q.phi[3]);
            }
        }
        double (*L)[R][64] = lds.v;
        L[L_FC + 0][p][lane] = q.lam[0]; L[L_FC + 1][p][lane] = q.a[0]; L[L_FC + 2][p][lane] = q.ca[0];
        L[L_CC + 0][p][lane] = q.lam[1]; L[L_CC + 1][p][lane] = q.a[1]; L[L_CC + 2][p][lane] = q.ca[1];
        L[L_CC + 3][p][lane] = q.X[0];   L[L_CC + 4][p][lane] = q.Y[0]; L[L_CC + 5][p][lane] = q.Z[0];
        L[L_FF + 0][p][lane] = q.lam[2]; L[L_FF + 1][p][lane] = q.a[2]; L[L_FF + 2][p][lane] = q.ca[2];
        L[L_FF + 3][p][lane] = q.X[1];   L[L_FF + 4][p][lane] = q.Y[1]; L[L_FF + 5][p][lane] = q.Z[1];
        L[L_CF + 0][p][lane] = q.lam[3]; L[L_CF + 1][p][lane] = q.a[3]; L[L_CF + 2][p][lane] = q.ca[3];
    }
    __syncthreads();                                                         // the only barrier: point sets in LDS
    if (p == 0) {
        // The apron wave has no cell row of its own: instead of idling through phase 2 it computes one of the
        // eight haversines, Dy_ff = hav(FC(i,s), FC(i,s-1)), for every row of the tile from LDS (two rows at a time)
        if (!col_emit) return;
        if constexpr (EDGE) { if (hp.south && ty == 0) emit_south<T, NT>(g, hp, o, rowoff(1)); }      // rows j <= 1 / j < 1 of this column
        double (*L)[R][64] = lds.v;
#pragma unroll 1
        for (int r = 1; r < R; r += 2) {
            const int sa = s0 + r, sb = sa + 1;
            if (sa > g.jm_hi) break;
            const bool two = (r + 1 < R) && sb <= g.jm_hi;
            const int rb = two ? r + 1 : r;
            Nb X[2] = { Nb{ L[L_FC + 0][r][lane], L[L_FC + 1][r][lane], L[L_FC + 2][r][lane] },
                        Nb{ L[L_FC + 0][rb][lane], L[L_FC + 1][rb][lane], L[L_FC + 2][rb][lane] } };
            Nb Y[2] = { Nb{ L[L_FC + 0][r - 1][lane], L[L_FC + 1][r - 1][lane], L[L_FC + 2][r - 1][lane] },
                        Nb{ L[L_FC + 0][rb - 1][lane], L[L_FC + 1][rb - 1][lane], L[L_FC + 2][rb - 1][lane] } };
            double dd2[2];
            hav_batch<2>(X, Y, Rad, dd2);
            if (sa >= g.jm_lo) emit(TPG_DY_FF, 1, 1, true, sa, rowoff(sa), dd2[0]);
            if (two && sb >= g.jm_lo) emit(TPG_DY_FF, 1, 1, true, sb, rowoff(sb), dd2[1]);
        }
        return;
    }
    if (!active_row || s < g.jm_lo || !col_emit) return;                    // idle rows and apron lanes are done

    // ---- phase 2: the cell (i, s) from own registers + LDS neighbours, loaded just in time so that the
    //      live set stays under 128 VGPRs (4 waves/SIMD supply the ILP; batches of 2 suffice)
    double (*L)[R][64] = lds.v;
    const int pm = p - 1, le = lane + 1, lw = lane - 1;
    const unsigned off = rowoff(s);

    // 2 spherical quadrilaterals first (unit vectors only): Az_cc from FF points, Az_ff from CC points
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        V3 a, b, c, dd;
        if (k == 0) {   // ffP = FF(i,s), ffEP = FF(i+1,s), ffE = FF(i+1,s+1), ff = FF(i,s+1)
            a = V3{ L[L_FF + 3][pm][lane], L[L_FF + 4][pm][lane], L[L_FF + 5][pm][lane] };
            b = V3{ L[L_FF + 3][pm][le], L[L_FF + 4][pm][le], L[L_FF + 5][pm][le] };
            c = V3{ L[L_FF + 3][p][le], L[L_FF + 4][p][le], L[L_FF + 5][p][le] };
            dd = V3{ q.X[1], q.Y[1], q.Z[1] };
        } else {        // ccWP = CC(i-1,s-1), ccP = CC(i,s-1), cc = CC(i,s), ccW = CC(i-1,s)
            a = V3{ L[L_CC + 3][pm][lw], L[L_CC + 4][pm][lw], L[L_CC + 5][pm][lw] };
            b = V3{ L[L_CC + 3][pm][lane], L[L_CC + 4][pm][lane], L[L_CC + 5][pm][lane] };
            c = V3{ q.X[0], q.Y[0], q.Z[0] };
            dd = V3{ L[L_CC + 3][p][lw], L[L_CC + 4][p][lw], L[L_CC + 5][p][lw] };
        }
        double tt[4], at[4];
        tt[0] = tri_tan_nr(a, b, c); tt[1] = tri_tan_nr(a, b, dd);
        tt[2] = tri_tan_nr(a, c, dd); tt[3] = tri_tan_nr(b, c, dd);
        if (__any(tpgb::atan_small_b<4>(tt, at))) {                 // wave-uniform fallback (large or degenerate triangles)
            tt[0] = tri_tan(a, b, c); tt[1] = tri_tan(a, b, dd); tt[2] = tri_tan(a, c, dd); tt[3] = tri_tan(b, c, dd);
            tpgb::atan_b<4>(tt, at);
        }
        // (2 t0 + 2 t1 + 2 t2 + 2 t3) / 2, summed left to right, is t0 + t1 + t2 + t3 summed left to right: doubling and halving are
        // exact and commute with every rounding (no overflow / underflow at these magnitudes) -- 5 multiplications less per quadrilateral
        double A = at[0];
        A += at[1];
        A += at[2];
        A += at[3];
        emit(k == 0 ? TPG_AZ_CC : TPG_AZ_FF, k, k, true, s, off, A * (Rad * Rad));
    }

    // 8 haversines in pairs; operand e = (x point, y point), each {lam, a, ca}
    //   0 dxcc(fcE,fc) 1 dxfc(cc,ccW) 2 dxcf(ffEP,ffP) 3 dxff(cfP,cfWP) 4 dycc(cf,cfP) 5 dyfc(ff,ffP) 6 dycf(cc,ccP) 7 dyff(fc,fcP)
    auto ld = [&](int f, int pp, int ll) -> Nb { return Nb{ L[f + 0][pp][ll], L[f + 1][pp][ll], L[f + 2][pp][ll] }; };
    const Nb own[4] = { Nb{ q.lam[0], q.a[0], q.ca[0] }, Nb{ q.lam[1], q.a[1], q.ca[1] },
                        Nb{ q.lam[2], q.a[2], q.ca[2] }, Nb{ q.lam[3], q.a[3], q.ca[3] } };   // fc cc ff cf
    double d[8];
#pragma unroll
    for (int hb = 0; hb < 6; hb += 2) {
        Nb X[2], Y[2];
        if (hb == 0)      { X[0] = ld(L_FC, p, le); Y[0] = own[0];            X[1] = own[1];            Y[1] = ld(L_CC, p, lw); }
        else if (hb == 2) { X[0] = ld(L_FF, pm, le); Y[0] = ld(L_FF, pm, lane); X[1] = ld(L_CF, pm, lane); Y[1] = ld(L_CF, pm, lw); }
        else              { X[0] = own[3];           Y[0] = ld(L_CF, pm, lane); X[1] = own[2];            Y[1] = ld(L_FF, pm, lane); }
        double dd2[2];
        hav_batch<2>(X, Y, Rad, dd2);
        d[hb] = dd2[0]; d[hb + 1] = dd2[1];
    }
    {   // Dy_cf here; Dy_ff is the apron wave's
        Nb X[1] = { own[1] }, Y[1] = { ld(L_CC, pm, lane) };
        double dd1[1];
        hav_batch<1>(X, Y, Rad, dd1);
        d[6] = dd1[0];
    }
    emit(TPG_DX_CC, 0, 0, true, s, off, d[0]); emit(TPG_DX_FC, 1, 0, true, s, off, d[1]);
    emit(TPG_DX_CF, 0, 1, true, s, off, d[2]); emit(TPG_DX_FF, 1, 1, true, s, off, d[3]);
    emit(TPG_DY_CC, 0, 0, true, s, off, d[4]); emit(TPG_DY_FC, 1, 0, true, s, off, d[5]);
    emit(TPG_DY_CF, 0, 1, true, s, off, d[6]);
    emit(TPG_AZ_FC, 1, 0, true, s, off, d[5] * d[1]);
    emit(TPG_AZ_CF, 0, 1, true, s, off, d[6] * d[2]);
}

// The halo-push form (TPG_CELLS_VARIANT=3, test library only -- see the verdict on it at HaloPush above): K0 + this kernel is the whole build.
template <typename T, bool NT, int R>
__global__ __launch_bounds__(64 * R, 4) void k_cells_tile_push(GridK g, OutPtrs o, int tiles_x, HaloPush hp)
{
    __shared__ __attribute__((aligned(16))) double atabs[R][TPG_ATAN_TABLE_DOUBLES];
    __shared__ TileLds<R> lds;
    // the EDGE tile rows carry extra stores and must not be the last blocks of the launch, where their extra time would add to the
    // kernel's: the southernmost tile row is dispatched right after the northernmost, the others follow north to south
    const int ny_t = (int)gridDim.y, by = (int)blockIdx.y;
    const int ty = by == 0 ? ny_t - 1 : (by == 1 ? 0 : ny_t - by);
    // EDGE tile rows (block-uniform): those that reach row Ny - Hy (fold images, row-Ny substitution) and the southernmost one (south rows)
    const bool edge = g.jm_lo - 1 + ty * (R - 1) + R >= g.Ny - g.Hy || (hp.south && ty == 0);
    if (edge) cells_tile_body<T, NT, R, 2>(g, o, tiles_x, hp, ty, atabs, lds);
    else      cells_tile_body<T, NT, R, 1>(g, o, tiles_x, hp, ty, atabs, lds);
}
#endif

// ---- K2: halo cells of the 20 arrays ------------------------------------------------------------
// x/y location of array q (order of enum tpg_array)
__device__ __forceinline__ void array_loc(int q, int& xl, int& yl)
{
    // cc, fc, cf, ff for every group except dy (cc, cf, fc, ff)
    int r = q & 3;
    bool dy = (q >= TPG_DY_CC && q <= TPG_DY_FF);
    int xr = dy ? (r >> 1) : (r & 1);
    int yr = dy ? (r & 1) : (r >> 1);
    xl = xr; yl = yr;
}

struct HaloRegions {
    int nA;   // x-halo columns of evaluated rows: (jm_hi-jm_lo+1) * 2Hx
    int nB;   // north rows j = Ny+1..jend+Hy present in the band (all Hy of them on the north rank)
    int nC;   // south rows j < 1 present in the band (+ row 1 when K3 is merged): (nsouth + row1) * sx
    int nD;   // row-Ny substitution cells i = Nx/2+1..Nx (bands that contain row Ny): Nx/2
    int nsouth;
    int merged_south;   // 1: this launch also writes the lat-lon continuation rows j <= 1 of the 12 metrics (K3)
};

template <typename T>
__global__ __launch_bounds__(256) void k_halos(GridK g, OutPtrs o, HaloRegions h)
{
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    int q = blockIdx.y;
    int xl, yl;
    array_loc(q, xl, yl);
    const bool is_coord = q < TPG_DX_CC;
    T* A = static_cast<T*>(o.p[q]);
    const long long sx = g.sx;
    auto at = [&](int i, int j) -> long long { return (long long)(i + g.Hx - 1) + sx * (j - g.jstart + g.Hy); };

    int i, j;
    if (t < h.nA) {
        int r = t / (2 * g.Hx), c = t - r * 2 * g.Hx;
        j = g.jm_lo + r;
        i = c < g.Hx ? 1 - g.Hx + c : g.Nx + 1 + (c - g.Hx);
        if (h.merged_south && !is_coord && j <= 1) return;   // row 1 of the metrics is written whole by region C
    } else if ((t -= h.nA) < h.nB) {
        int r = t / g.sx, c = t - r * g.sx;
        j = g.Ny + 1 + r; i = 1 - g.Hx + c;
    } else if ((t -= h.nB) < h.nC) {
        int r = t / g.sx, c = t - r * g.sx;
        j = 1 - h.nsouth + r; i = 1 - g.Hx + c;
        if (is_coord) { if (j < 1) A[at(i, j)] = (T)0; return; }   // south = nothing: halos stay zero (tripolar_grid.jl:148)
        if (!h.merged_south) return;             // metrics: rows j <= 1 belong to K3
        // continue_south! (tripolar_grid.jl:287-300,336-357): Dx and Az by y-location, one Dy for all
        const int n = g.Hy + 1, rt = j - (1 - g.Hy);
        const int col = (q >= TPG_DY_CC && q <= TPG_DY_FF) ? 4 : ((q >= TPG_AZ_CC ? 2 : 0) + (yl == TPG_FACE ? 1 : 0));
        A[at(i, j)] = (T)g.ts[col * n + rt];
        return;
    } else if ((t -= h.nC) < h.nD) {
        if (is_coord || yl != TPG_CENTER) return; // coordinates were evaluated through the substitution already
        i = g.Nx / 2 + 1 + t; j = g.Ny;
    } else {
        return;
    }

    int iw = i < 1 ? i + g.Nx : (i > g.Nx ? i - g.Nx : i);
    int js = j;
    if (j > g.Ny) {
        int dj = j - g.Ny;
        js = (yl == TPG_FACE) ? g.Ny - dj + 1 : g.Ny - dj;
        iw = fold_partner(xl, iw, g.Nx);
        if (js < 1) { A[at(i, j)] = (T)0; return; }     // folds a zero south-halo row (Ny <= Hy grids)
    } else if (j == g.Ny && yl == TPG_CENTER && iw > g.Nx / 2) {
        iw = fold_partner(xl, iw, g.Nx);
    }
    A[at(i, j)] = A[at(iw, js)];
}

// ---- K3: lat-lon continuation rows (continue_south!, src/tripolar_grid.jl:287-300,336-357) ----
template <typename T>
__global__ __launch_bounds__(256) void k_south(GridK g, OutPtrs o)
{
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    int r = blockIdx.y;                              // global row j = 1-Hy+r
    int lr = (1 - g.Hy + r) - (g.jstart - g.Hy);     // local row of the band
    if (c >= g.sx || lr < 0 || lr >= g.rows) return;
    int n = g.Hy + 1;
    double dxc = g.ts[0 * n + r], dxf = g.ts[1 * n + r], azc = g.ts[2 * n + r], azf = g.ts[3 * n + r], dy = g.ts[4 * n + r];
    long long off = (long long)c + (long long)g.sx * lr;
    put<T>(o, TPG_DX_FF, off, dxf); put<T>(o, TPG_DX_FC, off, dxc); put<T>(o, TPG_DX_CF, off, dxf); put<T>(o, TPG_DX_CC, off, dxc);
    put<T>(o, TPG_DY_FF, off, dy);  put<T>(o, TPG_DY_FC, off, dy);  put<T>(o, TPG_DY_CF, off, dy);  put<T>(o, TPG_DY_CC, off, dy);
    put<T>(o, TPG_AZ_FF, off, azf); put<T>(o, TPG_AZ_FC, off, azc); put<T>(o, TPG_AZ_CF, off, azf); put<T>(o, TPG_AZ_CC, off, azc);
}

size_t table_doubles(const tpg_params* p) { return 4 * (size_t)p->Nx + 4 * (size_t)p->Ny + 5 * (size_t)(p->Hy + 1); }

int check_params(const tpg_params* p)
{
    if (!p) { tpg::set_error("null params"); return TPG_ERR_INVALID_ARGUMENT; }
    int rc = tpg::check_geom(p->Nx, p->Ny, p->Nz, p->Hx, p->Hy, p->Hz, p->ft);
    if (rc) return rc;
    if (p->jstart < 1 || p->jend > p->Ny || p->jend < p->jstart) {
        tpg::set_error("latitude band %d:%d outside 1:%d", p->jstart, p->jend, p->Ny);
        return TPG_ERR_BAD_PARTITION;
    }
    if (p->reserved & ~TPG_BUILD_TABLES_VALID) {
        tpg::set_error("tpg_params.reserved: unknown flag bits 0x%x", p->reserved & ~TPG_BUILD_TABLES_VALID);
        return TPG_ERR_INVALID_ARGUMENT;
    }
    if (!(p->radius > 0) || !(p->north_poles_latitude < 90) || !(p->southernmost_latitude < 90)) {
        tpg::set_error("invalid radius / latitudes");
        return TPG_ERR_INVALID_ARGUMENT;
    }
    return TPG_OK;
}

template <typename T>
int launch_build(const GridK& g, const OutPtrs& o, const HaloRegions& h, hipStream_t s)
{
    dim3 grid1(((g.Nx + 255) / 256) * (g.jm_hi - g.jm_lo + 1));
    // knobs (tpg::config(), read once; test library): TPG_CELLS_VARIANT 2 = k_cells_tile + k_halos (default, and the product's only form),
    // 3 = k_cells_tile_push (the tile kernel writes the halo cells too: one launch less; measured, not adopted -- HaloPush above),
    // 0 = k_cells (thread per cell) + k_halos -- the cross-checks: tests/test_gpu_variants.py; TPG_BUILD_NT 1 = streaming stores (default), 0 = plain
    const tpg::Config& cfg = tpg::config();
    const bool nt = cfg.build_nt;
    constexpr int R = 8;                                       // point rows per tile (16 = one block per CU: measured 25 % slower)
    const int nrows = g.jm_hi - g.jm_lo + 1;
    const int tiles_y = (nrows + (R - 1) - 1) / (R - 1);
    const bool offsets32 = (unsigned long long)g.sx * (unsigned long long)(g.jend - g.jstart + 1 + 2 * g.Hy) * sizeof(T) < (1ull << 32);
    const bool south_in_band = g.jstart - g.Hy <= 1;
    const bool tile = cfg.cells_variant != 0 && tiles_y <= 65535 && offsets32;     // tile rows ride on gridDim.y; stores use 32-bit byte offsets
#ifdef TPG_TEST_ABI
    // TPG_CELLS_VARIANT=3: the tile kernel writes the halo cells itself wherever sources and images are distinct cells
    if (tile && cfg.cells_variant == 3 && g.Ny > 2 * g.Hy + 2 && g.Nx >= 2 * g.Hx + 2) {
        HaloPush hp{ 1, g.Ny, south_in_band ? 1 : 0, g.jstart - g.Hy };
        if (g.jend + g.Hy > g.Ny) hp.jn_hi = g.jend + g.Hy < g.Ny + g.Hy ? g.jend + g.Hy : g.Ny + g.Hy;
        const int tiles_xp = (g.sx + 61) / 62;                           // the tile grid spans the x halos too
        dim3 gridp((unsigned)tiles_xp, (unsigned)tiles_y);
        if (nt) hipLaunchKernelGGL((k_cells_tile_push<T, true, R>), gridp, dim3(64 * R), 0, s, g, o, tiles_xp, hp);
        else    hipLaunchKernelGGL((k_cells_tile_push<T, false, R>), gridp, dim3(64 * R), 0, s, g, o, tiles_xp, hp);
        return tpg::launch_status("k_cells_tile_push");                 // K0 + K1: every halo cell has been written
    }
#endif
    if (tile) {
        const int tiles_x = (g.Nx + 61) / 62;
        dim3 gridt((unsigned)tiles_x, (unsigned)tiles_y);
        if (nt) hipLaunchKernelGGL((k_cells_tile<T, true, R>), gridt, dim3(64 * R), 0, s, g, o, tiles_x);
        else    hipLaunchKernelGGL((k_cells_tile<T, false, R>), gridt, dim3(64 * R), 0, s, g, o, tiles_x);
    }
    else if (nt) hipLaunchKernelGGL((k_cells<T, true>), grid1, dim3(256), 0, s, g, o);
    else         hipLaunchKernelGGL((k_cells<T, false>), grid1, dim3(256), 0, s, g, o);
    int rc = tpg::launch_status("k_cells");
    if (rc) return rc;
    // K3 rides in the K2 launch unless the grid is so short that a north-fold source row or the row-Ny
    // substitution could be a continuation row (then K3 must run after K2, as in the reference's order)
    HaloRegions hm = h;
    hm.merged_south = (south_in_band && g.Ny > 2 * g.Hy + 2) ? 1 : 0;
    if (hm.merged_south) hm.nC = (h.nsouth + 1) * g.sx;
    int nh = hm.nA + hm.nB + hm.nC + hm.nD;
    if (nh > 0) {
        hipLaunchKernelGGL(k_halos<T>, dim3((nh + 255) / 256, TPG_NUM_ARRAYS), dim3(256), 0, s, g, o, hm);
        if ((rc = tpg::launch_status("k_halos"))) return rc;
    }
    if (south_in_band && !hm.merged_south) {
        hipLaunchKernelGGL(k_south<T>, dim3((g.sx + 255) / 256, g.Hy + 1), dim3(256), 0, s, g, o);
        if ((rc = tpg::launch_status("k_south"))) return rc;
    }
    return TPG_OK;
}

}  // namespace

extern "C" {

size_t tpg_build_grid_workspace_bytes(const tpg_params* p)
{
    if (!p || p->Nx < 1 || p->Ny < 1 || p->Hy < 0) return 0;
    return (table_doubles(p) * sizeof(double) + 255) & ~(size_t)255;
}

int tpg_build_grid(const tpg_params* p, void* const out[TPG_NUM_ARRAYS], void* workspace,
                   size_t workspace_bytes, void* stream)
{
    int rc = check_params(p);
    if (rc) return rc;
    if (!out) { tpg::set_error("null output table"); return TPG_ERR_INVALID_ARGUMENT; }
    OutPtrs o;
    for (int q = 0; q < TPG_NUM_ARRAYS; ++q) {
        if (!out[q]) { tpg::set_error("null output array %d", q); return TPG_ERR_INVALID_ARGUMENT; }
        o.p[q] = out[q];
    }
    if (!workspace || workspace_bytes < table_doubles(p) * sizeof(double) || ((uintptr_t)workspace & 7)) {
        tpg::set_error("workspace: need %zu bytes, 8-byte aligned", table_doubles(p) * sizeof(double));
        return TPG_ERR_WORKSPACE;
    }
    hipStream_t s = tpg::as_stream(stream);
    double* w = static_cast<double*>(workspace);

    TableArgs t;
    t.Nx = p->Nx; t.Ny = p->Ny; t.Hy = p->Hy; t.shift = p->Nx / 4; t.ft = p->ft;
    t.south = p->southernmost_latitude; t.npl = p->north_poles_latitude; t.R = p->radius;
    t.ti = w; t.tj = w + 4 * (size_t)p->Nx; t.ts = t.tj + 4 * (size_t)p->Ny;
    if (!(p->reserved & TPG_BUILD_TABLES_VALID)) {       // the caller may vouch for tables left in the workspace by an earlier call
        int nt = p->Nx + 2 * p->Ny + p->Hy + 1;
        hipLaunchKernelGGL(k_tables, dim3((nt + 63) / 64), dim3(64), 0, s, t);
        if ((rc = tpg::launch_status("k_tables"))) return rc;
    }

    GridK g;
    g.Nx = p->Nx; g.Ny = p->Ny; g.Hx = p->Hx; g.Hy = p->Hy; g.shift = p->Nx / 4;
    g.jstart = p->jstart; g.jend = p->jend;
    g.jm_lo = p->jstart - p->Hy < 1 ? 1 : p->jstart - p->Hy;
    g.jm_hi = p->jend + p->Hy > p->Ny ? p->Ny : p->jend + p->Hy;
    g.sx = p->Nx + 2 * p->Hx; g.rows = p->jend - p->jstart + 1 + 2 * p->Hy;
    g.ft = p->ft; g.fplp90 = p->first_pole_longitude + 90.0; g.R = p->radius;
    g.ti = t.ti; g.tj = t.tj; g.ts = t.ts;

    HaloRegions h;
    h.nA = (g.jm_hi - g.jm_lo + 1) * 2 * g.Hx;
    // north fold rows / row-Ny substitution: on the north rank, and on any band whose halo reaches them
    // (bands thinner than the halo: the reference slices them out of the global padded arrays)
    h.nB = (p->jend + p->Hy > p->Ny) ? (p->jend + p->Hy - p->Ny) * g.sx : 0;
    h.nsouth = (1 - (p->jstart - p->Hy)) > 0 ? (1 - (p->jstart - p->Hy)) : 0;   // rows j < 1 in the band
    h.nC = h.nsouth * g.sx;
    h.nD = (p->jend + p->Hy >= p->Ny) ? p->Nx / 2 : 0;

    if (p->ft == TPG_F64) return launch_build<double>(g, o, h, s);
    return launch_build<float>(g, o, h, s);
}

}  // extern "C"
