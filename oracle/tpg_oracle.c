/*
 * oracle/tpg_oracle.c -- TEST INFRASTRUCTURE: CPU restatement (parity oracle + timed CPU baseline)
 * of the TripolarGrid metric-precompute and zipper halo-fill path of
 * CliMA/OrthogonalSphericalShellGrids.jl v0.2.1.  NOT product code: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 *
 * It follows the reference's own sequence of full-array passes step by step (tables -> dense
 * coordinates -> circshift -> Field set!/fill_halo_regions! -> metrics -> Field fills ->
 * continue_south! -> map(FT)), which is deliberately NOT how the HIP product is organised, so a
 * mistake in the product's fused index maps cannot be mirrored here by construction.
 *
 * Reference citations are relative to /root/reference.  "[recalled]" marks third-party semantics
 * (Julia Base, Oceananigans 0.95-0.99, Distances 0.10) whose sources are absent from this
 * container; they are restated from their published algorithms and pinned
 * only by README.md:52-60 (6 digits) and test/test_zipper_boundary_conditions.jl.
 * Parity status: zipper index/sign map PINNED (reference tests); Float64 metric values beyond 6
 * digits, Az^cc/Az^ff and the j<=1 lat-lon rows: "parity unpinned".
 *
 * Build: make -C oracle   (gcc -O2 -ffp-contract=off -fopenmp)
 */
#include "detmath.h"
#include <stdlib.h>
#include <stdio.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct {
    int32_t Nx, Ny, Nz;
    int32_t Hx, Hy, Hz;
    double southernmost_latitude;
    double north_poles_latitude;
    double first_pole_longitude;
    double radius;
    int32_t ft;      /* 0 = Float32, 1 = Float64 */
    int32_t jstart;  /* first owned global row (1-based); serial grid: 1 */
    int32_t jend;    /* last owned global row; serial grid: Ny */
    int32_t reserved;
} tpo_params;

enum { LOC_C = 0, LOC_F = 1 };

/* order of src/tripolar_grid.jl:308-328 (z omitted) */
enum {
    A_LCC, A_LFC, A_LCF, A_LFF, A_PCC, A_PFC, A_PCF, A_PFF,
    A_DXCC, A_DXFC, A_DXCF, A_DXFF, A_DYCC, A_DYCF, A_DYFC, A_DYFF,
    A_AZCC, A_AZFC, A_AZCF, A_AZFF, A_COUNT
};

static int g_threads = 1;
void tpo_set_threads(int n) { g_threads = n < 1 ? 1 : n; }
int tpo_get_threads(void) { return g_threads; }
int tpo_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ------------------------------------------------------------------------------------------
 * a1: 1-D tables (src/tripolar_grid.jl:73-97).
 * lambda tables: Oceananigans generate_coordinate on a regular (-180,180) interval [recalled]:
 * a Julia range whose elements are the correctly rounded exact rationals
 *   face   i : -180 + 360 (i-1)/N        center i : -180 + 360 (2i-1)/(2N)
 * (one IEEE division of two exactly representable integers is that rounding).  For FT=Float32
 * the range is a Float32 range (src/tripolar_grid.jl:90 passes FT).
 * phi tables: collect(range(south, 90, length=N)) -> RN(south + (90-south)(j-1)/(N-1)), always
 * Float64 (:95); dphi = phi_c[2]-phi_c[1] (:96); phi_f = phi_c .- dphi/2 (:97).
 * ------------------------------------------------------------------------------------------ */
static double lambda_face(int i, int N, int ft)
{
    long num = 360L * (i - 1) - 180L * N;
    if (ft == 0) return (double)((float)num / (float)N);
    return (double)num / (double)N;
}
static double lambda_center(int i, int N, int ft)
{
    long num = 360L * (2L * i - 1) - 360L * N;
    if (ft == 0) return (double)((float)num / (float)(2L * N));
    return (double)num / (double)(2L * N);
}
static double phi_center(int j, int N, double south)
{
    if (N == 1) return south;
    if (south == rint(south) && fabs(south) < 1e6) {
        long s = (long)south;
        long num = s * (N - 1) + (90 - s) * (long)(j - 1);
        return (double)num / (double)(N - 1);
    }
    /* non-integer start: twice-precision linear interpolation (Julia StepRangeLen) */
    dm_dd step = dm_dd_div(dm_two_sum(90.0, -south), dm_dd_make((double)(N - 1), 0.0));
    dm_dd v = dm_dd_add(dm_dd_make(south, 0.0), dm_dd_mul_d(step, (double)(j - 1)));
    return v.hi + v.lo;
}

void tpo_tables(const tpo_params *p, double *lam_f, double *lam_c, double *phi_f, double *phi_c)
{
    for (int i = 1; i <= p->Nx; ++i) {
        lam_f[i - 1] = lambda_face(i, p->Nx, p->ft);
        lam_c[i - 1] = lambda_center(i, p->Nx, p->ft);
    }
    for (int j = 1; j <= p->Ny; ++j) phi_c[j - 1] = phi_center(j, p->Ny, p->southernmost_latitude);
    double dphi = p->Ny > 1 ? phi_c[1] - phi_c[0] : 0.0;
    for (int j = 0; j < p->Ny; ++j) phi_f[j] = phi_c[j] - dphi / 2;
}

/* sind/cosd of a lambda-table element.  For FT=Float32 the table element is a Float32, so Julia
 * evaluates sind(::Float32) -> Float32 (correctly rounded from a Float64 evaluation) [recalled]. */
static double sind_ft(double lam, int ft) { double s = dm_sind(lam); return ft == 0 ? (double)(float)s : s; }
static double cosd_ft(double lam, int ft) { double c = dm_cosd(lam); return ft == 0 ? (double)(float)c : c; }

/* src/OrthogonalSphericalShellGrids.jl:24 */
static double convert_to_0_360(double x) { return fmod(fmod(x, 360.0) + 360.0, 360.0); }

/* ------------------------------------------------------------------------------------------
 * a2: one (i, j, location) evaluation of src/generate_tripolar_coordinates.jl:66-87.
 * sl = sind(lambda1D[i]), cl = cosd(lambda1D[i]), sh = sinh(psi), ch = cosh(psi) with
 * psi = asinh(tand((90 - phi1D[j]) / 2) / focal_distance)  (:66).
 * ------------------------------------------------------------------------------------------ */
static void tripolar_point(int i, int Nl, double a, double sl, double cl, double sh, double ch,
                           double fpl, double *lam2, double *phi2)
{
    double x = a * sl * ch;                                         /* :67 */
    double y = a * cl * sh;                                         /* :68 */
    int on_the_north_pole = (x == 0.0) & (y == 0.0);                /* :74 */
    double north_pole_value = (i == 1) ? -90.0 : 90.0;              /* :75 */
    double l = on_the_north_pole ? north_pole_value
                                 : -(180.0 / DM_PI) * dm_atan(y / x);   /* :77 */
    *phi2 = 90.0 - (360.0 / DM_PI) * dm_atan(sqrt(y * y + x * x));  /* :78 */
    l += (i <= Nl / 2) ? -90.0 : 90.0;                              /* :82 */
    l += fpl + 90.0;                                                /* :86 */
    *lam2 = convert_to_0_360(l);                                    /* :87 */
}

/* ------------------------------------------------------------------------------------------
 * padded-array helpers.  Logical (i,j,k), i in 1-Hx..Nx+Hx etc., column-major, i fastest.
 * ------------------------------------------------------------------------------------------ */
typedef struct { int Nx, Ny, Nz, Hx, Hy, Hz; size_t sx, sy; } dims_t;
static dims_t mkdims(int Nx, int Ny, int Nz, int Hx, int Hy, int Hz)
{
    dims_t d = { Nx, Ny, Nz, Hx, Hy, Hz, (size_t)(Nx + 2 * Hx), (size_t)(Ny + 2 * Hy) };
    return d;
}
static inline size_t IDX(const dims_t *d, int i, int j, int k)
{
    return (size_t)(i + d->Hx - 1) + d->sx * ((size_t)(j + d->Hy - 1) + d->sy * (size_t)(k + d->Hz - 1));
}

/* ------------------------------------------------------------------------------------------
 * a5-a8: the four fold functions, src/zipper_boundary_condition.jl:70-138, one (i,k) column.
 * `sign * c[...]` is an Int * Float multiplication in the reference; restated as a multiply.
 * ------------------------------------------------------------------------------------------ */
#define DEFINE_FOLDS(T, SUF)                                                                        \
static void fold_north_face_face_##SUF(int i, int k, const dims_t *d, int sign, T *c)              \
{                                                                                                   \
    int Nx = d->Nx, Ny = d->Ny;                                                                     \
    int ip = Nx - i + 2;                                            /* :73 */                       \
    sign = ip > Nx ? abs(sign) : sign;                              /* :74 */                       \
    ip = ip > Nx ? ip - Nx : ip;                                    /* :75 */                       \
    for (int j = 1; j <= d->Hy; ++j)                                                                \
        c[IDX(d, i, Ny + j, k)] = (T)sign * c[IDX(d, ip, Ny - j + 1, k)];   /* :80 */               \
}                                                                                                   \
static void fold_north_face_center_##SUF(int i, int k, const dims_t *d, int sign, T *c)            \
{                                                                                                   \
    int Nx = d->Nx, Ny = d->Ny;                                                                     \
    int ip = Nx - i + 2;                                            /* :90 */                       \
    sign = ip > Nx ? abs(sign) : sign;                              /* :91 */                       \
    ip = ip > Nx ? ip - Nx : ip;                                    /* :92 */                       \
    for (int j = 1; j <= d->Hy; ++j)                                                                \
        c[IDX(d, i, Ny + j, k)] = (T)sign * c[IDX(d, ip, Ny - j, k)];       /* :97 */               \
    if (i > Nx / 2)                                                 /* :102 */                      \
        c[IDX(d, i, Ny, k)] = (T)sign * c[IDX(d, ip, Ny, k)];                                       \
}                                                                                                   \
static void fold_north_center_face_##SUF(int i, int k, const dims_t *d, int sign, T *c)            \
{                                                                                                   \
    int Nx = d->Nx, Ny = d->Ny;                                                                     \
    int ip = Nx - i + 1;                                            /* :110 */                      \
    for (int j = 1; j <= d->Hy; ++j)                                                                \
        c[IDX(d, i, Ny + j, k)] = (T)sign * c[IDX(d, ip, Ny - j + 1, k)];   /* :115 */              \
}                                                                                                   \
static void fold_north_center_center_##SUF(int i, int k, const dims_t *d, int sign, T *c)          \
{                                                                                                   \
    int Nx = d->Nx, Ny = d->Ny;                                                                     \
    int ip = Nx - i + 1;                                            /* :125 */                      \
    for (int j = 1; j <= d->Hy; ++j)                                                                \
        c[IDX(d, i, Ny + j, k)] = (T)sign * c[IDX(d, ip, Ny - j, k)];       /* :130 */              \
    if (i > Nx / 2)                                                 /* :135 */                      \
        c[IDX(d, i, Ny, k)] = (T)sign * c[IDX(d, ip, Ny, k)];                                       \
}                                                                                                   \
/* location dispatch, src/zipper_boundary_condition.jl:140-155; launch range (i,k) in            */ \
/* 1..Nx x kstart..kstart+kcount-1 (Oceananigans' :xz south/north halo kernel [recalled])         */ \
static void zipper_fill_##SUF(T *c, const dims_t *d, int xloc, int yloc, int sign,                  \
                              int kstart, int kcount)                                               \
{                                                                                                   \
    _Pragma("omp parallel for num_threads(g_threads) schedule(static)")                             \
    for (int k = kstart; k < kstart + kcount; ++k)                                                  \
        for (int i = 1; i <= d->Nx; ++i) {                                                          \
            if (xloc == LOC_C && yloc == LOC_C)      fold_north_center_center_##SUF(i, k, d, sign, c); \
            else if (xloc == LOC_F && yloc == LOC_C) fold_north_face_center_##SUF(i, k, d, sign, c);   \
            else if (xloc == LOC_C && yloc == LOC_F) fold_north_center_face_##SUF(i, k, d, sign, c);   \
            else                                     fold_north_face_face_##SUF(i, k, d, sign, c);     \
        }                                                                                           \
}                                                                                                   \
/* Oceananigans periodic west/east fill [recalled]: runs AFTER the zipper (pinned by              */ \
/* test/test_zipper_boundary_conditions.jl:42-45) over every row and level of the parent array,   */ \
/* which is what fills the north-halo corners.                                                    */ \
static void periodic_x_fill_##SUF(T *c, const dims_t *d)                                            \
{                                                                                                   \
    _Pragma("omp parallel for num_threads(g_threads) schedule(static)")                             \
    for (int k = 1 - d->Hz; k <= d->Nz + d->Hz; ++k)                                                \
        for (int j = 1 - d->Hy; j <= d->Ny + d->Hy; ++j)                                            \
            for (int h = 1; h <= d->Hx; ++h) {                                                      \
                c[IDX(d, 1 - h, j, k)] = c[IDX(d, d->Nx - h + 1, j, k)];                            \
                c[IDX(d, d->Nx + h, j, k)] = c[IDX(d, h, j, k)];                                    \
            }                                                                                       \
}

DEFINE_FOLDS(double, f64)
DEFINE_FOLDS(float, f32)

/* C entry points for 3-D fields (config 3).  ft: 0 = Float32, 1 = Float64. */
int tpo_zipper_fill(void *field, int xloc, int yloc, int sign, int Nx, int Ny, int Nz,
                    int Hx, int Hy, int Hz, int kstart, int kcount, int ft)
{
    dims_t d = mkdims(Nx, Ny, Nz, Hx, Hy, Hz);
    if (ft == 1) zipper_fill_f64((double *)field, &d, xloc, yloc, sign, kstart, kcount);
    else         zipper_fill_f32((float *)field, &d, xloc, yloc, sign, kstart, kcount);
    return 0;
}
int tpo_periodic_x_fill(void *field, int Nx, int Ny, int Nz, int Hx, int Hy, int Hz, int ft)
{
    dims_t d = mkdims(Nx, Ny, Nz, Hx, Hy, Hz);
    if (ft == 1) periodic_x_fill_f64((double *)field, &d);
    else         periodic_x_fill_f32((float *)field, &d);
    return 0;
}
/* fill_halo_regions! on a tripolar field: zipper first, periodic x second */
int tpo_fill_halo_regions(void *field, int xloc, int yloc, int sign, int Nx, int Ny, int Nz,
                          int Hx, int Hy, int Hz, int ft)
{
    tpo_zipper_fill(field, xloc, yloc, sign, Nx, Ny, Nz, Hx, Hy, Hz, 1, Nz, ft);
    return tpo_periodic_x_fill(field, Nx, Ny, Nz, Hx, Hy, Hz, ft);
}

/* ------------------------------------------------------------------------------------------
 * a7 of Appendix A: Distances.haversine [recalled], call sites src/tripolar_grid_utils.jl:13-21.
 * Points are (lambda, phi) in degrees.
 * ------------------------------------------------------------------------------------------ */
static double haversine(double l1, double p1, double l2, double p2, double radius)
{
    double dl = (l2 - l1) * DM_DEG2RAD;
    double a1 = p1 * DM_DEG2RAD;
    double a2 = p2 * DM_DEG2RAD;
    double dp = a2 - a1;
    double s1 = dm_sin(dp / 2), s2 = dm_sin(dl / 2);
    double a = s1 * s1 + dm_cos(a1) * dm_cos(a2) * (s2 * s2);
    double r = sqrt(a);
    return 2 * (radius * dm_asin(r != r ? r : (r < 1.0 ? r : 1.0)));   /* min(sqrt(a), 1) */
}

/* Oceananigans lat_lon_to_cartesian(phi, lambda, 1) [recalled] */
typedef struct { double x, y, z; } vec3;
static vec3 lat_lon_to_cartesian(double lat, double lon)
{
    vec3 v;
    double cl = dm_cosd(lat);
    v.x = dm_cosd(lon) * cl;
    v.y = dm_sind(lon) * cl;
    v.z = dm_sind(lat);
    return v;
}
static double dot3(vec3 a, vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static vec3 cross3(vec3 a, vec3 b)
{
    vec3 c = { a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x };
    return c;
}
/* Oceananigans spherical_area_triangle (Eriksson 1990) [recalled] */
static double spherical_area_triangle(vec3 a, vec3 b, vec3 c)
{
    double t = fabs(dot3(a, cross3(b, c)));
    t /= 1 + dot3(a, b) + dot3(b, c) + dot3(a, c);
    return 2 * dm_atan(t);
}
/* Oceananigans spherical_area_quadrilateral [recalled] */
static double spherical_area_quadrilateral(vec3 a, vec3 b, vec3 c, vec3 d)
{
    double A = spherical_area_triangle(a, b, c);
    A += spherical_area_triangle(a, b, d);
    A += spherical_area_triangle(a, c, d);
    A += spherical_area_triangle(b, c, d);
    return A / 2;
}

/* ------------------------------------------------------------------------------------------
 * the global grid in Float64, following src/tripolar_grid.jl:59-330 pass by pass.
 * G[A_*] are padded (Nx+2Hx) x (Ny+2Hy) arrays.
 * ------------------------------------------------------------------------------------------ */
static double *dalloc(size_t n)
{
    double *p = (double *)calloc(n ? n : 1, sizeof(double));
    if (!p) { fprintf(stderr, "tpo: out of memory\n"); abort(); }
    return p;
}

static void circshift_rows(const double *A, double *B, int N, int M, int shift)
{
    /* circshift(A, (shift, 0)):  B[mod1(i + shift, N), j] = A[i, j]   (:121-130) */
    for (int j = 0; j < M; ++j)
        for (int i = 0; i < N; ++i)
            B[(size_t)((i + shift) % N) + (size_t)N * j] = A[(size_t)i + (size_t)N * j];
}

/* Field set! + fill_halo_regions! + dropdims  (:154-199 and :230-273) */
static void set_and_fill(double *P, const dims_t *d, const double *dense, int xloc, int yloc)
{
    for (int j = 1; j <= d->Ny; ++j)
        for (int i = 1; i <= d->Nx; ++i)
            P[IDX(d, i, j, 1)] = dense[(size_t)(i - 1) + (size_t)d->Nx * (j - 1)];
    zipper_fill_f64(P, d, xloc, yloc, +1, 1, 1);    /* ZipperBoundaryCondition() == sign +1 (:147) */
    periodic_x_fill_f64(P, d);                      /* west/east periodic (:149-150); south = nothing */
}

static double **build_global(const tpo_params *p)
{
    const int Nl = p->Nx, Np = p->Ny, Hx = p->Hx, Hy = p->Hy;
    const int Nx = Nl, Ny = Np;
    const double R = p->radius, fpl = p->first_pole_longitude;
    const double a = dm_tand((90.0 - p->north_poles_latitude) / 2);      /* focal_distance, :76 */
    const size_t nd = (size_t)Nl * Np;
    dims_t d = mkdims(Nx, Ny, 1, Hx, Hy, 0);
    const size_t np = d.sx * d.sy;

    double *lam_f = dalloc(Nl), *lam_c = dalloc(Nl), *phi_f = dalloc(Np), *phi_c = dalloc(Np);
    tpo_tables(p, lam_f, lam_c, phi_f, phi_c);

    /* separable factors (identical values to evaluating :66-68 per cell) */
    double *slf = dalloc(Nl), *clf = dalloc(Nl), *slc = dalloc(Nl), *clc = dalloc(Nl);
    double *shf = dalloc(Np), *chf = dalloc(Np), *shc = dalloc(Np), *chc = dalloc(Np);
    for (int i = 0; i < Nl; ++i) {
        slf[i] = sind_ft(lam_f[i], p->ft); clf[i] = cosd_ft(lam_f[i], p->ft);
        slc[i] = sind_ft(lam_c[i], p->ft); clc[i] = cosd_ft(lam_c[i], p->ft);
    }
    for (int j = 0; j < Np; ++j) {
        double psi_f = dm_asinh(dm_tand((90.0 - phi_f[j]) / 2) / a);     /* :66 */
        double psi_c = dm_asinh(dm_tand((90.0 - phi_c[j]) / 2) / a);
        dm_sinh_cosh(psi_f, &shf[j], &chf[j]);
        dm_sinh_cosh(psi_c, &shc[j], &chc[j]);
    }

    /* a2: dense coordinates before the shift (:102-117) */
    double *dn[8], *sh[8];
    for (int q = 0; q < 8; ++q) { dn[q] = dalloc(nd); sh[q] = dalloc(nd); }
    /* q: 0 lFF 1 pFF 2 lFC 3 pFC 4 lCF 5 pCF 6 lCC 7 pCC */
#pragma omp parallel for num_threads(g_threads) schedule(static)
    for (int j = 1; j <= Np; ++j)
        for (int i = 1; i <= Nl; ++i) {
            size_t o = (size_t)(i - 1) + (size_t)Nl * (j - 1);
            tripolar_point(i, Nl, a, slf[i - 1], clf[i - 1], shf[j - 1], chf[j - 1], fpl, &dn[0][o], &dn[1][o]);
            tripolar_point(i, Nl, a, slf[i - 1], clf[i - 1], shc[j - 1], chc[j - 1], fpl, &dn[2][o], &dn[3][o]);
            tripolar_point(i, Nl, a, slc[i - 1], clc[i - 1], shf[j - 1], chf[j - 1], fpl, &dn[4][o], &dn[5][o]);
            tripolar_point(i, Nl, a, slc[i - 1], clc[i - 1], shc[j - 1], chc[j - 1], fpl, &dn[6][o], &dn[7][o]);
        }
    /* a3: circshift by Nl/4 (:121-130) */
    for (int q = 0; q < 8; ++q) circshift_rows(dn[q], sh[q], Nl, Np, Nl / 4);

    double **G = (double **)calloc(A_COUNT, sizeof(double *));
    for (int q = 0; q < A_COUNT; ++q) G[q] = dalloc(np);

    /* a4: coordinate halo fill (:154-199) */
    set_and_fill(G[A_LFF], &d, sh[0], LOC_F, LOC_F); set_and_fill(G[A_PFF], &d, sh[1], LOC_F, LOC_F);
    set_and_fill(G[A_LFC], &d, sh[2], LOC_F, LOC_C); set_and_fill(G[A_PFC], &d, sh[3], LOC_F, LOC_C);
    set_and_fill(G[A_LCF], &d, sh[4], LOC_C, LOC_F); set_and_fill(G[A_PCF], &d, sh[5], LOC_C, LOC_F);
    set_and_fill(G[A_LCC], &d, sh[6], LOC_C, LOC_C); set_and_fill(G[A_PCC], &d, sh[7], LOC_C, LOC_C);
    for (int q = 0; q < 8; ++q) { free(dn[q]); free(sh[q]); }

    /* a10: metrics (src/tripolar_grid_utils.jl:4-45) on the halo-filled coordinates */
    double *m[12];
    for (int q = 0; q < 12; ++q) m[q] = dalloc(nd);
    enum { DXCC, DXFC, DXCF, DXFF, DYCC, DYFC, DYCF, DYFF, AZCC, AZFC, AZCF, AZFF };
#define C(arr, i, j) G[arr][IDX(&d, (i), (j), 1)]
#define HAV(AL, AP, i1, j1, i2, j2) haversine(C(AL, i1, j1), C(AP, i1, j1), C(AL, i2, j2), C(AP, i2, j2), R)
#pragma omp parallel for num_threads(g_threads) schedule(static)
    for (int j = 1; j <= Ny; ++j)
        for (int i = 1; i <= Nx; ++i) {
            size_t o = (size_t)(i - 1) + (size_t)Nx * (j - 1);
            m[DXCC][o] = HAV(A_LFC, A_PFC, i + 1, j, i, j);              /* :13 */
            m[DXFC][o] = HAV(A_LCC, A_PCC, i, j, i - 1, j);              /* :14 */
            m[DXCF][o] = HAV(A_LFF, A_PFF, i + 1, j, i, j);              /* :15 */
            m[DXFF][o] = HAV(A_LCF, A_PCF, i, j, i - 1, j);              /* :16 */
            m[DYCC][o] = HAV(A_LCF, A_PCF, i, j + 1, i, j);              /* :18 */
            m[DYFC][o] = HAV(A_LFF, A_PFF, i, j + 1, i, j);              /* :19 */
            m[DYCF][o] = HAV(A_LCC, A_PCC, i, j, i, j - 1);              /* :20 */
            m[DYFF][o] = HAV(A_LFC, A_PFC, i, j, i, j - 1);              /* :21 */
            vec3 va = lat_lon_to_cartesian(C(A_PFF, i, j), C(A_LFF, i, j));              /* :23-26 */
            vec3 vb = lat_lon_to_cartesian(C(A_PFF, i + 1, j), C(A_LFF, i + 1, j));
            vec3 vc = lat_lon_to_cartesian(C(A_PFF, i + 1, j + 1), C(A_LFF, i + 1, j + 1));
            vec3 vd = lat_lon_to_cartesian(C(A_PFF, i, j + 1), C(A_LFF, i, j + 1));
            m[AZCC][o] = spherical_area_quadrilateral(va, vb, vc, vd) * (R * R);         /* :28 */
            m[AZFC][o] = m[DYFC][o] * m[DXFC][o];                        /* :34 */
            m[AZCF][o] = m[DYCF][o] * m[DXCF][o];                        /* :35 */
            va = lat_lon_to_cartesian(C(A_PCC, i - 1, j - 1), C(A_LCC, i - 1, j - 1));   /* :38-41 */
            vb = lat_lon_to_cartesian(C(A_PCC, i, j - 1), C(A_LCC, i, j - 1));
            vc = lat_lon_to_cartesian(C(A_PCC, i, j), C(A_LCC, i, j));
            vd = lat_lon_to_cartesian(C(A_PCC, i - 1, j), C(A_LCC, i - 1, j));
            m[AZFF][o] = spherical_area_quadrilateral(va, vb, vc, vd) * (R * R);         /* :43 */
        }
#undef HAV
#undef C

    /* a11: metric halo fills with the reused FF/CF/FC/CC fields (:230-273) */
    set_and_fill(G[A_DXFF], &d, m[DXFF], LOC_F, LOC_F); set_and_fill(G[A_DXCF], &d, m[DXCF], LOC_C, LOC_F);
    set_and_fill(G[A_DXFC], &d, m[DXFC], LOC_F, LOC_C); set_and_fill(G[A_DXCC], &d, m[DXCC], LOC_C, LOC_C);
    set_and_fill(G[A_DYFF], &d, m[DYFF], LOC_F, LOC_F); set_and_fill(G[A_DYCF], &d, m[DYCF], LOC_C, LOC_F);
    set_and_fill(G[A_DYFC], &d, m[DYFC], LOC_F, LOC_C); set_and_fill(G[A_DYCC], &d, m[DYCC], LOC_C, LOC_C);
    set_and_fill(G[A_AZFF], &d, m[AZFF], LOC_F, LOC_F); set_and_fill(G[A_AZCF], &d, m[AZCF], LOC_C, LOC_F);
    set_and_fill(G[A_AZFC], &d, m[AZFC], LOC_F, LOC_C); set_and_fill(G[A_AZCC], &d, m[AZCC], LOC_C, LOC_C);
    for (int q = 0; q < 12; ++q) free(m[q]);

    /* a12: continue_south! from a regular LatitudeLongitudeGrid (:277-300, :336-357).
     * Oceananigans lat-lon metrics [recalled, SURVEY.md A-8]:
     *   dlam = 360/Nx, dphi_L = (90-south)/Ny, phi_f_L[j] = south + (j-1) dphi_L, phi_c_L = +dphi_L/2
     *   dx{fc,cc}[j] = R deg2rad(dlam) cos(pi phi_c_L[j]/180),  dx{cf,ff}[j] with phi_f_L[j]
     *   dy = R deg2rad(dphi_L)   (a Number: regular latitude)
     *   Az{fc,cc}[j] = R^2 deg2rad(dlam) (sin(pi phi_f_L[j+1]/180) - sin(pi phi_f_L[j]/180))
     *   Az{cf,ff}[j] = R^2 deg2rad(dlam) (sin(pi phi_c_L[j]/180)   - sin(pi phi_c_L[j-1]/180))
     * rows j = 1-Hy..1 (interior row 1 included), all i = 1-Hx..Nx+Hx. */
    {
        const double south = p->southernmost_latitude;
        const double dlam = 360.0 / (double)Nx;
        const double dphiL = (90.0 - south) / (double)Ny;
        const double dy = R * (dphiL * DM_DEG2RAD);
        for (int j = 1 - Hy; j <= 1; ++j) {
            /* range elements: exact-rational rounding of south + (j-1)*dphi_L (+dphi_L/2) */
            double pf = south + (double)(j - 1) * dphiL, pfn = south + (double)j * dphiL;
            double pc = south + ((double)(2 * j - 1) * (90.0 - south)) / (double)(2 * Ny);
            double pcm = south + ((double)(2 * j - 3) * (90.0 - south)) / (double)(2 * Ny);
            if (south == rint(south)) {      /* exact rationals, one rounding */
                long s = (long)south;
                pf = (double)(s * Ny + (90 - s) * (long)(j - 1)) / (double)Ny;
                pfn = (double)(s * Ny + (90 - s) * (long)j) / (double)Ny;
                pc = (double)(2 * s * Ny + (90 - s) * (long)(2 * j - 1)) / (double)(2 * Ny);
                pcm = (double)(2 * s * Ny + (90 - s) * (long)(2 * j - 3)) / (double)(2 * Ny);
            }
            double dxc = R * (dlam * DM_DEG2RAD) * dm_cos(DM_PI * pc / 180);
            double dxf = R * (dlam * DM_DEG2RAD) * dm_cos(DM_PI * pf / 180);
            double azc = R * R * (dlam * DM_DEG2RAD) * (dm_sin(DM_PI * pfn / 180) - dm_sin(DM_PI * pf / 180));
            double azf = R * R * (dlam * DM_DEG2RAD) * (dm_sin(DM_PI * pc / 180) - dm_sin(DM_PI * pcm / 180));
            for (int i = 1 - Hx; i <= Nx + Hx; ++i) {
                size_t o = IDX(&d, i, j, 1);
                G[A_DXFF][o] = dxf; G[A_DXFC][o] = dxc; G[A_DXCF][o] = dxf; G[A_DXCC][o] = dxc;   /* :287-290 */
                G[A_DYFF][o] = dy;  G[A_DYFC][o] = dy;  G[A_DYCF][o] = dy;  G[A_DYCC][o] = dy;    /* :292-295 */
                G[A_AZFF][o] = azf; G[A_AZFC][o] = azc; G[A_AZCF][o] = azf; G[A_AZCC][o] = azc;   /* :297-300 */
            }
        }
    }

    free(lam_f); free(lam_c); free(phi_f); free(phi_c);
    free(slf); free(clf); free(slc); free(clc); free(shf); free(chf); free(shc); free(chc);
    return G;
}

/* a13 + a15: map(FT, .) (src/tripolar_grid.jl:308-328) and the latitude-band slice
 * jstart-Hy : jend+Hy (src/distributed_tripolar_grid.jl:47-49, 112-120).
 * out[q]: caller-owned (Nx+2Hx) x (jend-jstart+1+2Hy) arrays of FT, order of the enum above. */
int tpo_build_grid(const tpo_params *p, void *const out[A_COUNT])
{
    if (p->Nx % 2) return -2;                        /* ArgumentError, src/tripolar_grid.jl:81-83 */
    if (p->jstart < 1 || p->jend > p->Ny || p->jend < p->jstart) return -3;
    double **G = build_global(p);
    const size_t sx = (size_t)(p->Nx + 2 * p->Hx);
    const int rows = p->jend - p->jstart + 1 + 2 * p->Hy;
    for (int q = 0; q < A_COUNT; ++q) {
        const double *src = G[q] + sx * (size_t)(p->jstart - 1);   /* global row jstart-Hy */
        size_t n = sx * (size_t)rows;
        if (p->ft == 1) memcpy(out[q], src, n * sizeof(double));
        else { float *o = (float *)out[q]; for (size_t t = 0; t < n; ++t) o[t] = (float)src[t]; }
        free(G[q]);
    }
    free(G);
    return 0;
}

/* elementary-function probes for tests/test_detmath.py */
void tpo_math_probe(int which, const double *x, double *y, long n)
{
    for (long t = 0; t < n; ++t) {
        double v = x[t], s, c;
        switch (which) {
        case 0: y[t] = dm_sin(v); break;
        case 1: y[t] = dm_cos(v); break;
        case 2: y[t] = dm_sind(v); break;
        case 3: y[t] = dm_cosd(v); break;
        case 4: y[t] = dm_tand(v); break;
        case 5: y[t] = dm_atan(v); break;
        case 6: y[t] = dm_asin(v); break;
        case 7: y[t] = dm_asinh(v); break;
        case 8: dm_sinh_cosh(v, &s, &c); y[t] = s; break;
        case 9: dm_sinh_cosh(v, &s, &c); y[t] = c; break;
        case 10: y[t] = dm_acos(v); break;
        default: y[t] = 0.0;
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * SURVEY 8(f-4) geometry utilities.
 *
 * compute_nonorthogonality_angle! (test/test_tripolar_grid.jl:8-34), launched over (Nx-1, Ny-1) (:70):
 *   P = cartesian FF node (get_cartesian_nodes_and_vertices(grid, Face(), Face(), Center()) [recalled]: the unit
 *   vector lat_lon_to_cartesian(phi, lambda, 1)); v1 = P[i+1,j] - P[i,j]; v2 = P[i,j+1] - P[i,j];
 *   cos = dot(v1, v2) / (norm(v1) * norm(v2)); angle = rad2deg(ifelse(immersed, pi/2, acos(cos)) - pi/2).
 * dot / norm of 3-tuples [recalled, LinearAlgebra generic]: left-to-right sums, norm = sqrt(sum of squares).
 * lam_ff / phi_ff: padded (Nx+2Hx) x (Ny+2Hy) arrays of the grid's FT; angle: dense Nx x Ny Float64, i fastest
 * (zeros(size(grid)...), :64: entries with i = Nx or j = Ny stay 0); immersed: dense Nx x Ny bytes or NULL.
 * ------------------------------------------------------------------------------------------ */
static double ldft(const void *a, size_t idx, int ft) { return ft == 0 ? (double)((const float *)a)[idx] : ((const double *)a)[idx]; }

int tpo_nonorthogonality_angle(const void *lam_ff, const void *phi_ff, const unsigned char *immersed, double *angle,
                               int Nx, int Ny, int Hx, int Hy, int ft)
{
    const size_t sx = (size_t)Nx + 2 * Hx;
    for (size_t n = 0; n < (size_t)Nx * Ny; ++n) angle[n] = 0.0;
    for (int j = 1; j <= Ny - 1; ++j)
        for (int i = 1; i <= Nx - 1; ++i) {
            size_t c = (size_t)(i + Hx - 1) + sx * (size_t)(j + Hy - 1);
            vec3 p0 = lat_lon_to_cartesian(ldft(phi_ff, c, ft), ldft(lam_ff, c, ft));
            vec3 p1 = lat_lon_to_cartesian(ldft(phi_ff, c + 1, ft), ldft(lam_ff, c + 1, ft));
            vec3 p2 = lat_lon_to_cartesian(ldft(phi_ff, c + sx, ft), ldft(lam_ff, c + sx, ft));
            vec3 v1 = { p1.x - p0.x, p1.y - p0.y, p1.z - p0.z };                    /* :23 */
            vec3 v2 = { p2.x - p0.x, p2.y - p0.y, p2.z - p0.z };                    /* :24 */
            double n1 = sqrt(v1.x * v1.x + v1.y * v1.y + v1.z * v1.z);
            double n2 = sqrt(v2.x * v2.x + v2.y * v2.y + v2.z * v2.z);
            double cs = dot3(v1, v2) / (n1 * n2);                                   /* :27 */
            size_t o = (size_t)(i - 1) + (size_t)Nx * (size_t)(j - 1);
            int imm = immersed ? immersed[o] != 0 : 0;
            double a = (imm ? DM_PIO2_HI : dm_acos(cs)) - DM_PIO2_HI;               /* :29 */
            angle[o] = a * DM_RAD2DEG;                                              /* :32 */
        }
    return 0;
}

/* convert_to_latlong_frame / convert_to_native_frame (examples/convert_to_latlong_frame.jl:12-55), for every
 * (i, j, k) of the interior, in the grid's FT:
 *   ut = deg2rad(phi_cf[i,j+1] - phi_cf[i,j]) / dy_cc[i,j];  vt = -deg2rad(phi_fc[i+1,j] - phi_fc[i,j]) / dx_cc[i,j]
 *   U = sqrt(ut^2 + vt^2); d1 = ut / U; d2 = vt / U
 *   to lat-lon: (u d1 - v d2, u d2 + v d1)      to native: (u d1 + v d2, u d2 - v d1)
 * grid arrays padded 2-D, u/v/out padded 3-D (Center, Center, Center) parents; only the interior of out is written. */
#define DEFINE_FRAME(T, SUF, SQRT, D2R)                                                                     \
static void convert_frame_##SUF(const T *phi_cf, const T *phi_fc, const T *dy_cc, const T *dx_cc,          \
                                const T *u, const T *v, T *uo, T *vo, int to_native, const dims_t *d)      \
{                                                                                                           \
    for (int k = 1; k <= d->Nz; ++k)                                                                        \
        for (int j = 1; j <= d->Ny; ++j)                                                                    \
            for (int i = 1; i <= d->Nx; ++i) {                                                              \
                size_t c2 = (size_t)(i + d->Hx - 1) + (size_t)d->sx * (size_t)(j + d->Hy - 1);              \
                size_t c3 = IDX(d, i, j, k);                                                                \
                T ut = ((phi_cf[c2 + d->sx] - phi_cf[c2]) * (T)(D2R)) / dy_cc[c2];                          \
                T vt = -((phi_fc[c2 + 1] - phi_fc[c2]) * (T)(D2R)) / dx_cc[c2];                             \
                T U = SQRT(ut * ut + vt * vt);                                                              \
                T d1 = ut / U, d2 = vt / U;                                                                 \
                T a = u[c3], b = v[c3];                                                                     \
                if (to_native) { uo[c3] = a * d1 + b * d2; vo[c3] = a * d2 - b * d1; }                      \
                else           { uo[c3] = a * d1 - b * d2; vo[c3] = a * d2 + b * d1; }                      \
            }                                                                                               \
}
DEFINE_FRAME(double, f64, sqrt, DM_DEG2RAD)
DEFINE_FRAME(float, f32, sqrtf, (float)DM_DEG2RAD)

int tpo_convert_frame(const void *phi_cf, const void *phi_fc, const void *dy_cc, const void *dx_cc,
                      const void *u, const void *v, void *uo, void *vo, int to_native,
                      int Nx, int Ny, int Nz, int Hx, int Hy, int Hz, int ft)
{
    dims_t d = mkdims(Nx, Ny, Nz, Hx, Hy, Hz);
    if (ft == 1) convert_frame_f64(phi_cf, phi_fc, dy_cc, dx_cc, u, v, uo, vo, to_native, &d);
    else         convert_frame_f32(phi_cf, phi_fc, dy_cc, dx_cc, u, v, uo, vo, to_native, &d);
    return 0;
}

/* per-j stretching table probe (sinh psi, cosh psi at Face and Center rows) */
void tpo_stretch_tables(const tpo_params *p, double *shf, double *chf, double *shc, double *chc)
{
    int Np = p->Ny;
    double *lf = dalloc(p->Nx), *lc = dalloc(p->Nx), *pf = dalloc(Np), *pc = dalloc(Np);
    tpo_tables(p, lf, lc, pf, pc);
    double a = dm_tand((90.0 - p->north_poles_latitude) / 2);
    for (int j = 0; j < Np; ++j) {
        dm_sinh_cosh(dm_asinh(dm_tand((90.0 - pf[j]) / 2) / a), &shf[j], &chf[j]);
        dm_sinh_cosh(dm_asinh(dm_tand((90.0 - pc[j]) / 2) / a), &shc[j], &chc[j]);
    }
    free(lf); free(lc); free(pf); free(pc);
}
