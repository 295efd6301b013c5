/*
 * tripolar_hip_test.h -- the test / bench-only entry points of libtripolar_hip_test.so.
 *
 * libtripolar_hip_test.so = every object of libtripolar_hip.so + these hooks + the TPG_* cross-check knobs.  It is loaded by
 * tests/, tools/ and bench.py only (tools/testlib.py); a host of the reference never loads it, and the product library
 * exports none of these symbols and reads no environment variable (tests/test_abi.py).  None of them has a reference
 * counterpart.
 *
 * Knobs (environment; read ONCE, at the first call into the test library, into an immutable record; tpg_reload_config()
 * re-reads them; every setting gives identical results, tests/test_gpu_variants.py):
 *    TPG_CELLS_VARIANT        cell kernel of tpg_build_grid: 2 LDS tile + the halo pass k_halos (default, the product's form), 3 LDS tile with
 *                             the halo cells pushed by the producing thread (k_cells_tile_push: two launches per build instead of three;
 *                             compiled into the TEST library only -- measured in round 6, not adopted: profiles/r06/build_push_ab.txt),
 *                             0 thread per cell + k_halos (cross-check)
 *    TPG_BUILD_NT             1 streaming stores in tpg_build_grid (default), 0 plain stores
 *    TPG_ZIPPER_VARIANT       3 column work items (default), 0 row work items (the fallback kernels, everywhere)
 *    TPG_FILL_FUSED           0 never / 1 always (where valid) use the fused small-field fill; 2 = always, in its one-thread-per-cell form
 *                             (k_fill_fused: the cross-check of the chunk-item form, which every geometry has since round 6)
 *    TPG_FILL_MERGED          0 never / 1 always (where valid) use the merged large-field fill
 *    TPG_EXCHANGE_IN_CAPTURE  1 lets tpg_halo_exchange_y through on a capturing stream (tools/rccl_capture_probe.py only)
 *    TPG_EXCHANGE_FAIL_STAGE  k >= 0: tpg_halo_exchange_y_pipelined* returns an injected TPG_ERR_RCCL right after the RCCL group of
 *                             stage k has been enqueued on comm_stream (error-path post-condition test); -1 / unset: off
 *    TPG_RCCL_LIBRARY         path of the library to bind INSTEAD of librccl: the test double tools/nccl_shim/libnccl_shim.so (shared-memory
 *                             mailboxes between processes that share one GPU), so that the exchange entry points run with more than one rank
 *                             on a one-GPU box; read once per process, before the first exchange call
 */
#ifndef TRIPOLAR_HIP_TEST_H
#define TRIPOLAR_HIP_TEST_H

#include "tripolar_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

int tpg_reload_config(void); /* re-read the TPG_* knobs (not for use while other threads are in the library) */

/* Same-shape copy ceiling of the fold (bench.py `zipper_copy_ceiling_ms`): the launch of tpg_zipper_fill over
 * k = 1..Nz with identical rows, bytes and work decomposition, but destination column = source column and no
 * sign -- what a pure copy of the fold's bytes costs on this device.  OVERWRITES the north halo rows (and the
 * east half of row Ny of y-Center fields) with unfolded copies. */
int tpg_zipper_copy_probe(void *const fields[], int nfields, const int8_t yloc[],
                          int Nx, int Ny, int Nz, int Hx, int Hy, int Hz, int ft, void *stream,
                          void *start_event, void *stop_event);

/* Deterministic synthetic field fill: interior (i,j,k) gets a splitmix64(seed, linear index)
 * value in (-1,1); every halo cell gets `halo_sentinel`. */
int tpg_fill_synthetic(void *field, uint64_t seed, double halo_sentinel,
                       int Nx, int Ny, int Nz, int Hx, int Hy, int Hz, int ft, void *stream);

/* Validation hook (tests/test_gpu_math.py): evaluates one of the library's deterministic Float64
 * elementary functions (which = 0 sin, 1 cos, 2 sind, 3 cosd, 4 tand, 5 atan, 6 asin, 7 asinh, 8 sinh,
 * 9 cosh, 10 acos; 20 sqrt_nr(x), 21 div_nr over pairs x = (a0, b0, a1, b1, ...): the unscaled square root and
 * division of the metric kernel) or one of the straight-line batch forms used by the metric kernel (100 sin_small, 101 cos,
 * 102 atan, 103 atan_tab, 104 atan_small, 105 asin_small, 106 sind / 107 cosd of sincosd, 108 sind / 109 cosd of the latitude form
 * sincosd_lat, 110 cos of a latitude in radians) on n device
 * doubles x -> y; rare[i] (int32) = 1 where a batch form reports "outside my fast domain".
 * These stand in for Julia Base / Distances arithmetic. */
int tpg_math_probe(int which, const void *x, void *y, void *rare, long long n, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* TRIPOLAR_HIP_TEST_H */
