/*
 * tripolar_hip.h -- C ABI of libtripolar_hip.so: MI355X (gfx950) implementation of the
 * TripolarGrid metric precompute and Zipper halo fill of CliMA/OrthogonalSphericalShellGrids.jl.
 *
 * The reference has no FFI of its own (pure Julia, multiple dispatch on Oceananigans generics,
 * SURVEY.md 8b).  Each entry point below names the reference interface it replaces
 * (file:line under the reference tree); INTEGRATION.md shows the Julia `ccall` methods a
 * maintainer would add so that Oceananigans keeps seeing TripolarGrid() /
 * ZipperBoundaryCondition / fill_halo_regions!.
 *
 * Conventions
 *  - extern "C", plain pointers and sizes only; no exceptions cross the boundary.
 *  - every call returns TPG_OK (0), a negative tpg_status, or a positive hipError_t;
 *    tpg_last_error() returns a thread-local message for the last failure on this thread.
 *  - all array memory is DEVICE memory owned by the caller (Julia GC / torch); the library
 *    never allocates or frees device memory and keeps no reference past stream completion.
 *  - calls are asynchronous on `stream` (a hipStream_t passed as void*; NULL = default stream),
 *    re-entrant, thread-safe for distinct streams, and capturable into a HIP graph -- except the RCCL seam exchange
 *    (tpg_halo_exchange_y*, tpg_fill_halo_regions_distributed* with a seam), which REFUSES a capturing stream with
 *    TPG_ERR_UNSUPPORTED instead of stalling.
 *  - array layout: column-major padded parent arrays, i fastest:
 *      2-D  A[i,j]   at  (i+Hx-1) + (Nx+2Hx) * (j+Hy-1)
 *      3-D  c[i,j,k] at  (i+Hx-1) + (Nx+2Hx) * ((j+Hy-1) + (Ny+2Hy) * (k+Hz-1))
 *    i.e. exactly the `parent` of Oceananigans' OffsetArrays.
 *  - element type selected by `ft`: TPG_F32 or TPG_F64.
 *
 *
 * The library reads no environment variable and holds no mutable global state beyond a thread-local error string, the
 * lazily bound librccl entry points (std::call_once) and a thread-local pool of ordering events for the pipelined exchange (freed with its thread).  Test / bench hooks (synthetic fill, the copy probe of the fold, the
 * elementary-function probe) and the TPG_* cross-check knobs that force the fallback kernels live in a separate
 * library, libtripolar_hip_test.so (include/tripolar_hip_test.h), which a host never loads.
 */
#ifndef TRIPOLAR_HIP_H
#define TRIPOLAR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TPG_VERSION 600 /* 0.6.0 */

enum tpg_status {
    TPG_OK = 0,
    TPG_ERR_INVALID_ARGUMENT = -1, /* null pointer, negative size, unknown ft/location ...        */
    TPG_ERR_ODD_NLAMBDA = -2,      /* ArgumentError of src/tripolar_grid.jl:81-83                 */
    TPG_ERR_BAD_PARTITION = -3,    /* row band outside 1..Ny (src/distributed_tripolar_grid.jl:28-49) */
    TPG_ERR_WORKSPACE = -4,        /* workspace missing or too small                              */
    TPG_ERR_UNSUPPORTED = -5,      /* size outside what the kernels index (see tpg_limits)        */
    TPG_ERR_NOT_NORTH = -6,        /* zipper requested on a non-north side
                                      (src/zipper_boundary_condition.jl:58-62)                    */
    TPG_ERR_RCCL = -7              /* librccl missing, or an ncclResult_t error (see tpg_last_error) */
};

enum tpg_ft { TPG_F32 = 0, TPG_F64 = 1 };
enum tpg_loc { TPG_CENTER = 0, TPG_FACE = 1 };

/* Order of the 20 horizontal arrays = positional order of src/tripolar_grid.jl:308-328
 * (z is 1-D and stays with the host glue).  Note the dy order: cc, cf, fc, ff. */
enum tpg_array {
    TPG_LAMBDA_CC = 0, TPG_LAMBDA_FC, TPG_LAMBDA_CF, TPG_LAMBDA_FF,
    TPG_PHI_CC, TPG_PHI_FC, TPG_PHI_CF, TPG_PHI_FF,
    TPG_DX_CC, TPG_DX_FC, TPG_DX_CF, TPG_DX_FF,
    TPG_DY_CC, TPG_DY_CF, TPG_DY_FC, TPG_DY_FF,
    TPG_AZ_CC, TPG_AZ_FC, TPG_AZ_CF, TPG_AZ_FF,
    TPG_NUM_ARRAYS
};

/* Keyword arguments of TripolarGrid(arch, FT; size, southernmost_latitude, halo, radius,
 * north_poles_latitude, first_pole_longitude)  (src/tripolar_grid.jl:59-66), plus the latitude
 * band of src/distributed_tripolar_grid.jl:47-49 (serial grid: jstart = 1, jend = Ny). */
typedef struct tpg_params {
    int32_t Nx, Ny, Nz;           /* size = (Nlambda, Nphi, Nz) of the GLOBAL grid                */
    int32_t Hx, Hy, Hz;           /* halo                                                         */
    double southernmost_latitude; /* default -80                                                  */
    double north_poles_latitude;  /* default  55                                                  */
    double first_pole_longitude;  /* default  70                                                  */
    double radius;                /* default R_Earth = 6371e3                                     */
    int32_t ft;                   /* enum tpg_ft: element type of the 20 output arrays            */
    int32_t jstart;               /* first global row owned by this rank (1-based)                */
    int32_t jend;                 /* last global row owned by this rank (== Ny on the north rank) */
    int32_t reserved;             /* flags: 0, or TPG_BUILD_TABLES_VALID (below); other bits are refused */
} tpg_params;

/* tpg_params.reserved flag: the workspace already holds the 1-D tables of an EARLIER tpg_build_grid call whose Nx, Ny, Hy, ft,
 * southernmost_latitude, north_poles_latitude and radius were the same (jstart / jend, Hx, Hz, Nz, first_pole_longitude may
 * differ): the table kernel (~9 us: one double-double asinh -> sinh, cosh chain per latitude row) is skipped.  For hosts that
 * build several grids of one geometry -- with_halo (src/with_halo.jl:5-44: same size, new halo in x or z),
 * reconstruct_global_grid after a band build, repeated band builds: the Python host (grids.py: TableWorkspace kept with the
 * grid) and the Julia glue (build_band) set it on exactly these paths and allocate a fresh workspace whenever the key differs.
 * The caller vouches for the workspace contents; results are identical to a build without the flag.  bench.py's timed steps
 * never set it. */
#define TPG_BUILD_TABLES_VALID 1

int tpg_version(void);
const char *tpg_last_error(void);
const char *tpg_status_string(int status);

/* ---- metric precompute ------------------------------------------------------------------
 * Replaces, in one call, src/tripolar_grid.jl:73-328: the 1-D tables (:76-97),
 * _compute_tripolar_coordinates! (src/generate_tripolar_coordinates.jl:53-89) + circshift
 * (:121-130) + coordinate halo fill (:137-199), _calculate_metrics!
 * (src/tripolar_grid_utils.jl:4-45), the 12 metric halo fills (:230-273), continue_south!
 * (:277-300, :336-357) and map(FT, .) (:308-328); with jstart/jend it also replaces the
 * per-rank slicing of src/distributed_tripolar_grid.jl:36-73 without building the globe.
 *
 * out[q] (q = enum tpg_array): device array of (Nx+2Hx) x (jend-jstart+1+2Hy) elements of `ft`;
 * local row r (0-based) holds global row jstart-Hy+r.  Three launches: the 1-D tables (skipped with TPG_BUILD_TABLES_VALID), the cell
 * kernel, the halo pass; every element of the 20 arrays is written.
 * workspace: device scratch of at least tpg_build_grid_workspace_bytes(p) bytes, 16-B aligned.
 */
size_t tpg_build_grid_workspace_bytes(const tpg_params *p);
int tpg_build_grid(const tpg_params *p, void *const out[TPG_NUM_ARRAYS],
                   void *workspace, size_t workspace_bytes, void *stream);

/* ---- zipper halo fill -------------------------------------------------------------------
 * Replaces Oceananigans' south/north halo kernel when the north BC is a ZipperBoundaryCondition:
 * _fill_north_halo!(i, k, grid, c, bc::ZBC, loc, args...) for every (i,k)
 * (src/zipper_boundary_condition.jl:146-155), i.e. fold_north_{center_center,face_center,
 * center_face,face_face}! (:70-138), dispatched on (xloc[f], yloc[f]); sign[f] is bc.condition
 * (ZipperBoundaryCondition(sign), :52).  All `nfields` fields share one geometry and are folded
 * by ONE kernel launch (batched pointer table, up to TPG_MAX_FIELDS per launch; more are split), whatever the halo width and
 * whether or not the fields are 16-B aligned (any pointer aligned to the element type is accepted).
 * Levels k = kstart .. kstart+kcount-1 (1-based, may include halo levels 1-Hz..Nz+Hz).
 * In place; halo columns i<1, i>Nx are left to the periodic pass, as in the reference.
 */
#define TPG_MAX_FIELDS 16
int tpg_zipper_fill(void *const fields[], int nfields,
                    const int8_t xloc[], const int8_t yloc[], const int32_t sign[],
                    int Nx, int Ny, int Nz, int Hx, int Hy, int Hz,
                    int kstart, int kcount, int ft, void *stream);

/* ---- profiling API (NO reference counterpart: measurement only, never needed by a host of the reference) ----
 * tpg_zipper_fill_timed / tpg_fill_halo_regions_timed / tpg_event_create / tpg_event_destroy / tpg_event_elapsed_ms.
 * Same call, with the kernel's own start / stop device timestamps recorded into two HIP events
 * (hipExtLaunchKernelGGL): what bench.py uses for roofline.achieved, so that the live number is the
 * kernel duration rocprofv3 reports, free of stream-marker and launch-boundary overhead.
 * nfields <= TPG_MAX_FIELDS (one kernel).  Events: tpg_event_create / hipEventCreate.  tpg_fill_halo_regions_timed (below)
 * is the same for the whole fill: the events ride on the FIRST kernel the call launches, which is the only one whenever the
 * fill is one merged or fused launch. */
int tpg_zipper_fill_timed(void *const fields[], int nfields,
                          const int8_t xloc[], const int8_t yloc[], const int32_t sign[],
                          int Nx, int Ny, int Nz, int Hx, int Hy, int Hz,
                          int kstart, int kcount, int ft, void *stream,
                          void *start_event, void *stop_event);
int tpg_event_create(void **event);
int tpg_event_destroy(void *event);
int tpg_event_elapsed_ms(void *start_event, void *stop_event, float *ms); /* waits for stop_event */

/* Oceananigans' periodic west/east halo fill, which fill_halo_regions! runs AFTER the zipper
 * (pinned by test/test_zipper_boundary_conditions.jl:42-45): every row and level of the parent. */
int tpg_periodic_x_fill(void *const fields[], int nfields,
                        int Nx, int Ny, int Nz, int Hx, int Hy, int Hz, int ft, void *stream);

/* fill_halo_regions!(field) on a (Periodic, RightConnected, *) tripolar field: zipper on
 * k = 1..Nz if `north_is_zipper` (serial grid, or last rank: src/distributed_tripolar_grid.jl:
 * 143-147,177-185), then periodic x.  Small fields (2-D free-surface / barotropic fields: fewer than 2^20
 * written cells per call) take ONE fused launch in which every written cell is computed from original
 * interior values through the composed index map; results are identical to the two-launch sequence.
 * Large fields with Hy <= 8 take ONE merged launch as well: column-chunk fold blocks that also
 * write the corner cells (composed map) beside periodic-x blocks for all other rows -- for EVERY halo width and every element-aligned
 * pointer: 16-B aligned chunks where Hx and Nx are whole numbers of them (the default halo 4), the same chunks stored element-aligned
 * otherwise (an odd Hx such as the halo (5, 5, 5) of examples/bickley_jet.jl:21, Float32 with Nx = 2 mod 4, 8-B aligned fields).
 * Only Hy > 8 on a large field, Nx < 2 Hx + 2 or Ny < 2 Hy + 2 run tpg_zipper_fill then tpg_periodic_x_fill (two launches). */
int tpg_fill_halo_regions(void *const fields[], int nfields,
                          const int8_t xloc[], const int8_t yloc[], const int32_t sign[],
                          int Nx, int Ny, int Nz, int Hx, int Hy, int Hz,
                          int north_is_zipper, int ft, void *stream);
/* profiling API, see tpg_zipper_fill_timed above */
int tpg_fill_halo_regions_timed(void *const fields[], int nfields,
                                const int8_t xloc[], const int8_t yloc[], const int32_t sign[],
                                int Nx, int Ny, int Nz, int Hx, int Hy, int Hz,
                                int north_is_zipper, int ft, void *stream, void *start_event, void *stop_event);

/* ---- latitude-band halo exchange helpers (config 4) -------------------------------------
 * The interior seams of a y-slab partition exchange Hy full rows (all i incl. x halos, all
 * levels incl. z halos) per side and field; the transport (RCCL send/recv, ROCm-aware MPI) stays
 * with the host or with tpg_halo_exchange_y below, as it stays with Oceananigans DistributedComputations in
 * the reference (reached from src/distributed_tripolar_grid.jl:171,195).  These two kernels gather / scatter the rows
 * between the padded 3-D fields and one contiguous message buffer of
 * nfields * (Nx+2Hx) * Hy * (Nz+2Hz) elements.
 * side: 0 = south, 1 = north.  pack reads the INTERIOR rows adjacent to that side
 * (south: j = 1..Hy, north: j = Ny-Hy+1..Ny); unpack writes the HALO rows of that side
 * (south: j = 1-Hy..0, north: j = Ny+1..Ny+Hy).
 */
size_t tpg_y_halo_buffer_elems(int nfields, int Nx, int Nz, int Hx, int Hy, int Hz);
int tpg_pack_y_halo(void *const fields[], int nfields, void *buffer, int side,
                    int Nx, int Ny, int Nz, int Hx, int Hy, int Hz, int ft, void *stream);
int tpg_unpack_y_halo(void *const fields[], int nfields, const void *buffer, int side,
                      int Nx, int Ny, int Nz, int Hx, int Hy, int Hz, int ft, void *stream);

/* ---- y-seam exchange over RCCL (config 4) ---------------------------------------------------
 * fill_halo_regions! of a Field on a DistributedTripolarGrid hands its south / north sides to Oceananigans'
 * halo communication (src/distributed_tripolar_grid.jl:171 inject_halo_communication_boundary_conditions, :195
 * FieldBoundaryBuffers; MPI Isend/Irecv of one packed buffer per side [recalled]).  Here one call issues the whole
 * seam exchange of `nfields` fields of one geometry on `stream`: ONE ncclGroupStart/ncclGroupEnd of point-to-point
 * ncclSend/ncclRecv (RCCL over xGMI), no host wait, no collective.  A stream that is being captured into a HIP graph is
 * refused (TPG_ERR_UNSUPPORTED): capture the local fill and issue the exchange eagerly.
 *   comm            ncclComm_t of the latitude-band chain (as void*): created by the host's RCCL binding, or by
 *                   tpg_comm_init_rank below (librccl is bound lazily with dlopen; TPG_ERR_RCCL if absent).
 *   rank, nranks    position in the chain: rank 0 is the southernmost band and has no south seam, rank nranks-1
 *                   owns the zipper and has no north seam (src/distributed_tripolar_grid.jl:75,143-147).
 *   send_* / recv_* message buffers of tpg_y_halo_buffer_elems(...) elements each (sides without a peer may be
 *                   NULL): PACKED exchange = tpg_pack_y_halo -> one message per direction -> tpg_unpack_y_halo.
 *                   All four NULL: PACK-FREE exchange -- the Hy seam rows of one (field, level) are one contiguous
 *                   window of the parent array (Hy * (Nx+2Hx) elements), sent from / received into the fields
 *                   directly, (nfields * (Nz+2Hz)) send/recv pairs per direction inside the one group.  Every pair is
 *                   one RCCL operation (~3.4 us each on the loop-back): the pack-free form is for 2-D and few-level
 *                   fields only; 3-D fields use the packed form (one operation per direction).
 * Call it after the zipper (north rank) and the periodic-x pass of the same fill, as the reference orders them.
 * tpg_halo_exchange_y_peers is the same with explicit peer ranks (-1 = no seam on that side). */
#define TPG_COMM_ID_BYTES 128
int tpg_comm_available(void);                                                    /* TPG_OK if librccl could be bound (no collective call):
                                                                                    agree on this across ranks BEFORE tpg_comm_init_rank */
int tpg_comm_unique_id(void *id128);                                             /* ncclGetUniqueId     */
int tpg_comm_init_rank(void **comm, int nranks, const void *id128, int rank);    /* ncclCommInitRank on the current device */
int tpg_comm_destroy(void *comm);                                                /* ncclCommDestroy     */
int tpg_halo_exchange_y(void *comm, int rank, int nranks, void *const fields[], int nfields,
                        void *send_south, void *send_north, void *recv_south, void *recv_north,
                        int Nx, int Ny, int Nz, int Hx, int Hy, int Hz, int ft, void *stream);
int tpg_halo_exchange_y_peers(void *comm, int south_peer, int north_peer, void *const fields[], int nfields,
                              void *send_south, void *send_north, void *recv_south, void *recv_north,
                              int Nx, int Ny, int Nz, int Hx, int Hy, int Hz, int ft, void *stream);

/* The packed exchange as a PIPELINE over stages of `fields_per_stage` fields (0 = 1; the message layout is
 * [field][level][Hy][Nx+2Hx], so a stage is one contiguous slice of each message buffer):
 *     stream       pack(0) pack(1) .. pack(S-1)            unpack(0)          unpack(1)  ..  unpack(S-1)
 *     comm_stream          group(0)           group(1)  ..            group(S-1)
 * group(s) = one ncclGroupStart/End with the sends / receives of stage s.  The link starts after ONE stage is packed, the
 * other pack kernels run beside the first transfer and every unpack but the last beside the next transfer -- what the
 * reference's per-field fill_halo_regions! gets from MPI Isend/Irecv progressing behind the next field's pack
 * (src/distributed_tripolar_grid.jl:171,195 [Oceananigans' transport, recalled]).  Delivers exactly what the monolithic
 * form delivers (same pack / unpack kernels on slices).  On return -- WITH ANY STATUS: after a failure that follows the first
 * group the two streams are joined before the status goes back -- `stream` is ordered after every transfer and unpack that was
 * enqueued, and comm_stream holds no work `stream` does not wait for; the message buffers may be reused by the next call.
 * comm_stream: a second hipStream_t of the caller on the same device (NULL or == stream: the same stages on one stream, no
 * overlap).  All four message buffers are required for every side with a peer (no pack-free form).
 * ALL RANKS MUST AGREE on nfields and fields_per_stage: group(k) of one rank pairs with group(k) of its neighbour and the two
 * must carry equal element counts (a mismatch is an RCCL size error or a stall, not detectable from one rank); the Python host
 * (HaloFillPlan) and bench.py agree on the value once, collectively, before the first exchange.
 * The ordering events come from a pool kept per host thread and device: created at that thread's first call, destroyed when the
 * thread ends. */
int tpg_halo_exchange_y_pipelined(void *comm, int rank, int nranks, void *const fields[], int nfields,
                                  void *send_south, void *send_north, void *recv_south, void *recv_north,
                                  int Nx, int Ny, int Nz, int Hx, int Hy, int Hz, int ft,
                                  void *stream, void *comm_stream, int fields_per_stage);
int tpg_halo_exchange_y_pipelined_peers(void *comm, int south_peer, int north_peer, void *const fields[], int nfields,
                                        void *send_south, void *send_north, void *recv_south, void *recv_north,
                                        int Nx, int Ny, int Nz, int Hx, int Hy, int Hz, int ft,
                                        void *stream, void *comm_stream, int fields_per_stage);

/* fill_halo_regions!(fields...) on a DistributedTripolarGrid, whole, in ONE call and in the reference's order
 * (src/distributed_tripolar_grid.jl:143-147,177-185: the zipper only on the last rank; src/distributed_tripolar_grid.jl:171,195: every
 * other south / north side is neighbour communication): zipper fold (rank nranks-1) -> periodic x (merged / fused launch where
 * the geometry allows) -> tpg_halo_exchange_y, all enqueued on `stream`.  nranks = 1 is the serial fill (comm may be NULL).
 * xloc / yloc / sign are read on the zipper rank only.  Buffers as for tpg_halo_exchange_y (all NULL = pack-free).
 * The _peers form takes explicit neighbours (-1 = none) and the north side's kind, for hosts with their own rank map. */
int tpg_fill_halo_regions_distributed(void *comm, int rank, int nranks, void *const fields[], int nfields,
                                      const int8_t xloc[], const int8_t yloc[], const int32_t sign[],
                                      void *send_south, void *send_north, void *recv_south, void *recv_north,
                                      int Nx, int Ny, int Nz, int Hx, int Hy, int Hz, int ft, void *stream);
int tpg_fill_halo_regions_distributed_peers(void *comm, int south_peer, int north_peer, int north_is_zipper,
                                            void *const fields[], int nfields,
                                            const int8_t xloc[], const int8_t yloc[], const int32_t sign[],
                                            void *send_south, void *send_north, void *recv_south, void *recv_north,
                                            int Nx, int Ny, int Nz, int Hx, int Hy, int Hz, int ft, void *stream);
/* The same whole fill with the seam exchange in its pipelined form (tpg_halo_exchange_y_pipelined above). */
int tpg_fill_halo_regions_distributed_pipelined(void *comm, int rank, int nranks, void *const fields[], int nfields,
                                                const int8_t xloc[], const int8_t yloc[], const int32_t sign[],
                                                void *send_south, void *send_north, void *recv_south, void *recv_north,
                                                int Nx, int Ny, int Nz, int Hx, int Hy, int Hz, int ft,
                                                void *stream, void *comm_stream, int fields_per_stage);
int tpg_fill_halo_regions_distributed_pipelined_peers(void *comm, int south_peer, int north_peer, int north_is_zipper,
                                                      void *const fields[], int nfields,
                                                      const int8_t xloc[], const int8_t yloc[], const int32_t sign[],
                                                      void *send_south, void *send_north, void *recv_south, void *recv_north,
                                                      int Nx, int Ny, int Nz, int Hx, int Hy, int Hz, int ft,
                                                      void *stream, void *comm_stream, int fields_per_stage);

/* ---- geometry utilities over the grid arrays -------------------------------
 * tpg_nonorthogonality_angle: compute_nonorthogonality_angle! of test/test_tripolar_grid.jl:8-34 as launched at
 * :70 over (Nx-1, Ny-1): angle[i,j] = rad2deg(acos(v1.v2 / (|v1||v2|)) - pi/2) with v1, v2 the chords from the
 * Face-Face node (i,j) to (i+1,j) and (i,j+1) on the unit sphere; 0 where immersed[i,j] != 0 and for i = Nx or
 * j = Ny.  lambda_ff / phi_ff: padded 2-D grid arrays (type `ft`); immersed: dense Nx x Ny bytes or NULL;
 * angle: dense Nx x Ny Float64, i fastest (the reference's zeros(size(grid)...)).
 * tpg_convert_frame: convert_to_latlong_frame (to_native = 0) / convert_to_native_frame (to_native = 1) of
 * examples/convert_to_latlong_frame.jl:12-55 for every interior (i, j, k): rotation of (u, v) by the local
 * direction cosines d1, d2 derived from phi_cf, phi_fc, dy_cc, dx_cc.  u, v, u_out, v_out: padded 3-D
 * (Center, Center, Center) parents (only the interior of the outputs is written; outputs may alias nothing). */
int tpg_nonorthogonality_angle(const void *lambda_ff, const void *phi_ff, const uint8_t *immersed, double *angle,
                               int Nx, int Ny, int Hx, int Hy, int ft, void *stream);
int tpg_convert_frame(const void *phi_cf, const void *phi_fc, const void *dy_cc, const void *dx_cc,
                      const void *u, const void *v, void *u_out, void *v_out, int to_native,
                      int Nx, int Ny, int Nz, int Hx, int Hy, int Hz, int ft, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* TRIPOLAR_HIP_H */
