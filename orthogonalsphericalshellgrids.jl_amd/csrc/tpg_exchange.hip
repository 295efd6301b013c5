// tpg_exchange.hip -- y-seam halo exchange of a latitude-band TripolarGrid over RCCL (xGMI), behind the C ABI.
//
// Replaces what a Field on a DistributedTripolarGrid gets from Oceananigans' DistributedComputations
// (inject_halo_communication_boundary_conditions / FieldBoundaryBuffers, reached from
// src/distributed_tripolar_grid.jl:171,195): per fill, each interior seam swaps Hy full rows (all i incl. the x
// halos, all levels incl. the z halos) in both directions.  The reference's transport is MPI Isend/Irecv of one
// packed buffer per side [recalled]; here it is ONE ncclGroupStart/ncclGroupEnd of point-to-point
// ncclSend/ncclRecv on the caller's stream: no host wait, no collective (a y-slab chain only talks to its two
// neighbours).  The exchange refuses a stream that is being captured into a HIP graph (TPG_ERR_UNSUPPORTED).  That fence is a
// PRECAUTION for multi-rank first use (the first group towards a peer sets up the connection: allocations, IPC handles, a
// proxy thread), which no box available to the build could exercise; on a single-rank communicator capture was shown to work
// (five configurations complete and replay bit-exact: tools/rccl_capture_probe.py, profiles/r03/capture_probe/*.log), and the
// stall reported in round 2 never reproduced -- its cause is unknown, the fence is not presented as its fix.
// Two message shapes:
//   packed    : tpg_pack_y_halo -> one message per seam direction ([field][level][Hy][sx], 9.58 MB per field at
//               1/10 deg x 75 levels) -> tpg_unpack_y_halo;
//   pack-free : the Hy seam rows of one (field, level) are already one contiguous window of the parent array
//               (Hy * sx elements = 115 KB at 1/10 deg), so each window is sent from / received into the field
//               itself: no staging buffers, no pack / unpack kernels, (fields x levels) send/recv pairs per
//               direction inside the one group.
// librccl is bound at first use with dlopen (no link-time dependency: a host that never exchanges -- serial grids,
// this container -- needs no RCCL).  xGMI is point-to-point (7 links x ~153 GB/s per GPU): one seam direction of a
// 4-field fill is 38.3 MB, i.e. >= 0.25 ms on one link; the two directions use the link's two directions.
#include "tpg_common.hpp"
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <string.h>
#include <mutex>

namespace {

struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    char why[256] = "";
};

Rccl g_rccl;
std::once_flag g_rccl_once;

void load_rccl()
{
    const char* names[] = { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so" };
    const char* forced = tpg::config().rccl_library;             // test library only (TPG_RCCL_LIBRARY): the test double of tools/nccl_shim
    if (forced) g_rccl.handle = dlopen(forced, RTLD_NOW | RTLD_LOCAL);
    for (const char* n : names) {
        if (g_rccl.handle || forced) break;
        g_rccl.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    }
    if (!g_rccl.handle) { snprintf(g_rccl.why, sizeof g_rccl.why, "librccl not found: %s", dlerror()); return; }
    bool ok = true;
#define BIND(field, sym)                                                                     \
    do {                                                                                     \
        *reinterpret_cast<void**>(&g_rccl.field) = dlsym(g_rccl.handle, sym);               \
        if (!g_rccl.field) { snprintf(g_rccl.why, sizeof g_rccl.why, "librccl lacks %s", sym); ok = false; } \
    } while (0)
    BIND(GetUniqueId, "ncclGetUniqueId"); BIND(CommInitRank, "ncclCommInitRank"); BIND(CommDestroy, "ncclCommDestroy");
    BIND(CommCount, "ncclCommCount"); BIND(CommUserRank, "ncclCommUserRank");
    BIND(Send, "ncclSend"); BIND(Recv, "ncclRecv"); BIND(GroupStart, "ncclGroupStart"); BIND(GroupEnd, "ncclGroupEnd");
    BIND(GetErrorString, "ncclGetErrorString");
#undef BIND
    if (!ok) { dlclose(g_rccl.handle); g_rccl.handle = nullptr; }
}

const Rccl* rccl()
{
    std::call_once(g_rccl_once, load_rccl);
    if (!g_rccl.handle) { tpg::set_error("%s", g_rccl.why); return nullptr; }
    return &g_rccl;
}

int nccl_status(const Rccl* r, ncclResult_t e, const char* what)
{
    if (e == ncclSuccess) return TPG_OK;
    tpg::set_error("%s: %s (ncclResult_t %d)", what, r->GetErrorString(e), (int)e);
    return TPG_ERR_RCCL;
}

// Ordering events of the pipelined exchange: one small pool per (host thread, device), owned by a thread_local object -- the events are
// destroyed when their thread ends (or, for the main thread, when the process exits: thread_local destructors run before the runtime's own
// exit handlers).  2 x TPG_MAX_FIELDS stage events + 1 join event for the error path.
struct EventPools {
    static constexpr int kMaxDevices = 64, kEvents = 2 * TPG_MAX_FIELDS + 1;
    struct Pool { hipEvent_t ev[kEvents]; int n = 0; };
    Pool* pools[kMaxDevices] = {};
    int get(int device, Pool** out)
    {
        if (device < 0 || device >= kMaxDevices) { tpg::set_error("device ordinal %d outside the event pool", device); return TPG_ERR_UNSUPPORTED; }
        if (!pools[device]) pools[device] = new Pool;
        Pool& p = *pools[device];
        while (p.n < kEvents) {                                    // once per thread and device
            const hipError_t e = hipEventCreateWithFlags(&p.ev[p.n], hipEventDisableTiming);
            if (e != hipSuccess) return tpg::hip_status(e, "hipEventCreateWithFlags");
            ++p.n;
        }
        *out = &p;
        return TPG_OK;
    }
    ~EventPools()
    {
        for (Pool*& p : pools) {
            if (!p) continue;
            for (int i = 0; i < p->n; ++i) (void)hipEventDestroy(p->ev[i]);
            delete p;
            p = nullptr;
        }
    }
};
thread_local EventPools t_event_pools;

}  // namespace

extern "C" {

int tpg_comm_available(void)
{
    return rccl() ? TPG_OK : TPG_ERR_RCCL;
}

int tpg_comm_unique_id(void* id128)
{
    if (!id128) { tpg::set_error("null id buffer"); return TPG_ERR_INVALID_ARGUMENT; }
    const Rccl* r = rccl();
    if (!r) return TPG_ERR_RCCL;
    static_assert(sizeof(ncclUniqueId) == TPG_COMM_ID_BYTES, "ncclUniqueId size");
    return nccl_status(r, r->GetUniqueId(static_cast<ncclUniqueId*>(id128)), "ncclGetUniqueId");
}

int tpg_comm_init_rank(void** comm, int nranks, const void* id128, int rank)
{
    if (!comm || !id128 || nranks < 1 || rank < 0 || rank >= nranks) { tpg::set_error("bad communicator arguments"); return TPG_ERR_INVALID_ARGUMENT; }
    const Rccl* r = rccl();
    if (!r) return TPG_ERR_RCCL;
    ncclUniqueId id;
    memcpy(&id, id128, sizeof id);
    ncclComm_t c = nullptr;
    int rc = nccl_status(r, r->CommInitRank(&c, nranks, id, rank), "ncclCommInitRank");
    *comm = rc ? nullptr : c;
    return rc;
}

int tpg_comm_destroy(void* comm)
{
    if (!comm) return TPG_OK;
    const Rccl* r = rccl();
    if (!r) return TPG_ERR_RCCL;
    return nccl_status(r, r->CommDestroy(static_cast<ncclComm_t>(comm)), "ncclCommDestroy");
}

// one stream's capture status: TPG_OK if the exchange may be enqueued on it
static int capture_fence(hipStream_t s)
{
    // capture fence: a precaution, not a known failure -- single-rank capture works (profiles/r03/capture_probe/), but a multi-rank
    // group's first use of a peer does connection setup that nothing here could test inside a capture; a host
    // that wants graphs captures the local part of the fill and issues the exchange eagerly between replays
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    const hipError_t ce = hipStreamIsCapturing(s, &cs);
    if (ce != hipSuccess) (void)hipGetLastError();               // e.g. the legacy stream while another stream captures in global mode
    if ((ce != hipSuccess || cs != hipStreamCaptureStatusNone) && !tpg::config().exchange_in_capture) {
        tpg::set_error("tpg_halo_exchange_y: the stream is being captured into a HIP graph; the RCCL seam exchange is not offered inside a "
                       "capture (untested with more than one rank: capture the local fill, issue the exchange eagerly)");
        return TPG_ERR_UNSUPPORTED;
    }
    return TPG_OK;
}

// argument checks shared by the monolithic and the pipelined exchange; *nothing_to_do = 1 for Hy = 0 or a chain without seams
static int exchange_checks(void* comm, int south_peer, int north_peer, void* const fields[], int nfields,
                           void* send_south, void* send_north, void* recv_south, void* recv_north,
                           int Nx, int Ny, int Nz, int Hx, int Hy, int Hz, int ft, bool need_packed, int* nothing_to_do)
{
    *nothing_to_do = 0;
    int rc = tpg::check_geom(Nx, Ny, Nz, Hx, Hy, Hz, ft);
    if (rc) return rc;
    if (!comm) { tpg::set_error("null communicator"); return TPG_ERR_INVALID_ARGUMENT; }
    if (!fields || nfields < 1) { tpg::set_error("no fields"); return TPG_ERR_INVALID_ARGUMENT; }
    for (int f = 0; f < nfields; ++f) if (!fields[f]) { tpg::set_error("null field %d", f); return TPG_ERR_INVALID_ARGUMENT; }
    if (nfields > TPG_MAX_FIELDS) { tpg::set_error("at most %d fields per exchange", TPG_MAX_FIELDS); return TPG_ERR_UNSUPPORTED; }
    if (Hy == 0 || (south_peer < 0 && north_peer < 0)) { *nothing_to_do = 1; return TPG_OK; }
    const bool packed = send_south || send_north || recv_south || recv_north;
    if ((packed || need_packed) && ((south_peer >= 0 && (!send_south || !recv_south)) || (north_peer >= 0 && (!send_north || !recv_north)))) {
        tpg::set_error("packed exchange: a message buffer is missing for a side that has a peer");
        return TPG_ERR_INVALID_ARGUMENT;
    }
    return TPG_OK;
}

int tpg_halo_exchange_y_peers(void* comm, int south_peer, int north_peer, void* const fields[], int nfields,
                              void* send_south, void* send_north, void* recv_south, void* recv_north,
                              int Nx, int Ny, int Nz, int Hx, int Hy, int Hz, int ft, void* stream)
{
    int nothing = 0;
    int rc = exchange_checks(comm, south_peer, north_peer, fields, nfields, send_south, send_north, recv_south, recv_north,
                             Nx, Ny, Nz, Hx, Hy, Hz, ft, false, &nothing);
    if (rc || nothing) return rc;
    const bool packed = send_south || send_north || recv_south || recv_north;
    const Rccl* r = rccl();
    if (!r) return TPG_ERR_RCCL;
    ncclComm_t c = static_cast<ncclComm_t>(comm);
    hipStream_t s = tpg::as_stream(stream);
    if ((rc = capture_fence(s))) return rc;
    const ncclDataType_t dt = ft == TPG_F64 ? ncclFloat64 : ncclFloat32;
    const size_t esz = ft == TPG_F64 ? 8 : 4;
    const size_t sx = (size_t)Nx + 2 * Hx, sy = (size_t)Ny + 2 * Hy, nlev = (size_t)Nz + 2 * Hz;
    const size_t window = (size_t)Hy * sx;                      // elements of one (field, level) seam window
    const size_t msg = tpg_y_halo_buffer_elems(nfields, Nx, Nz, Hx, Hy, Hz);

    if (packed) {
        if (north_peer >= 0 && (rc = tpg_pack_y_halo(fields, nfields, send_north, 1, Nx, Ny, Nz, Hx, Hy, Hz, ft, stream))) return rc;
        if (south_peer >= 0 && (rc = tpg_pack_y_halo(fields, nfields, send_south, 0, Nx, Ny, Nz, Hx, Hy, Hz, ft, stream))) return rc;
    }
    // Order inside the group: sends north-then-south, receives from-south-then-from-north, so that a rank whose two
    // peers are the same rank (a ring of two, or the single-rank loop-back of the tests) pairs "sent north" with
    // "received from the south".
    if ((rc = nccl_status(r, r->GroupStart(), "ncclGroupStart"))) return rc;
    ncclResult_t e = ncclSuccess;
    auto send_side = [&](int peer, void* buffer, size_t row0) {
        if (peer < 0 || e != ncclSuccess) return;
        if (packed) { e = r->Send(buffer, msg, dt, peer, c, s); return; }
        for (int f = 0; f < nfields && e == ncclSuccess; ++f)
            for (size_t l = 0; l < nlev && e == ncclSuccess; ++l)
                e = r->Send(static_cast<const char*>(fields[f]) + (sx * sy * l + sx * row0) * esz, window, dt, peer, c, s);
    };
    auto recv_side = [&](int peer, void* buffer, size_t row0) {
        if (peer < 0 || e != ncclSuccess) return;
        if (packed) { e = r->Recv(buffer, msg, dt, peer, c, s); return; }
        for (int f = 0; f < nfields && e == ncclSuccess; ++f)
            for (size_t l = 0; l < nlev && e == ncclSuccess; ++l)
                e = r->Recv(static_cast<char*>(fields[f]) + (sx * sy * l + sx * row0) * esz, window, dt, peer, c, s);
    };
    send_side(north_peer, send_north, (size_t)Ny);               // interior rows j = Ny-Hy+1..Ny  (parent rows Ny..Ny+Hy-1)
    send_side(south_peer, send_south, (size_t)Hy);               // interior rows j = 1..Hy
    recv_side(south_peer, recv_south, 0);                        // halo rows j = 1-Hy..0
    recv_side(north_peer, recv_north, (size_t)Ny + Hy);          // halo rows j = Ny+1..Ny+Hy
    ncclResult_t e2 = r->GroupEnd();
    if ((rc = nccl_status(r, e, "ncclSend/ncclRecv"))) return rc;
    if ((rc = nccl_status(r, e2, "ncclGroupEnd"))) return rc;
    if (packed) {
        if (south_peer >= 0 && (rc = tpg_unpack_y_halo(fields, nfields, recv_south, 0, Nx, Ny, Nz, Hx, Hy, Hz, ft, stream))) return rc;
        if (north_peer >= 0 && (rc = tpg_unpack_y_halo(fields, nfields, recv_north, 1, Nx, Ny, Nz, Hx, Hy, Hz, ft, stream))) return rc;
    }
    return TPG_OK;
}

// The packed exchange as a PIPELINE of stages of `fields_per_stage` fields (the message layout is [field][level][Hy][sx], so a
// stage is one contiguous slice of each message buffer):
//     stream       pack(0) pack(1) ... pack(S-1)          unpack(0)         unpack(1)  ...  unpack(S-1)
//     comm_stream          group(0)          group(1) ...            group(S-1)
// group(s) = one ncclGroupStart/End holding the sends / receives of stage s; it waits (event) for pack(s) and unpack(s) waits
// (event) for it, so the link starts after ONE stage has been packed instead of all of them, the remaining pack kernels run
// beside the first transfer and every unpack but the last runs beside the next stage's transfer.  At 1/10 deg x 75 levels, 4
// fields, 8 bands: 2 x 13 us of pack + 2 x 13 us of unpack against >= 250 us on the link, of which 3/4 can hide.  What is
// delivered is bit-identical to the monolithic form (same pack / unpack kernels on slices).  When the function returns, `stream`
// is ordered after every transfer and unpack (its last wait), and comm_stream holds no work that `stream` does not wait for: the
// message buffers may be reused by the next call on the same pair of streams -- on error returns as well (a failure after the first
// group joins the two streams before the status goes back).  comm_stream = NULL (or = stream) runs the same stages on the one stream
// (no overlap).  The events that order the two streams come from a thread-local pool (timing disabled), destroyed with its thread.
// EVERY RANK OF THE CHAIN MUST PASS THE SAME fields_per_stage (and nfields): group(k) here pairs with group(k) on the neighbour, with
// equal element counts; different values give mismatched ncclSend / ncclRecv sizes or a stall.
int tpg_halo_exchange_y_pipelined_peers(void* comm, int south_peer, int north_peer, void* const fields[], int nfields,
                                        void* send_south, void* send_north, void* recv_south, void* recv_north,
                                        int Nx, int Ny, int Nz, int Hx, int Hy, int Hz, int ft,
                                        void* stream, void* comm_stream, int fields_per_stage)
{
    int nothing = 0;
    int rc = exchange_checks(comm, south_peer, north_peer, fields, nfields, send_south, send_north, recv_south, recv_north,
                             Nx, Ny, Nz, Hx, Hy, Hz, ft, true, &nothing);
    if (rc || nothing) return rc;
    if (fields_per_stage < 0) { tpg::set_error("fields_per_stage %d < 0", fields_per_stage); return TPG_ERR_INVALID_ARGUMENT; }
    const int fps = fields_per_stage == 0 ? 1 : (fields_per_stage > nfields ? nfields : fields_per_stage);
    const int nstages = (nfields + fps - 1) / fps;
    const Rccl* r = rccl();
    if (!r) return TPG_ERR_RCCL;
    ncclComm_t c = static_cast<ncclComm_t>(comm);
    hipStream_t s = tpg::as_stream(stream);
    hipStream_t cs = comm_stream ? tpg::as_stream(comm_stream) : s;
    const bool two = cs != s;
    if ((rc = capture_fence(s))) return rc;
    if (two && (rc = capture_fence(cs))) return rc;
    const ncclDataType_t dt = ft == TPG_F64 ? ncclFloat64 : ncclFloat32;
    const size_t esz = ft == TPG_F64 ? 8 : 4;
    const size_t per_field = tpg_y_halo_buffer_elems(1, Nx, Nz, Hx, Hy, Hz);
    auto slice = [&](void* base, int f0) -> void* { return base ? static_cast<char*>(base) + (size_t)f0 * per_field * esz : nullptr; };

    // ordering events: a pool (timing disabled) per host thread and DEVICE -- an event may only be recorded on a stream of the device it was
    // created on, and one host thread may drive several devices -- created at a thread's first pipelined exchange on that device and owned
    // by a thread_local object whose destructor destroys them when the thread ends (creating and destroying 2 x stages events per fill cost
    // more host time than the RCCL groups themselves).  Re-recording an event does not disturb a wait enqueued on it earlier (the wait took
    // the state it had then).
    hipEvent_t *packed_ev = nullptr, *moved_ev = nullptr, join_ev = nullptr;
    if (two) {
        int device = 0;
        const hipError_t de = hipGetDevice(&device);
        if (de != hipSuccess) return tpg::hip_status(de, "hipGetDevice");
        EventPools::Pool* pool = nullptr;
        if ((rc = t_event_pools.get(device, &pool))) return rc;
        packed_ev = pool->ev; moved_ev = pool->ev + TPG_MAX_FIELDS; join_ev = pool->ev[2 * TPG_MAX_FIELDS];
    }
    // Post-condition on EVERY return, error returns included: `stream` is ordered after everything this call has put on comm_stream.
    // `groups_on_cs` counts the RCCL groups enqueued there so far; a failure after the first one joins the two streams (one event on
    // comm_stream, one wait on `stream`; their own status is ignored -- the first error is the one reported) before the status goes back.
    int groups_on_cs = 0;
    auto leave = [&](int status) {
        if (status != TPG_OK && two && groups_on_cs > 0) {
            if (hipEventRecord(join_ev, cs) == hipSuccess) (void)hipStreamWaitEvent(s, join_ev, 0);
            (void)hipGetLastError();
        }
        return status;
    };
#define TPG_PIPE_CHECK(expr) do { if ((rc = (expr))) return leave(rc); } while (0)
    // every pack first: they depend on nothing but the local fill that precedes the call on `stream`
    for (int k = 0; k < nstages; ++k) {
        const int f0 = k * fps, n = nfields - f0 < fps ? nfields - f0 : fps;
        if (north_peer >= 0) TPG_PIPE_CHECK(tpg_pack_y_halo(fields + f0, n, slice(send_north, f0), 1, Nx, Ny, Nz, Hx, Hy, Hz, ft, stream));
        if (south_peer >= 0) TPG_PIPE_CHECK(tpg_pack_y_halo(fields + f0, n, slice(send_south, f0), 0, Nx, Ny, Nz, Hx, Hy, Hz, ft, stream));
        if (two) TPG_PIPE_CHECK(tpg::hip_status(hipEventRecord(packed_ev[k], s), "hipEventRecord"));
    }
    for (int k = 0; k < nstages; ++k) {
        const int f0 = k * fps, n = nfields - f0 < fps ? nfields - f0 : fps;
        const size_t msg = (size_t)n * per_field;
        if (two) TPG_PIPE_CHECK(tpg::hip_status(hipStreamWaitEvent(cs, packed_ev[k], 0), "hipStreamWaitEvent"));
        // same order inside every group as in the monolithic form (a rank whose two peers are one rank pairs "sent north" with
        // "received from the south"); groups towards one peer are matched in issue order, the same on both ends
        TPG_PIPE_CHECK(nccl_status(r, r->GroupStart(), "ncclGroupStart"));
        ncclResult_t e = ncclSuccess;
        if (north_peer >= 0 && e == ncclSuccess) e = r->Send(slice(send_north, f0), msg, dt, north_peer, c, cs);
        if (south_peer >= 0 && e == ncclSuccess) e = r->Send(slice(send_south, f0), msg, dt, south_peer, c, cs);
        if (south_peer >= 0 && e == ncclSuccess) e = r->Recv(slice(recv_south, f0), msg, dt, south_peer, c, cs);
        if (north_peer >= 0 && e == ncclSuccess) e = r->Recv(slice(recv_north, f0), msg, dt, north_peer, c, cs);
        ncclResult_t e2 = r->GroupEnd();
        ++groups_on_cs;                                              // whatever GroupEnd says, part of the group may be on comm_stream
        TPG_PIPE_CHECK(nccl_status(r, e, "ncclSend/ncclRecv"));
        TPG_PIPE_CHECK(nccl_status(r, e2, "ncclGroupEnd"));
        if (tpg::config().exchange_fail_stage == k) {                // test library only (TPG_EXCHANGE_FAIL_STAGE); -1 in the product
            tpg::set_error("injected failure after the RCCL group of stage %d (TPG_EXCHANGE_FAIL_STAGE)", k);
            return leave(TPG_ERR_RCCL);
        }
        if (two) {
            TPG_PIPE_CHECK(tpg::hip_status(hipEventRecord(moved_ev[k], cs), "hipEventRecord"));
            TPG_PIPE_CHECK(tpg::hip_status(hipStreamWaitEvent(s, moved_ev[k], 0), "hipStreamWaitEvent"));
        }
        if (south_peer >= 0) TPG_PIPE_CHECK(tpg_unpack_y_halo(fields + f0, n, slice(recv_south, f0), 0, Nx, Ny, Nz, Hx, Hy, Hz, ft, stream));
        if (north_peer >= 0) TPG_PIPE_CHECK(tpg_unpack_y_halo(fields + f0, n, slice(recv_north, f0), 1, Nx, Ny, Nz, Hx, Hy, Hz, ft, stream));
    }
#undef TPG_PIPE_CHECK
    return TPG_OK;
}

int tpg_halo_exchange_y_pipelined(void* comm, int rank, int nranks, void* const fields[], int nfields,
                                  void* send_south, void* send_north, void* recv_south, void* recv_north,
                                  int Nx, int Ny, int Nz, int Hx, int Hy, int Hz, int ft,
                                  void* stream, void* comm_stream, int fields_per_stage)
{
    if (nranks < 1 || rank < 0 || rank >= nranks) { tpg::set_error("rank %d outside 0:%d", rank, nranks - 1); return TPG_ERR_BAD_PARTITION; }
    return tpg_halo_exchange_y_pipelined_peers(comm, rank > 0 ? rank - 1 : -1, rank < nranks - 1 ? rank + 1 : -1, fields, nfields,
                                               send_south, send_north, recv_south, recv_north, Nx, Ny, Nz, Hx, Hy, Hz, ft,
                                               stream, comm_stream, fields_per_stage);
}

int tpg_halo_exchange_y(void* comm, int rank, int nranks, void* const fields[], int nfields,
                        void* send_south, void* send_north, void* recv_south, void* recv_north,
                        int Nx, int Ny, int Nz, int Hx, int Hy, int Hz, int ft, void* stream)
{
    if (nranks < 1 || rank < 0 || rank >= nranks) { tpg::set_error("rank %d outside 0:%d", rank, nranks - 1); return TPG_ERR_BAD_PARTITION; }
    // rank 0 is the southernmost band, rank nranks-1 owns the zipper: neither of those two sides communicates
    return tpg_halo_exchange_y_peers(comm, rank > 0 ? rank - 1 : -1, rank < nranks - 1 ? rank + 1 : -1, fields, nfields,
                                     send_south, send_north, recv_south, recv_north, Nx, Ny, Nz, Hx, Hy, Hz, ft, stream);
}

// fill_halo_regions! of fields on a DistributedTripolarGrid, whole, in the reference's order
// (src/distributed_tripolar_grid.jl:143-147,177-185: the zipper only where the north side is the fold, i.e. on the last
// rank; every other south / north side is neighbour communication): zipper -> periodic x (one merged or fused launch where
// the geometry allows) -> the seam exchange, all enqueued on `stream`.
int tpg_fill_halo_regions_distributed_peers(void* comm, int south_peer, int north_peer, int north_is_zipper,
                                            void* const fields[], int nfields,
                                            const int8_t xloc[], const int8_t yloc[], const int32_t sign[],
                                            void* send_south, void* send_north, void* recv_south, void* recv_north,
                                            int Nx, int Ny, int Nz, int Hx, int Hy, int Hz, int ft, void* stream)
{
    if (north_is_zipper && north_peer >= 0) {
        tpg::set_error("the north side is either the zipper or a seam, not both (north_peer %d)", north_peer);
        return TPG_ERR_INVALID_ARGUMENT;
    }
    if (nfields > TPG_MAX_FIELDS && (south_peer >= 0 || north_peer >= 0)) {
        tpg::set_error("at most %d fields per distributed fill (one seam message per side)", TPG_MAX_FIELDS);
        return TPG_ERR_UNSUPPORTED;
    }
    int rc = tpg_fill_halo_regions(fields, nfields, xloc, yloc, sign, Nx, Ny, Nz, Hx, Hy, Hz, north_is_zipper ? 1 : 0, ft, stream);
    if (rc) return rc;
    if (south_peer < 0 && north_peer < 0) return TPG_OK;           // a one-band chain: the serial fill
    return tpg_halo_exchange_y_peers(comm, south_peer, north_peer, fields, nfields, send_south, send_north, recv_south, recv_north,
                                     Nx, Ny, Nz, Hx, Hy, Hz, ft, stream);
}

int tpg_fill_halo_regions_distributed(void* comm, int rank, int nranks, void* const fields[], int nfields,
                                      const int8_t xloc[], const int8_t yloc[], const int32_t sign[],
                                      void* send_south, void* send_north, void* recv_south, void* recv_north,
                                      int Nx, int Ny, int Nz, int Hx, int Hy, int Hz, int ft, void* stream)
{
    if (nranks < 1 || rank < 0 || rank >= nranks) { tpg::set_error("rank %d outside 0:%d", rank, nranks - 1); return TPG_ERR_BAD_PARTITION; }
    return tpg_fill_halo_regions_distributed_peers(comm, rank > 0 ? rank - 1 : -1, rank < nranks - 1 ? rank + 1 : -1, rank == nranks - 1,
                                                   fields, nfields, xloc, yloc, sign, send_south, send_north, recv_south, recv_north,
                                                   Nx, Ny, Nz, Hx, Hy, Hz, ft, stream);
}

// The same whole fill with the seam exchange pipelined per stage of `fields_per_stage` fields on `comm_stream`
// (tpg_halo_exchange_y_pipelined_peers); packed messages only.
int tpg_fill_halo_regions_distributed_pipelined_peers(void* comm, int south_peer, int north_peer, int north_is_zipper,
                                                      void* const fields[], int nfields,
                                                      const int8_t xloc[], const int8_t yloc[], const int32_t sign[],
                                                      void* send_south, void* send_north, void* recv_south, void* recv_north,
                                                      int Nx, int Ny, int Nz, int Hx, int Hy, int Hz, int ft,
                                                      void* stream, void* comm_stream, int fields_per_stage)
{
    if (north_is_zipper && north_peer >= 0) {
        tpg::set_error("the north side is either the zipper or a seam, not both (north_peer %d)", north_peer);
        return TPG_ERR_INVALID_ARGUMENT;
    }
    if (nfields > TPG_MAX_FIELDS && (south_peer >= 0 || north_peer >= 0)) {
        tpg::set_error("at most %d fields per distributed fill (one seam message per side)", TPG_MAX_FIELDS);
        return TPG_ERR_UNSUPPORTED;
    }
    int rc = tpg_fill_halo_regions(fields, nfields, xloc, yloc, sign, Nx, Ny, Nz, Hx, Hy, Hz, north_is_zipper ? 1 : 0, ft, stream);
    if (rc) return rc;
    if (south_peer < 0 && north_peer < 0) return TPG_OK;
    return tpg_halo_exchange_y_pipelined_peers(comm, south_peer, north_peer, fields, nfields, send_south, send_north, recv_south, recv_north,
                                               Nx, Ny, Nz, Hx, Hy, Hz, ft, stream, comm_stream, fields_per_stage);
}

int tpg_fill_halo_regions_distributed_pipelined(void* comm, int rank, int nranks, void* const fields[], int nfields,
                                                const int8_t xloc[], const int8_t yloc[], const int32_t sign[],
                                                void* send_south, void* send_north, void* recv_south, void* recv_north,
                                                int Nx, int Ny, int Nz, int Hx, int Hy, int Hz, int ft,
                                                void* stream, void* comm_stream, int fields_per_stage)
{
    if (nranks < 1 || rank < 0 || rank >= nranks) { tpg::set_error("rank %d outside 0:%d", rank, nranks - 1); return TPG_ERR_BAD_PARTITION; }
    return tpg_fill_halo_regions_distributed_pipelined_peers(comm, rank > 0 ? rank - 1 : -1, rank < nranks - 1 ? rank + 1 : -1, rank == nranks - 1,
                                                             fields, nfields, xloc, yloc, sign, send_south, send_north, recv_south, recv_north,
                                                             Nx, Ny, Nz, Hx, Hy, Hz, ft, stream, comm_stream, fields_per_stage);
}

}  // extern "C"
