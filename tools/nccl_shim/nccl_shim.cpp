// nccl_shim.cpp -- TEST DOUBLE of the ten librccl entry points libtripolar_hip binds (csrc/tpg_exchange.hip: ncclGetUniqueId, ncclCommInitRank,
// ncclCommDestroy, ncclCommCount, ncclCommUserRank, ncclSend, ncclRecv, ncclGroupStart, ncclGroupEnd, ncclGetErrorString).
//
// WHY.  RCCL refuses two ranks on one device and the build boxes have one GPU, so the production branch of the seam exchange -- the C ABI's
// tpg_fill_halo_regions_distributed(_pipelined)_peers with a communicator of MORE THAN ONE rank, as bench.py's N > 1 run and HaloFillPlan issue
// it -- could only ever run on a one-rank communicator whose peers are the rank itself.  With this shim behind the TEST library
// (TPG_RCCL_LIBRARY, read by tools/libtripolar_hip_test.so only) the same calls run between several REAL processes that share one GPU:
// distinct ranks, distinct data, both neighbours different, the pipelined stage groups matched across processes.  It validates OUR use of the
// API (which buffer goes to which peer, group order, stage pairing, message sizes) -- NOT RCCL, not xGMI, nothing about performance, and NOT
// the stream / event ordering around the groups: ncclGroupEnd here synchronises every operation's stream and moves the data with blocking
// copies, so a missing event wait between `stream` and `comm_stream` cannot show (that ordering is exercised on the one-rank RCCL loop-back,
// tools/rccl_selftest.py).  The product library never loads it (it binds librccl by its fixed names and reads no environment variable).
//
// HOW.  One POSIX shared-memory mailbox (one message at a time) per ordered pair (src, dst), created by whichever side gets there first;
// host-staged: a send is hipMemcpy device -> mailbox once the previous message of the pair has been consumed, a receive is hipMemcpy mailbox ->
// device once a message is there.  Operations between ncclGroupStart / ncclGroupEnd are queued and executed at ncclGroupEnd (after
// hipStreamSynchronize of their streams) by a small progress engine: whatever is ready runs, in issue order per (direction, peer) -- the
// order NCCL matches the messages of one pair in -- so sends and receives towards both neighbours advance side by side and a chain of ranks
// cannot dead-lock, whatever the number of messages per group.  If nothing progresses for TPG_SHIM_DEADLINE_S (default 60 s) the group ends in
// ncclSystemError naming the operations still pending: a mis-paired exchange never hangs.  Counts that differ between a send and its receive are an error (ncclInvalidUsage), as a size mismatch
// would be on the wire.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <atomic>
#include <dirent.h>
#include <errno.h>
#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>
#include <map>
#include <vector>

namespace {

constexpr size_t kSlotBytes = 96ull << 20;          // one message per pair at a time; pages are committed only when touched

struct Mailbox {                                    // lives in shared memory
    std::atomic<unsigned long long> written;        // messages put in so far
    std::atomic<unsigned long long> read;           // messages taken out so far
    std::atomic<unsigned long long> bytes;          // size of the message in the slot
    std::atomic<unsigned long long> attached;       // processes that have mapped this mailbox (each end once): unlinked only when both have
    char pad[32];
    char slot[1];
};

struct Comm {
    int rank, nranks;
    char tag[40];                                   // from the unique id: names the shared-memory objects of this communicator
    std::map<std::pair<int, int>, Mailbox*> boxes;  // (src, dst) -> mapping
};

struct Op { bool send; void* ptr; size_t bytes; int peer; Comm* comm; hipStream_t stream; };
thread_local int t_depth = 0;
thread_local std::vector<Op> t_ops;

double now_s() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
double deadline_s() { const char* e = getenv("TPG_SHIM_DEADLINE_S"); return (e && *e) ? atof(e) : 60.0; }

Mailbox* mailbox(Comm* c, int src, int dst)
{
    auto it = c->boxes.find({ src, dst });
    if (it != c->boxes.end()) return it->second;
    char name[96];
    snprintf(name, sizeof name, "/tpgshim_%s_%d_%d", c->tag, src, dst);
    int fd = shm_open(name, O_CREAT | O_RDWR, 0600);
    if (fd < 0) return nullptr;
    const size_t total = sizeof(Mailbox) + kSlotBytes;
    if (ftruncate(fd, (off_t)total) != 0) { close(fd); return nullptr; }     // idempotent: both sides set the same size; new pages read as zero
    void* p = mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) return nullptr;
    Mailbox* m = static_cast<Mailbox*>(p);
    m->attached.fetch_add(1);
    c->boxes[{ src, dst }] = m;
    return m;
}

// one attempt: ncclSuccess = done, ncclInProgress = the mailbox is not ready for this operation yet, anything else = error
ncclResult_t try_send(const Op& op)
{
    Mailbox* m = mailbox(op.comm, op.comm->rank, op.peer);
    if (!m || op.bytes > kSlotBytes) return ncclSystemError;
    if (m->written.load(std::memory_order_acquire) != m->read.load(std::memory_order_acquire)) return ncclInProgress;   // previous message not consumed
    if (hipMemcpy(m->slot, op.ptr, op.bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    m->bytes.store(op.bytes, std::memory_order_relaxed);
    m->written.fetch_add(1, std::memory_order_release);
    return ncclSuccess;
}

ncclResult_t try_recv(const Op& op)
{
    Mailbox* m = mailbox(op.comm, op.peer, op.comm->rank);
    if (!m) return ncclSystemError;
    if (m->written.load(std::memory_order_acquire) == m->read.load(std::memory_order_acquire)) return ncclInProgress;   // nothing there yet
    if (m->bytes.load(std::memory_order_relaxed) != op.bytes) {
        fprintf(stderr, "[nccl_shim] rank %d: receive of %zu bytes from %d meets a message of %llu bytes\n", op.comm->rank, op.bytes, op.peer,
                (unsigned long long)m->bytes.load());
        return ncclInvalidUsage;
    }
    if (hipMemcpy(op.ptr, m->slot, op.bytes, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    m->read.fetch_add(1, std::memory_order_release);
    return ncclSuccess;
}

// A group = a small progress engine: every pending operation whose mailbox is ready is executed, in issue order per (direction, peer)
// -- the order NCCL matches messages of one pair in -- until all are done; sends and receives of BOTH neighbours make progress side by side,
// so groups of many messages per pair (the pack-free exchange) cannot dead-lock on the one-message mailboxes.
ncclResult_t run_all()
{
    std::vector<Op> ops;
    ops.swap(t_ops);
    for (const Op& op : ops)                                         // everything the group's operations may depend on has to be finished
        if (hipStreamSynchronize(op.stream) != hipSuccess) return ncclUnhandledCudaError;
    std::vector<char> done(ops.size(), 0);
    size_t left = ops.size();
    double t_last = now_s();
    while (left) {
        bool progressed = false;
        for (size_t i = 0; i < ops.size(); ++i) {
            if (done[i]) continue;
            bool earlier = false;                                    // FIFO per (direction, peer, communicator)
            for (size_t q = 0; q < i && !earlier; ++q)
                earlier = !done[q] && ops[q].send == ops[i].send && ops[q].peer == ops[i].peer && ops[q].comm == ops[i].comm;
            if (earlier) continue;
            const ncclResult_t rc = ops[i].send ? try_send(ops[i]) : try_recv(ops[i]);
            if (rc == ncclSuccess) { done[i] = 1; --left; progressed = true; }
            else if (rc != ncclInProgress) return rc;
        }
        if (progressed) { t_last = now_s(); continue; }
        if (now_s() - t_last > deadline_s()) {
            for (size_t i = 0; i < ops.size(); ++i)
                if (!done[i]) fprintf(stderr, "[nccl_shim] rank %d: %s %d never completed (%zu bytes): the peer did not post its half\n",
                                      ops[i].comm->rank, ops[i].send ? "send to" : "receive from", ops[i].peer, ops[i].bytes);
            return ncclSystemError;
        }
        usleep(50);
    }
    return ncclSuccess;
}

size_t type_bytes(ncclDataType_t t) { return t == ncclFloat64 ? 8 : (t == ncclFloat32 ? 4 : 0); }

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id)
{
    memset(id, 0, sizeof *id);
    timespec t; clock_gettime(CLOCK_REALTIME, &t);
    snprintf(id->internal, sizeof id->internal, "%lx%lx%x", (long)t.tv_sec, (long)t.tv_nsec, (unsigned)getpid());
    return ncclSuccess;
}

// mailboxes a killed rank left behind (96 MB of address space each, the touched pages resident): anything of ours older than 15 minutes
static void sweep_stale_mailboxes()
{
    DIR* d = opendir("/dev/shm");
    if (!d) return;
    const time_t now = time(nullptr);
    while (dirent* e = readdir(d)) {
        if (strncmp(e->d_name, "tpgshim_", 8) != 0) continue;
        std::string path = std::string("/dev/shm/") + e->d_name;
        struct stat st;
        if (stat(path.c_str(), &st) == 0 && now - st.st_mtime > 900) shm_unlink((std::string("/") + e->d_name).c_str());
    }
    closedir(d);
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank)
{
    if (!comm || nranks < 1 || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    if (rank == 0) sweep_stale_mailboxes();
    Comm* c = new Comm;
    c->rank = rank; c->nranks = nranks;
    id.internal[sizeof c->tag - 1] = 0;
    snprintf(c->tag, sizeof c->tag, "%s", id.internal);
    *comm = reinterpret_cast<ncclComm_t>(c);
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm)
{
    Comm* c = reinterpret_cast<Comm*>(comm);
    if (!c) return ncclSuccess;
    for (auto& kv : c->boxes) {
        // the name goes only once BOTH ends have mapped the object: a peer that has not opened it yet would otherwise create a fresh, empty
        // one under the same name and never see what was sent.  (An end that leaves first keeps the name for the other; a mailbox whose
        // peer never came is removed by the stale-object sweep of a later communicator.)
        const bool both = kv.second->attached.load() >= 2;
        munmap(kv.second, sizeof(Mailbox) + kSlotBytes);
        if (both) {
            char name[96];
            snprintf(name, sizeof name, "/tpgshim_%s_%d_%d", c->tag, kv.first.first, kv.first.second);
            shm_unlink(name);                                          // both ends try; the second one finds it gone
        }
    }
    delete c;
    return ncclSuccess;
}

ncclResult_t ncclCommCount(const ncclComm_t comm, int* n) { *n = reinterpret_cast<Comm*>(comm)->nranks; return ncclSuccess; }
ncclResult_t ncclCommUserRank(const ncclComm_t comm, int* r) { *r = reinterpret_cast<Comm*>(comm)->rank; return ncclSuccess; }

ncclResult_t ncclGroupStart() { ++t_depth; return ncclSuccess; }
ncclResult_t ncclGroupEnd()
{
    if (t_depth <= 0) return ncclInvalidUsage;
    if (--t_depth > 0) return ncclSuccess;
    return run_all();
}

ncclResult_t ncclSend(const void* buf, size_t count, ncclDataType_t dt, int peer, ncclComm_t comm, hipStream_t stream)
{
    Comm* c = reinterpret_cast<Comm*>(comm);
    if (!c || !buf || peer < 0 || peer >= c->nranks || !type_bytes(dt)) return ncclInvalidArgument;
    t_ops.push_back(Op{ true, const_cast<void*>(buf), count * type_bytes(dt), peer, c, stream });
    return t_depth > 0 ? ncclSuccess : run_all();
}

ncclResult_t ncclRecv(void* buf, size_t count, ncclDataType_t dt, int peer, ncclComm_t comm, hipStream_t stream)
{
    Comm* c = reinterpret_cast<Comm*>(comm);
    if (!c || !buf || peer < 0 || peer >= c->nranks || !type_bytes(dt)) return ncclInvalidArgument;
    t_ops.push_back(Op{ false, buf, count * type_bytes(dt), peer, c, stream });
    return t_depth > 0 ? ncclSuccess : run_all();
}

const char* ncclGetErrorString(ncclResult_t r)
{
    switch (r) {
    case ncclSuccess: return "no error";
    case ncclUnhandledCudaError: return "nccl_shim: a HIP call failed";
    case ncclSystemError: return "nccl_shim: mailbox unavailable or deadline expired (mis-paired exchange)";
    case ncclInvalidArgument: return "nccl_shim: invalid argument";
    case ncclInvalidUsage: return "nccl_shim: invalid usage (message size mismatch / unbalanced group)";
    default: return "nccl_shim: error";
    }
}

}  // extern "C"
