"""Latitude-band (y-slab) halo exchange for a distributed TripolarGrid.

Reference: the reference forbids x-partitioning (src/distributed_tripolar_grid.jl:28-31) and leaves
the seam traffic to Oceananigans' DistributedComputations (MPI Isend/Irecv of one packed buffer per
side; reached from :171,195).  Here: one process per GPU; each interior seam swaps Hy full rows
(all i incl. x halos, all levels incl. z halos) in both directions with point-to-point RCCL
send/recv over xGMI.  No collective is involved: a y-slab chain only ever talks to its two neighbours.

Three transports, all bit-identical in what they deliver:
  * RcclComm (the production path): the C ABI's tpg_fill_halo_regions_distributed -- the WHOLE fill of a band in one call: zipper
    (last rank) -> periodic x -> tpg_halo_exchange_y = pack -> ONE ncclGroupStart/End of ncclSend/ncclRecv on the caller's stream ->
    unpack (or pack-free: the per-level contiguous seam windows sent from / received into the fields directly).  No host wait.
    `fields_per_stage = k` selects the PIPELINED form (tpg_fill_halo_regions_distributed_pipelined): the same messages in stages of k
    fields, the RCCL group of stage s on a second stream beside pack(s+1..) and unpack(s-1); identical results.
    HaloFillPlan issues that call when the architecture carries an RcclComm.  The communicator is librccl's own
    (tpg_comm_init_rank); torch.distributed only ferries the 128-byte unique id, after every rank has reported that it can bind librccl.
  * torch_distributed_transport: `batch_isend_irecv` of the packed messages (backend "nccl" = RCCL, or "gloo" with
    host tensors): the Python convenience, and what the CPU/gloo tests of the host protocol run.
  * any object with post(plan, send, recv, group) / wait(handle) (two-phase), or a plain callable
    transport(plan, send, recv, group): test transports (loop-back emulation of R ranks on one GPU, host staging).
Message buffers belong to the exchange object (one SeamBuffers per batch of TPG_MAX_FIELDS fields), never to a module-level cache:
two plans or two batches that are in flight together must not share staging memory.
"""
import ctypes as C
from dataclasses import dataclass
from typing import Callable, Dict, List, Optional

import torch

from . import _lib

SOUTH, NORTH = 0, 1


@dataclass(frozen=True)
class SeamMessage:
    """One direction pair across one seam, seen from this rank: we SEND our interior rows next to
    `side` to `peer` and RECEIVE the peer's interior rows into our halo rows on `side`."""
    side: int
    peer: int


def exchange_plan(rank: int, nranks: int) -> List[SeamMessage]:
    """Seams of rank `rank` in a chain of `nranks` latitude bands (rank 0 = southernmost).
    The north side of the last rank is the zipper (device-local), the south side of rank 0 is the
    grid's southern edge: neither communicates."""
    plan = []
    if rank < nranks - 1:
        plan.append(SeamMessage(NORTH, rank + 1))
    if rank > 0:
        plan.append(SeamMessage(SOUTH, rank - 1))
    return plan


# -------------------------------------------------------------------------------------------------
# transports
# -------------------------------------------------------------------------------------------------
class TorchDistributedTransport:
    """All sends/recvs of one halo fill as ONE batched point-to-point group
    (ncclGroupStart/End under the "nccl" = RCCL backend; also valid on "gloo")."""

    def post(self, plan, send: Dict[int, torch.Tensor], recv: Dict[int, torch.Tensor], group=None):
        import torch.distributed as dist
        if not plan:
            return []
        ops = []
        for m in plan:
            ops.append(dist.P2POp(dist.isend, send[m.side], m.peer, group))
            ops.append(dist.P2POp(dist.irecv, recv[m.side], m.peer, group))
        return dist.batch_isend_irecv(ops)

    def wait(self, handle):
        # "nccl": makes the current stream wait for the communication (no host block); "gloo": blocks the host
        for req in handle:
            req.wait()

    def __call__(self, plan, send, recv, group=None):
        self.wait(self.post(plan, send, recv, group))


torch_distributed_transport = TorchDistributedTransport()


class LoopbackMailbox:
    """Two-phase transport for R latitude-band ranks emulated in ONE process on one GPU (tests/test_gpu_distributed.py,
    tests/soak/soak_distributed.py): post() parks the rank's packed messages in a shared mailbox, wait() delivers the peers'
    messages into the receive buffers.  Drive it as: plan_r.begin() for every rank r, then plan_r.finish() for every r."""

    def __init__(self):
        self.box = {}

    def endpoint(self, me):
        return _LoopbackEndpoint(self.box, me)


class _LoopbackEndpoint:
    def __init__(self, box, me):
        self.box, self.me = box, me

    def post(self, plan, send, recv, group=None):
        for m in plan:
            self.box.setdefault((self.me, m.peer), []).append(send[m.side].clone())
        return plan, recv

    def wait(self, handle):
        plan, recv = handle
        for m in plan:
            recv[m.side].copy_(self.box[(m.peer, self.me)].pop(0))


class RcclComm:
    """ncclComm_t of the latitude-band chain, created through the C ABI (tpg_comm_init_rank: librccl's
    ncclCommInitRank on the current device).  `RcclComm.from_torch(group)` bootstraps it over an initialised
    torch.distributed group of any backend: rank 0 draws the unique id, broadcast_object_list ferries it."""

    def __init__(self, handle, rank, nranks):
        self.handle, self.rank, self.nranks = handle, rank, nranks

    @staticmethod
    def unique_id() -> bytes:
        buf = (C.c_char * _lib.TPG_COMM_ID_BYTES)()
        _lib.check(_lib.lib().tpg_comm_unique_id(C.cast(buf, C.c_void_p)))
        return bytes(buf.raw)

    @classmethod
    def create(cls, unique_id: bytes, rank: int, nranks: int):
        if len(unique_id) != _lib.TPG_COMM_ID_BYTES:
            raise ValueError("unique id must be 128 bytes")
        comm = C.c_void_p()
        buf = C.create_string_buffer(unique_id, _lib.TPG_COMM_ID_BYTES)
        _lib.check(_lib.lib().tpg_comm_init_rank(C.byref(comm), nranks, C.cast(buf, C.c_void_p), rank))
        return cls(comm, rank, nranks)

    @staticmethod
    def available() -> bool:
        """librccl can be bound on this rank (no collective call): agree on it across ranks before create()"""
        return _lib.lib().tpg_comm_available() == 0

    @classmethod
    def from_torch(cls, group=None):
        """Collective over `group`.  Every rank first reports whether it can bind librccl and all ranks agree (a MIN
        reduction of the flag through all_gather_object) BEFORE anyone enters the collective ncclCommInitRank: a rank that
        cannot would otherwise leave the others blocked inside it."""
        import torch.distributed as dist
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        flags = [None] * world
        dist.all_gather_object(flags, bool(cls.available()), group=group)
        if not all(flags):
            raise RuntimeError(f"librccl unavailable on rank(s) {[r for r, ok in enumerate(flags) if not ok]}: no communicator created")
        box = [None]
        if rank == 0:
            try:
                box[0] = cls.unique_id()
            except Exception as e:                      # noqa: BLE001 -- every rank must leave the broadcast, then fail together
                box[0] = e
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        if isinstance(box[0], Exception):
            raise RuntimeError(f"rank 0 could not draw an RCCL unique id: {box[0]}")
        return cls.create(box[0], rank, world)

    def destroy(self):
        if self.handle is not None:
            _lib.check(_lib.lib().tpg_comm_destroy(self.handle))
            self.handle = None


def message_shape(nfields, f):
    """[field][level][Hy rows][Nx+2Hx] -- matches tpg_pack_y_halo"""
    return (nfields, f.Nz + 2 * f.Hz, f.Hy, f.Nx + 2 * f.Hx)


class SeamBuffers:
    """send / recv message buffers of ONE batch (<= TPG_MAX_FIELDS fields of one geometry) of ONE exchange object.
    They belong to the PendingExchange (and through it to the HaloFillPlan) that created them -- never to a module-level
    cache: two plans of equal geometry, or two batches of one plan, that are in flight together (begun before either is
    finished, or enqueued on different streams) must not share staging memory.  Allocated once, reused every fill."""

    def __init__(self, shape, dtype, device, plan):
        self.send = {m.side: torch.empty(shape, dtype=dtype, device=device) for m in plan}
        self.recv = {m.side: torch.empty(shape, dtype=dtype, device=device) for m in plan}

    def ptr(self, which, side):
        d = self.send if which == "send" else self.recv
        return None if side not in d else d[side].data_ptr()


class PendingExchange:
    """The seam exchange of one group of fields (one geometry): begin = pack + post, finish = wait + unpack.  Splitting the
    two lets a single process drive several emulated ranks (all post, then all finish) and lets a caller put work between
    them.  The object is reusable (a HaloFillPlan keeps one per geometry group and runs it every step) and owns its message
    buffers, one SeamBuffers per batch of TPG_MAX_FIELDS fields.

    With an RcclComm on the architecture (and no explicit transport) the exchange is the C ABI's: `run()` = ONE call of
    tpg_halo_exchange_y per batch (pack -> RCCL send/recv group -> unpack on the current stream, no host wait);
    begin() then does nothing and finish() runs it."""

    def __init__(self, fields, arch, transport=None, pack_free=False, fields_per_stage=0):
        self.plan = exchange_plan(arch.local_rank, arch.ranks[1])
        self.fields, self.arch = list(fields), arch
        self.comm = getattr(arch, "rccl_comm", None) if transport is None else None
        self.transport = transport if transport is not None else torch_distributed_transport
        self.pack_free = pack_free and self.comm is not None
        # fields_per_stage > 0: the PIPELINED form (tpg_halo_exchange_y_pipelined: stages of that many fields, the RCCL groups on a
        # second stream beside the pack / unpack kernels of the neighbouring stages).  Other transports run the same stages one after
        # the other -- same slices, same kernels, same result; only the RCCL path overlaps them.
        if fields_per_stage < 0 or (fields_per_stage and pack_free):
            raise ValueError("fields_per_stage must be >= 0 and excludes pack_free (the pipelined exchange is a packed one)")
        self.fields_per_stage = int(fields_per_stage)
        self._comm_stream = None
        self._handles = None
        self.batches = []                     # (fields, pointer table, SeamBuffers or None)
        if not self.plan:
            return
        f0 = self.fields[0]
        self.geom = (f0.Nx, f0.Ny, f0.Nz, f0.Hx, f0.Hy, f0.Hz)
        self.ft = _lib.ft_of(f0.data.dtype)
        self.device = f0.data.device
        for b0 in range(0, len(self.fields), _lib.TPG_MAX_FIELDS):
            batch = self.fields[b0:b0 + _lib.TPG_MAX_FIELDS]
            bufs = None if self.pack_free else SeamBuffers(message_shape(len(batch), f0), f0.data.dtype, self.device, self.plan)
            self.batches.append((batch, _lib.ptr_table([f.data for f in batch]), bufs))

    def run(self):
        """RCCL path: the whole exchange of every batch, enqueued on the current stream"""
        lib = _lib.lib()
        with torch.cuda.device(self.device):
            stream = _lib.current_stream_ptr(self.device)
            for batch, ptrs, bufs in self.batches:
                p = (lambda w, side: None) if bufs is None else bufs.ptr
                if self.fields_per_stage:
                    _lib.check(lib.tpg_halo_exchange_y_pipelined(self.comm.handle, self.arch.local_rank, self.arch.ranks[1], ptrs, len(batch),
                                                                 p("send", SOUTH), p("send", NORTH), p("recv", SOUTH), p("recv", NORTH),
                                                                 *self.geom, self.ft, stream, self.comm_stream_ptr(), self.fields_per_stage))
                else:
                    _lib.check(lib.tpg_halo_exchange_y(self.comm.handle, self.arch.local_rank, self.arch.ranks[1], ptrs, len(batch),
                                                       p("send", SOUTH), p("send", NORTH), p("recv", SOUTH), p("recv", NORTH),
                                                       *self.geom, self.ft, stream))

    def comm_stream_ptr(self):
        """the second stream of the pipelined exchange (this object's own, created at first use)"""
        if self._comm_stream is None:
            self._comm_stream = torch.cuda.Stream(self.device)
        return C.c_void_p(self._comm_stream.cuda_stream)

    def _stages(self, nbatch):
        """(first field, count) of every stage of a batch of `nbatch` fields; one stage = the whole batch when not pipelined"""
        fps = min(self.fields_per_stage, nbatch) if self.fields_per_stage else nbatch
        return [(f0, min(fps, nbatch - f0)) for f0 in range(0, nbatch, fps)]

    def _slice(self, batch, ptrs, buf, f0, n):
        """pointer table and message slice of fields f0 .. f0+n-1 of a batch (message layout [field][level][Hy][sx])"""
        if f0 == 0 and n == len(batch):
            return ptrs, buf.data_ptr()
        return _lib.ptr_table([f.data for f in batch[f0:f0 + n]]), buf[f0:f0 + n].data_ptr()

    def begin(self):
        if not self.plan or self.comm is not None:
            return self
        if self._handles is not None:
            raise RuntimeError("PendingExchange.begin() called twice without finish(): its message buffers are still in flight")
        lib = _lib.lib()
        group = getattr(self.arch, "process_group", None)
        self._handles = []
        with torch.cuda.device(self.device):
            stream = _lib.current_stream_ptr(self.device)
            for batch, ptrs, bufs in self.batches:
                for f0, n in self._stages(len(batch)):               # pipelined: one pack launch per stage and side, on that stage's slice
                    for m in self.plan:
                        pt, dst = self._slice(batch, ptrs, bufs.send[m.side], f0, n)
                        _lib.check(lib.tpg_pack_y_halo(pt, n, dst, m.side, *self.geom, self.ft, stream))
                self._handles.append(self.transport.post(self.plan, bufs.send, bufs.recv, group) if hasattr(self.transport, "post") else None)
        return self

    def finish(self):
        if not self.plan:
            return
        if self.comm is not None:
            return self.run()
        if self._handles is None:
            raise RuntimeError("PendingExchange.finish() without begin()")
        lib = _lib.lib()
        group = getattr(self.arch, "process_group", None)
        with torch.cuda.device(self.device):
            stream = _lib.current_stream_ptr(self.device)
            for (batch, ptrs, bufs), handle in zip(self.batches, self._handles):
                if hasattr(self.transport, "post"):
                    self.transport.wait(handle)
                else:
                    self.transport(self.plan, bufs.send, bufs.recv, group)
                for f0, n in self._stages(len(batch)):
                    for m in self.plan:
                        pt, src = self._slice(batch, ptrs, bufs.recv[m.side], f0, n)
                        _lib.check(lib.tpg_unpack_y_halo(pt, n, src, m.side, *self.geom, self.ft, stream))
        self._handles = None


def exchange_y_halos(fields, arch, transport: Optional[Callable] = None, pack_free: bool = False, fields_per_stage: int = 0):
    """Fill the y-seam halo rows of `fields` (one geometry) from the neighbour ranks (one-off: allocates its message
    buffers; keep a PendingExchange / HaloFillPlan for repeated fills)."""
    PendingExchange(fields, arch, transport, pack_free, fields_per_stage).begin().finish()
