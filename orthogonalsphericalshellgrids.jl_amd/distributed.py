"""Latitude-band (y-slab) halo exchange for a distributed TripolarGrid.

Reference: the reference forbids x-partitioning (src/distributed_tripolar_grid.jl:28-31) and leaves
the seam traffic to Oceananigans' DistributedComputations (MPI Isend/Irecv of one packed buffer per
side; reached from :171,195).  Here: one process per GPU; each interior seam swaps Hy full rows
(all i incl. x halos, all levels incl. z halos) in both directions with point-to-point RCCL
send/recv over xGMI (torch.distributed backend "nccl"); gathering / scattering between the padded
fields and the contiguous message is done by the HIP kernels tpg_pack_y_halo / tpg_unpack_y_halo.
No collective is involved: a y-slab chain only ever talks to its two neighbours.
"""
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


def torch_distributed_transport(plan, send: Dict[int, torch.Tensor], recv: Dict[int, torch.Tensor], group=None):
    """All sends/recvs of one halo fill as ONE batched point-to-point group
    (ncclGroupStart/End under the "nccl" = RCCL backend; also valid on "gloo")."""
    import torch.distributed as dist
    if not plan:
        return
    ops = []
    for m in plan:
        ops.append(dist.P2POp(dist.isend, send[m.side], m.peer, group))
        ops.append(dist.P2POp(dist.irecv, recv[m.side], m.peer, group))
    for req in dist.batch_isend_irecv(ops):
        req.wait()


def message_shape(nfields, f):
    """[field][level][Hy rows][Nx+2Hx] -- matches tpg_pack_y_halo"""
    return (nfields, f.Nz + 2 * f.Hz, f.Hy, f.Nx + 2 * f.Hx)


_BUFFERS: Dict[tuple, Dict[str, Dict[int, torch.Tensor]]] = {}


def _message_buffers(shape, dtype, device, plan):
    """send / recv message buffers, kept across fills (a halo fill runs every time step: no
    allocator traffic on the hot path; one pair per seam side, geometry, dtype and device)"""
    key = (tuple(shape), dtype, str(device), tuple(m.side for m in plan))
    bufs = _BUFFERS.get(key)
    if bufs is None:
        bufs = {"send": {m.side: torch.empty(shape, dtype=dtype, device=device) for m in plan},
                "recv": {m.side: torch.empty(shape, dtype=dtype, device=device) for m in plan}}
        _BUFFERS[key] = bufs
    return bufs["send"], bufs["recv"]


def exchange_y_halos(fields, arch, transport: Optional[Callable] = None):
    """Fill the y-seam halo rows of `fields` (one geometry) from the neighbour ranks."""
    plan = exchange_plan(arch.local_rank, arch.ranks[1])
    if not plan:
        return
    transport = transport or torch_distributed_transport
    lib = _lib.lib()
    f0 = fields[0]
    geom = (f0.Nx, f0.Ny, f0.Nz, f0.Hx, f0.Hy, f0.Hz)
    ft = _lib.ft_of(f0.data.dtype)
    dev = f0.data.device
    for b0 in range(0, len(fields), _lib.TPG_MAX_FIELDS):
        batch = fields[b0:b0 + _lib.TPG_MAX_FIELDS]
        ptrs = _lib.ptr_table([f.data for f in batch])
        shape = message_shape(len(batch), f0)
        send, recv = _message_buffers(shape, f0.data.dtype, dev, plan)
        with torch.cuda.device(dev):
            stream = _lib.current_stream_ptr(dev)
            for m in plan:
                _lib.check(lib.tpg_pack_y_halo(ptrs, len(batch), send[m.side].data_ptr(), m.side, *geom, ft, stream))
            transport(plan, send, recv, getattr(arch, "process_group", None))
            for m in plan:
                _lib.check(lib.tpg_unpack_y_halo(ptrs, len(batch), recv[m.side].data_ptr(), m.side, *geom, ft, stream))
