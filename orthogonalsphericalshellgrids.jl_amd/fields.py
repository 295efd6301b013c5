"""Fields on a TripolarGrid and fill_halo_regions!: host-side mirror of the Field constructor
override (src/tripolar_grid_extensions.jl:57-80, src/distributed_tripolar_grid.jl:159-198) and of
the halo-fill entry that reaches _fill_north_halo! (src/zipper_boundary_condition.jl:146-155).

Field data is one torch tensor of shape (Nz'+2Hz, Ny+2Hy, Nx+2Hx) in HBM whose memory is the
`parent` of the reference's OffsetArray (i fastest).  fill_halo_regions batches every field of
one geometry into ONE zipper launch + one periodic-x launch, or into a single fused launch when the
fields are small (tpg_fill_halo_regions).
"""
import ctypes as C

import torch

from . import _lib
from .boundary_conditions import (Center, Face, FieldBoundaryConditions, PeriodicBoundaryCondition,
                                  ZipperBoundaryCondition, HaloCommunicationBoundaryCondition,
                                  is_zipper, sign)
from .grids import is_tripolar


def _loc_code(L):
    if L is Center:
        return _lib.TPG_CENTER
    if L is Face:
        return _lib.TPG_FACE
    return None          # Nothing: reduced dimension


class Field:
    """Field((LX, LY, LZ), grid::TRG; boundary_conditions, data)

    If the supplied north boundary condition is not already a Zipper, it is replaced by
    ZipperBoundaryCondition(sign(LX, LY)) -- -1 on (Face,Center) / (Center,Face), +1 otherwise
    (tripolar_grid_extensions.jl:49-53,65-67).  On a distributed grid only the last rank gets the
    zipper; the other ranks' north side, and every seam, are neighbour communication
    (distributed_tripolar_grid.jl:171,177-185).
    """

    def __init__(self, loc, grid, data=None, boundary_conditions="default", name=None, indices=(slice(None), slice(None), slice(None))):
        if not is_tripolar(grid):
            raise TypeError("Field: grid must be a TripolarGrid")
        LX, LY, LZ = loc
        self.loc = (LX, LY, LZ)
        self.grid = grid
        self.name = name
        g = getattr(grid, "underlying_grid", grid)
        self.Nx, self.Ny = g.Nx, g.Ny
        self.Hx, self.Hy = g.Hx, g.Hy
        if LZ is None:                       # reduced in z (e.g. bottom_height: (Center, Center, Nothing))
            self.Nz, self.Hz = 1, 0
        else:
            self.Nz = g.Nz + (1 if LZ is Face else 0)   # Bounded z: Nz+1 faces
            self.Hz = g.Hz
        # validate_indices (src/tripolar_grid_extensions.jl:58): `indices` = (:, :, k1:k2) windows the field in z -- given here as a
        # range of 1-based levels, range(k1, k2 + 1), or one level k.  The parent of a windowed dimension holds exactly those levels
        # and no halo [recalled: Oceananigans offset_data], so the kernels see Nz = k2 - k1 + 1, Hz = 0.  Windows in x or y are
        # Oceananigans' own fill: the kernels address whole padded rows.
        full = slice(None)
        ix, iy, iz = indices
        if ix != full or iy != full:
            raise NotImplementedError("Field: indices windowed in x or y are not handled by this library (whole padded rows only)")
        self.z_window = None
        if iz != full:
            if LZ is None:
                raise ValueError("Field: a field reduced in z cannot be windowed in z")
            levels = range(iz, iz + 1) if isinstance(iz, int) else iz
            if not isinstance(levels, range) or levels.step != 1 or len(levels) < 1 or levels[0] < 1 or levels[-1] > self.Nz:
                raise ValueError(f"Field: z indices {iz!r} outside 1:{self.Nz} (a range of consecutive 1-based levels, or one level)")
            self.z_window = (levels[0], levels[-1])
            self.Nz, self.Hz = len(levels), 0
        self.indices = (full, full, iz)
        shape = (self.Nz + 2 * self.Hz, self.Ny + 2 * self.Hy, self.Nx + 2 * self.Hx)
        if data is None:
            data = torch.zeros(shape, dtype=g.dtype, device=g.device)
        else:
            # validate_field_data
            if tuple(data.shape) != shape or not data.is_contiguous() or data.device != g.device:
                raise ValueError(f"field data must be a contiguous {shape} tensor on {g.device}")
        self.data = data

        arch = g.architecture
        distributed = getattr(arch, "is_distributed", False)
        last_rank = (not distributed) or arch.local_rank == arch.ranks[1] - 1
        if boundary_conditions is None:
            self.boundary_conditions = None      # isnothing(old_bcs): kept as is (:62-63)
        else:
            old = FieldBoundaryConditions(west=PeriodicBoundaryCondition(), east=PeriodicBoundaryCondition()) \
                if isinstance(boundary_conditions, str) else boundary_conditions
            old.validate((LX, LY, LZ))       # validate_boundary_conditions (:60)
            default_zipper = ZipperBoundaryCondition(sign(LX, LY))
            north, south = old.north, old.south
            if distributed:
                r = arch.local_rank
                if r > 0:
                    south = HaloCommunicationBoundaryCondition(r, r - 1)
                north = (old.north if is_zipper(old.north) else default_zipper) if last_rank \
                    else HaloCommunicationBoundaryCondition(r, r + 1)
            else:
                north = old.north if is_zipper(old.north) else default_zipper
            self.boundary_conditions = FieldBoundaryConditions(
                west=old.west, east=old.east, south=south, north=north, top=old.top, bottom=old.bottom)

    # --- convenience mirroring Oceananigans' interior / set! ---------------------------------
    def interior(self):
        """interior(field): view indexed [k-1, j-1, i-1]"""
        return self.data[self.Hz:self.Hz + self.Nz, self.Hy:self.Hy + self.Ny, self.Hx:self.Hx + self.Nx]

    def nodes(self):
        """(λ, φ) of the field's horizontal location on the interior, each (Ny, Nx)"""
        g = getattr(self.grid, "underlying_grid", self.grid)
        suffix = ("f" if self.loc[0] is Face else "c") + ("f" if self.loc[1] is Face else "c")
        return g.interior("lambda_" + suffix), g.interior("phi_" + suffix)

    def set_(self, value):
        """set!(field, value): a number, a tensor broadcastable to the interior, or f(λ, φ, z)"""
        inter = self.interior()
        if callable(value):
            g = getattr(self.grid, "underlying_grid", self.grid)
            lam, phi = self.nodes()
            if self.loc[2] is None:
                z = torch.zeros(1, dtype=g.dtype, device=g.device)
            else:
                zz = g.z_faces if self.loc[2] is Face else g.z_centers
                k0 = g.Hz + (self.z_window[0] - 1 if self.z_window else 0)
                z = zz[k0:k0 + self.Nz]
            value = value(lam[None, :, :].to(inter.dtype), phi[None, :, :].to(inter.dtype), z[:, None, None])
        inter.copy_(torch.as_tensor(value, dtype=inter.dtype, device=inter.device).expand_as(inter))
        return self

    def __repr__(self):
        names = tuple("Nothing" if L is None else L.__name__ for L in self.loc)
        north = None if self.boundary_conditions is None else self.boundary_conditions.north
        return f"Field{names} on {self.Nx}×{self.Ny}×{self.Nz} tripolar grid, north: {north}"


def set_(field, value):
    return field.set_(value)


def interior(field):
    return field.interior()


def CenterField(grid, **kw):
    return Field((Center, Center, Center), grid, **kw)


def XFaceField(grid, **kw):
    return Field((Face, Center, Center), grid, **kw)


def YFaceField(grid, **kw):
    return Field((Center, Face, Center), grid, **kw)


def ZFaceField(grid, **kw):
    return Field((Center, Center, Face), grid, **kw)


# -------------------------------------------------------------------------------------------------
# fill_halo_regions!
# -------------------------------------------------------------------------------------------------
def _groups(fields):
    groups = {}
    for f in fields:
        if f.boundary_conditions is None:
            continue
        key = (f.data.dtype, f.data.device, f.Nx, f.Ny, f.Nz, f.Hx, f.Hy, f.Hz, id(f.grid))
        groups.setdefault(key, []).append(f)
    return groups.values()


def _check_supported(f):
    """Scope of this halo fill: the north Zipper (the reference's own code), Oceananigans' periodic x pass
    that must follow it, and the latitude-band seams.  South / bottom / top conditions and non-periodic x conditions are
    Oceananigans' (the reference's own fills leave them `nothing`, src/tripolar_grid.jl:147-152): a field that carries one
    is refused loudly instead of being returned with stale halos."""
    from .boundary_conditions import HaloCommunication, Periodic
    bcs = f.boundary_conditions
    for side in ("west", "east"):
        bc = getattr(bcs, side)
        if bc is not None and not isinstance(bc.classification, Periodic):
            raise NotImplementedError(f"fill_halo_regions: {side} boundary condition {bc.classification!r} is not handled "
                                      "(tripolar grids are Periodic in x)")
    for side in ("south", "north"):
        bc = getattr(bcs, side)
        if bc is not None and not is_zipper(bc) and not isinstance(bc.classification, HaloCommunication):
            raise NotImplementedError(f"fill_halo_regions: {side} boundary condition {bc.classification!r} is Oceananigans' to fill; "
                                      "this library fills the Zipper north side, periodic x and the latitude-band seams only")
    for side in ("bottom", "top"):
        if getattr(bcs, side) is not None:
            raise NotImplementedError(f"fill_halo_regions: {side} boundary conditions (z halos) are Oceananigans' to fill, not handled here")


def _tables(fs):
    xl, yl, sg = [], [], []
    for f in fs:
        x, y = _loc_code(f.loc[0]), _loc_code(f.loc[1])
        north = f.boundary_conditions.north
        if is_zipper(north) and (x is None or y is None):
            # _fill_north_halo! has methods for the four (x, y) location pairs only (:140-155)
            raise TypeError(f"no zipper method for a field at {f.loc}")
        xl.append(0 if x is None else x)
        yl.append(0 if y is None else y)
        sg.append(int(north.condition) if is_zipper(north) else 1)
    n = len(fs)
    return (C.c_int8 * n)(*xl), (C.c_int8 * n)(*yl), (C.c_int32 * n)(*sg)


def _agree_across_ranks(arch, value, what):
    """One collective at plan build: every rank of the chain must hold the same `value` (the stage layout of the seam exchange --
    group(k) of a rank pairs with group(k) of its neighbour, include/tripolar_hip.h).  Needs torch.distributed initialised over the
    same ranks as the chain; a host that ferried the RCCL id some other way vouches for the agreement itself."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return
    group = getattr(arch, "process_group", None)
    if dist.get_world_size(group) != arch.ranks[1]:
        return                                  # not the chain's own group (e.g. one process emulating a band of a longer chain)
    # Deliberately NOT remembered between calls: any scheme that lets a rank skip the collective on local knowledge (a memo of agreed layouts,
    # "same as last time") leaves the one rank that brings a different layout alone in it -- the very case this check exists to report
    # (tests/test_distributed_gloo.py::test_ranks_must_agree_on_the_stage_layout_of_the_pipelined_exchange hangs with such a memo; tried in
    # round 6).  The cost -- one host-blocking collective per plan build -- is the reason to build a plan ONCE for fields that are filled
    # every step (halo_fill_plan), which is what a model's time loop does.
    vals = [None] * arch.ranks[1]
    dist.all_gather_object(vals, value, group=group)
    if any(v != vals[0] for v in vals):
        raise ValueError(f"HaloFillPlan: the ranks of the latitude-band chain disagree on {what}: {vals} "
                         "(every rank must pass the same fields, pack_free and fields_per_stage)")


class HaloFillPlan:
    """fill_halo_regions!(fields...) with everything that does not change from call to call -- grouping by
    geometry, location / sign tables, pointer tables -- built once.  A halo fill runs every (sub-)step on
    the same fields: calling the plan costs one C call per geometry group instead of ~10 us of Python.

    Order (pinned by test/test_zipper_boundary_conditions.jl:42-45):
    zipper fold on the north side (serial grid or last rank) -> periodic x (fills corners) ->
    on a distributed grid the y-seam exchange of Hy rows with the neighbour ranks.
    `exchange` overrides the transport (used by the CPU/gloo tests of the host logic).
    The plan holds the fields' tensors: it must be rebuilt if a field's `data` is replaced.
    """

    def __init__(self, fields, *, exchange=None, pack_free=False, fields_per_stage=0):
        if isinstance(fields, Field):
            fields = [fields]
        self.fields = list(fields)
        self._exchange = exchange
        self._pack_free = pack_free
        if fields_per_stage < 0 or (fields_per_stage and pack_free):
            raise ValueError("fields_per_stage must be >= 0 and excludes pack_free (the pipelined seam exchange is a packed one)")
        self._fields_per_stage = int(fields_per_stage)      # > 0: pipelined seam exchange in stages of that many fields (distributed grids)
        self._comm_streams = {}                               # device -> the second stream of the pipelined RCCL exchange
        self._steps = []                      # (device, [(c function, argument tuple without the stream)], PendingExchange or None)
        self._keep = []                       # message buffers referenced by raw pointer from the argument tuples
        lib = _lib.lib()
        for f in self.fields:
            if f.boundary_conditions is not None:
                _check_supported(f)
        for fs in _groups(self.fields):
            f0 = fs[0]
            g = getattr(f0.grid, "underlying_grid", f0.grid)
            arch = g.architecture
            zip_fs = [f for f in fs if is_zipper(f.boundary_conditions.north)]
            ft = _lib.ft_of(f0.data.dtype)
            geom = (f0.Nx, f0.Ny, f0.Nz, f0.Hx, f0.Hy, f0.Hz)
            distributed = getattr(arch, "is_distributed", False) and arch.ranks[1] > 1
            comm = getattr(arch, "rccl_comm", None) if (distributed and exchange is None) else None
            uniform = len(zip_fs) in (0, len(fs))
            calls, pending = [], None
            if comm is not None:
                # EVERY rank of the chain enters the agreement, before it picks its branch: a rank whose fields are not uniformly zipped takes
                # the host-driven branch below, and its peers must not be left waiting in the collective for it
                _agree_across_ranks(arch, (len(fs), geom[0], geom[2:], bool(pack_free), self._fields_per_stage),
                                    "(fields, Nx, Nz + halos, pack_free, fields_per_stage) of a seam exchange")
            if comm is not None and uniform and (bool(zip_fs) == (arch.local_rank == arch.ranks[1] - 1)):
                # the production path of a DistributedTripolarGrid: ONE C call per batch of <= TPG_MAX_FIELDS fields does the whole
                # fill_halo_regions! -- zipper (last rank) -> periodic x -> RCCL seam exchange -- on the current stream
                from .distributed import NORTH, SOUTH, SeamBuffers, exchange_plan, message_shape
                plan = exchange_plan(arch.local_rank, arch.ranks[1])
                for b0 in range(0, len(fs), _lib.TPG_MAX_FIELDS):
                    batch = fs[b0:b0 + _lib.TPG_MAX_FIELDS]
                    xl, yl, sg = _tables(batch) if zip_fs else (None, None, None)
                    bufs = None if pack_free else SeamBuffers(message_shape(len(batch), f0), f0.data.dtype, f0.data.device, plan)
                    self._keep.append(bufs)
                    p = (lambda w, side: None) if bufs is None else bufs.ptr
                    args = (comm.handle, arch.local_rank, arch.ranks[1], _lib.ptr_table([f.data for f in batch]), len(batch), xl, yl, sg,
                            p("send", SOUTH), p("send", NORTH), p("recv", SOUTH), p("recv", NORTH), *geom, ft)
                    if self._fields_per_stage:
                        dev = f0.data.device
                        if dev not in self._comm_streams:
                            self._comm_streams[dev] = torch.cuda.Stream(dev)
                        calls.append((lib.tpg_fill_halo_regions_distributed_pipelined, args,
                                      (C.c_void_p(self._comm_streams[dev].cuda_stream), self._fields_per_stage)))
                    else:
                        calls.append((lib.tpg_fill_halo_regions_distributed, args))
            else:
                if zip_fs and len(zip_fs) == len(fs):
                    # the usual case: one entry point for zipper -> periodic x (a single fused launch for small
                    # fields such as the 2-D free-surface / barotropic fields, a single merged launch for large ones)
                    xl, yl, sg = _tables(fs)
                    calls.append((lib.tpg_fill_halo_regions, (_lib.ptr_table([f.data for f in fs]), len(fs), xl, yl, sg, *geom, 1, ft)))
                else:
                    if zip_fs:
                        xl, yl, sg = _tables(zip_fs)
                        calls.append((lib.tpg_zipper_fill, (_lib.ptr_table([f.data for f in zip_fs]), len(zip_fs), xl, yl, sg,
                                                            *geom, 1, f0.Nz, ft)))
                    calls.append((lib.tpg_periodic_x_fill, (_lib.ptr_table([f.data for f in fs]), len(fs), *geom, ft)))
                if distributed:
                    from .distributed import PendingExchange
                    pending = PendingExchange(fs, arch, exchange, pack_free, self._fields_per_stage)     # owns its message buffers; reused every fill
            self._steps.append((f0.data.device, calls, pending))

    @property
    def is_distributed(self):
        return any(p is not None for _, _, p in self._steps) or bool(self._keep)

    def begin(self):
        """local part of the fill (zipper, periodic x) on every geometry group, then pack + post of the seam exchange
        (on the RCCL path the one C call has done the whole fill, exchange included, by the time begin() returns)"""
        for device, calls, pending in self._steps:
            if torch.cuda.current_device() == device.index:     # the common case: no device switch to pay for
                stream = _lib.current_stream_ptr(device)
                for fn, args, *after in calls:                 # after: arguments that follow the stream (pipelined exchange)
                    _lib.check(fn(*args, stream, *(after[0] if after else ())))
            else:
                with torch.cuda.device(device):
                    stream = _lib.current_stream_ptr(device)
                    for fn, args, *after in calls:
                        _lib.check(fn(*args, stream, *(after[0] if after else ())))
            if pending is not None:
                pending.begin()
        return self

    def finish(self):
        """delivery + unpack of the seam messages posted by begin()"""
        for _, _, pending in self._steps:
            if pending is not None:
                pending.finish()
        return None

    def __call__(self):
        self.begin()
        return self.finish()

    def graph(self, repeat=1):
        """Capture `repeat` consecutive runs of this plan into one HIP graph (torch.cuda.CUDAGraph) and return
        it; `graph.replay()` then issues the whole sequence with a single launch -- the fills of a
        split-explicit sub-cycle are launch-bound, 2.7 us instead of 7 us per fill.
        Serial grids only: a seam exchange cannot be captured (torch.distributed cannot, and the C ABI's RCCL exchange refuses a
        capturing stream -- DESIGN.md 5)."""
        if self.is_distributed:
            raise ValueError("HaloFillPlan.graph: plans with a distributed seam exchange cannot be captured")
        self()                                           # first-call work (occupancy queries, lazy module load) outside the capture
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.cuda.graph(g, stream=side):
            for _ in range(repeat):
                self()
        torch.cuda.current_stream().wait_stream(side)
        return g


def halo_fill_plan(fields, *, exchange=None, pack_free=False, fields_per_stage=0):
    return HaloFillPlan(fields, exchange=exchange, pack_free=pack_free, fields_per_stage=fields_per_stage)


def fill_halo_regions(fields, *, exchange=None):
    """fill_halo_regions!(fields...) on a tripolar grid: builds a HaloFillPlan and runs it once.  For fields that are filled repeatedly
    (every step of a time loop) build the plan once -- `plan = halo_fill_plan(fields)`, then `plan()`: on a distributed grid every plan BUILD
    runs one host-blocking agreement collective per geometry group (with the nccl backend it also synchronises the device), a plan CALL
    runs none.  Plans are not cached behind this function: a plan holds its fields, and a cache reachable from a field ties 32 GB tensors
    into a reference cycle that only the cycle collector frees (tried in round 6: the next test ran out of HBM)."""
    return HaloFillPlan(fields, exchange=exchange)()
