"""TripolarGrid: host-side mirror of src/tripolar_grid.jl, src/distributed_tripolar_grid.jl,
src/with_halo.jl and the grid part of src/tripolar_grid_extensions.jl of the reference.

The constructor keeps the reference's keyword surface and error behaviour; all numerical work is
one call of tpg_build_grid (HIP, include/tripolar_hip.h) writing the 20 padded arrays directly in
device memory.  torch is used for device allocations / streams only.
"""
import ctypes as C
import threading
import unicodedata
import weakref
from dataclasses import dataclass
from typing import Any, Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import ARRAY_NAMES

R_Earth = 6371.0e3  # Oceananigans.Grids.R_Earth [recalled]


# ---------------------------------------------------------------------------------------------
# architectures
# ---------------------------------------------------------------------------------------------
class GPU:
    """Oceananigans' GPU() architecture: one MI355X (HIP device `index`)."""
    is_distributed = False

    def __init__(self, index: Optional[int] = None):
        self.index = index

    @property
    def device(self):
        if not torch.cuda.is_available():
            raise RuntimeError("no HIP device visible: this package has a MI355X backend only (no CPU path)")
        return torch.device("cuda", torch.cuda.current_device() if self.index is None else self.index)

    def __repr__(self):
        return "GPU()" if self.index is None else f"GPU({self.index})"


class CPU:
    """Placeholder so that reference scripts fail with a clear message instead of a NameError."""
    is_distributed = False

    @property
    def device(self):
        raise RuntimeError("CPU() architecture is not provided by this build: the product path is HIP-only "
                           "(the CPU restatement lives under oracle/ as test infrastructure)")


@dataclass
class Partition:
    """Oceananigans Partition(x, y, z): number of ranks per direction.  `y_sizes` optionally gives the
    per-rank row counts (Oceananigans Sizes); default is Equal()."""
    x: int = 1
    y: int = 1
    z: int = 1
    y_sizes: Optional[Sequence[int]] = None


class Distributed:
    """Oceananigans.DistributedComputations.Distributed(child_arch; partition): one process per GPU;
    `local_rank` / world size come from torch.distributed (RCCL) unless given explicitly
    (explicit values let a single process build any rank's band)."""
    is_distributed = True

    def __init__(self, child_architecture=None, partition: Optional[Partition] = None,
                 local_rank: Optional[int] = None, process_group=None, rccl_comm=None):
        self.child_architecture = child_architecture if child_architecture is not None else GPU()
        self.process_group = process_group
        self.rccl_comm = rccl_comm          # distributed.RcclComm: seam exchanges go through tpg_halo_exchange_y (C ABI)
        if partition is None:
            import torch.distributed as dist
            partition = Partition(y=dist.get_world_size(process_group) if dist.is_initialized() else 1)
        self.partition = partition
        if local_rank is None:
            import torch.distributed as dist
            local_rank = dist.get_rank(process_group) if dist.is_initialized() else 0
        self.local_rank = local_rank

    @property
    def ranks(self):
        return (self.partition.x, self.partition.y, self.partition.z)

    @property
    def device(self):
        return self.child_architecture.device

    def __repr__(self):
        return f"Distributed({self.child_architecture!r}, ranks={self.ranks}, local_rank={self.local_rank})"


def convert_to_0_360(x):
    """convert_to_0_360(x) = ((x % 360) + 360) % 360  (src/OrthogonalSphericalShellGrids.jl:24); Julia's `%` is the truncated
    remainder (C fmod), not Python's floored one"""
    import math
    return math.fmod(math.fmod(x, 360) + 360, 360)


def child_architecture(arch):
    return arch.child_architecture if getattr(arch, "is_distributed", False) else arch


# topologies (tags only)
class PeriodicTopology: pass
class Bounded: pass
class RightConnected: pass
class FullyConnected: pass


# ---------------------------------------------------------------------------------------------
# grid types
# ---------------------------------------------------------------------------------------------
@dataclass(frozen=True)
class Tripolar:
    """struct Tripolar{N, F, S}  (src/tripolar_grid.jl:6-10): parameters kept verbatim."""
    north_poles_latitude: Any
    first_pole_longitude: Any
    southernmost_latitude: Any


# NFKC-normalised reference property names -> ASCII array names (python normalises identifiers,
# so `grid.Δxᶜᶜᵃ` in source code arrives here as "Δxcca")
_UNICODE_ALIASES = {}
for _n in ARRAY_NAMES:
    _kind, _loc = _n.rsplit("_", 1)
    _sym = {"lambda": "λ", "phi": "φ", "dx": "Δx", "dy": "Δy", "az": "Az"}[_kind]
    _UNICODE_ALIASES[unicodedata.normalize("NFKC", f"{_sym}{_loc}a")] = _n


@dataclass
class OrthogonalSphericalShellGrid:
    """Oceananigans.OrthogonalSphericalShellGrid with conformal_mapping::Tripolar.
    The 20 horizontal arrays are torch tensors of shape (rows, Nx+2Hx) whose memory is exactly the
    `parent` of the reference's OffsetMatrix (i fastest): A[i, j] == tensor[j + Hy - 1, i + Hx - 1]."""
    architecture: Any
    Nx: int
    Ny: int
    Nz: int
    Hx: int
    Hy: int
    Hz: int
    Lz: float
    arrays: dict
    z_faces: torch.Tensor
    z_centers: torch.Tensor
    radius: float
    conformal_mapping: Tripolar
    topology: Tuple[Any, Any, Any]
    dtype: torch.dtype = torch.float64
    global_size: Optional[Tuple[int, int, int]] = None   # distributed grids: size of the global grid
    jrange: Optional[Tuple[int, int]] = None              # distributed grids: owned global rows
    z_spec: Any = (0, 1)
    workspace: Any = None            # TableWorkspace: the 1-D tables tpg_build_grid left behind, kept for builds of the same geometry
    tables_reused: bool = False      # this grid was built with TPG_BUILD_TABLES_VALID (its table kernel was skipped)

    def __getattr__(self, name):
        arrays = self.__dict__.get("arrays", {})
        if name in arrays:
            return arrays[name]
        alias = _UNICODE_ALIASES.get(unicodedata.normalize("NFKC", name))
        if alias is not None and alias in arrays:
            return arrays[alias]
        raise AttributeError(name)

    @property
    def size(self):
        return (self.Nx, self.Ny, self.Nz)

    @property
    def halo_size(self):
        return (self.Hx, self.Hy, self.Hz)

    @property
    def device(self):
        return self.arrays["lambda_cc"].device

    def interior(self, name):
        """view of the Nx x Ny interior of one of the 20 arrays, indexed [j-1, i-1]"""
        a = getattr(self, name)
        return a[self.Hy:self.Hy + self.Ny, self.Hx:self.Hx + self.Nx]

    def __repr__(self):
        tx, ty, tz = (t.__name__ for t in self.topology)
        return (f"{self.Nx}×{self.Ny}×{self.Nz} OrthogonalSphericalShellGrid{{{str(self.dtype).split('.')[-1]}, "
                f"{tx}, {ty}, {tz}}} on {self.architecture!r} with {self.Hx}×{self.Hy}×{self.Hz} halo "
                f"and with precomputed metrics (Tripolar: {self.conformal_mapping})")


def is_tripolar(grid):
    """grid isa TRG  (src/tripolar_grid.jl:371; immersed-boundary wrappers expose .underlying_grid)"""
    g = getattr(grid, "underlying_grid", grid)
    return isinstance(g, OrthogonalSphericalShellGrid) and isinstance(g.conformal_mapping, Tripolar)


def _torch_dtype(FT):
    if FT in (torch.float64, float, "Float64", "float64"):
        return torch.float64
    if FT in (torch.float32, "Float32", "float32"):
        return torch.float32
    try:
        import numpy as np
        return {np.dtype("float64"): torch.float64, np.dtype("float32"): torch.float32}[np.dtype(FT)]
    except Exception:
        raise TypeError(f"FT must be Float32 or Float64, got {FT!r}")


def _z_coordinate(z, Nz, Hz, dtype, device):
    """generate_coordinate(FT, topology, size, halo, z, :z, 3, CPU()) for a Bounded z
    (src/tripolar_grid.jl:91) [recalled]: regular interval tuple or explicit face array."""
    zz = z.flatten().tolist() if torch.is_tensor(z) else list(z)
    if len(zz) == 2:                      # regular interval (z0, z1)
        z0, z1 = float(zz[0]), float(zz[1])
        dz = (z1 - z0) / Nz
        k = torch.arange(-Hz, Nz + 1 + Hz, dtype=torch.float64)
        faces = z0 + k * dz
        Lz = z1 - z0
    else:                                 # explicit face positions
        f = torch.tensor(zz, dtype=torch.float64)
        if f.numel() != Nz + 1:
            raise ValueError(f"z must be a 2-tuple or {Nz + 1} face positions")
        lo = f[0] - (f[1] - f[0]) * torch.arange(Hz, 0, -1, dtype=torch.float64)
        hi = f[-1] + (f[-1] - f[-2]) * torch.arange(1, Hz + 1, dtype=torch.float64)
        faces = torch.cat([lo, f, hi])
        Lz = float(f[-1] - f[0])
    centers = 0.5 * (faces[1:] + faces[:-1])
    return float(Lz), faces.to(dtype).to(device), centers.to(dtype).to(device)


def local_sizes(N, R, sizes=None):
    """Rows per rank of a y-slab partition (Oceananigans local_size / concatenate_local_sizes,
    src/distributed_tripolar_grid.jl:41-44) [recalled for Equal(): N/R each; the remainder rule is
    Oceananigans-internal and unpinned -- here the last rank takes it]."""
    if sizes is not None:
        sizes = [int(s) for s in sizes]
        if len(sizes) != R or sum(sizes) != N or min(sizes) < 1:
            raise ValueError(f"y_sizes {sizes} do not partition {N} rows over {R} ranks")
        return sizes
    base = N // R
    if base < 1:
        raise ValueError(f"cannot split {N} rows over {R} ranks")
    return [base] * (R - 1) + [N - base * (R - 1)]


def local_row_range(Ny, arch):
    """jstart, jend of src/distributed_tripolar_grid.jl:47-48"""
    R = arch.ranks[1]
    n = local_sizes(Ny, R, arch.partition.y_sizes)
    r = arch.local_rank
    jstart = 1 + sum(n[:r])
    jend = Ny if r == R - 1 else sum(n[:r + 1])
    return jstart, jend


class TableWorkspace:
    """The workspace of tpg_build_grid with the 1-D tables of ONE geometry in it, kept with the grid that built them.  `key` is exactly
    what the tables depend on (include/tripolar_hip.h, TPG_BUILD_TABLES_VALID): global Nx, Ny, Hy, element type, southernmost latitude,
    north-poles latitude, radius -- and the device.  jstart / jend, Hx, Hz, Nz and first_pole_longitude do not enter.  A build whose key
    matches runs with the flag set (the ~9 us table kernel is skipped); any other build gets a NEW workspace, so the tables of a live
    grid are never overwritten.  `ready` orders a reusing build on another stream after the kernel that wrote the tables."""

    def __init__(self, key, tensor):
        self.key, self.tensor, self.ready, self.stream = key, tensor, None, None


# Table reuse is EXPLICIT.  with_halo and reconstruct_global_grid hand the old grid's workspace to the new build (`_tables_from`): that is
# always on.  Finding the tables of ANY live grid of the same geometry -- consecutive band builds of one geometry in one process (an
# emulated chain), repeated TripolarGrid() calls -- happens only inside a `share_tables()` scope: outside it a build never depends on which
# other grids happen to be alive (which path a test exercised used to depend on garbage-collection timing).
# the live workspaces (weak: a workspace lives exactly as long as a grid that holds it; several of one key may coexist)
_live_workspaces = weakref.WeakSet()
_live_lock = threading.Lock()
_share = threading.local()


class share_tables:
    """with share_tables(): builds in this thread may take the 1-D tables of any live grid of the same geometry (same Nx, Ny, Hy, element
    type, latitudes, radius, device) instead of recomputing them -- e.g. the bands of an emulated latitude-band chain built one after the
    other.  Results are identical either way; `grid.tables_reused` says which way a grid was built."""

    def __enter__(self):
        _share.depth = getattr(_share, "depth", 0) + 1
        return self

    def __exit__(self, *exc):
        _share.depth -= 1
        return False


def table_key(Nx, Ny, Hy, dtype, southernmost_latitude, north_poles_latitude, radius, device):
    return (int(Nx), int(Ny), int(Hy), _lib.ft_of(dtype), float(southernmost_latitude), float(north_poles_latitude), float(radius),
            str(device))


def _build(arch, dtype, size, halo, southernmost_latitude, radius, z, north_poles_latitude,
           first_pole_longitude, jstart, jend, topology_y, global_size=None, tables_from=None):
    Nx, Ny, Nz = (int(s) for s in size)
    Hx, Hy, Hz = (int(h) for h in halo)
    if Nx % 2 == 1:
        # ArgumentError of src/tripolar_grid.jl:81-83 (raised before touching the device)
        raise ValueError("The number of cells in the longitude dimension should be even!")
    device = arch.device
    lib = _lib.lib()
    p = _lib.TpgParams(Nx, Ny, Nz, Hx, Hy, Hz, float(southernmost_latitude), float(north_poles_latitude),
                       float(first_pole_longitude), float(radius), _lib.ft_of(dtype), jstart, jend, 0)
    rows = jend - jstart + 1 + 2 * Hy
    key = table_key(Nx, Ny, Hy, dtype, southernmost_latitude, north_poles_latitude, radius, device)
    with torch.cuda.device(device):
        arrays = {n: torch.empty((rows, Nx + 2 * Hx), dtype=dtype, device=device) for n in ARRAY_NAMES}
        nbytes = max(int(lib.tpg_build_grid_workspace_bytes(C.byref(p))), 256)
        stream = torch.cuda.current_stream(device)
        # tables of the same geometry already in a live workspace?  (the grid we were derived from, else any live grid's)
        ws = tables_from if (tables_from is not None and tables_from.key == key) else None
        if ws is None and getattr(_share, "depth", 0) > 0:
            with _live_lock:
                ws = next((w for w in list(_live_workspaces) if w.key == key and w.ready is not None and w.tensor.numel() >= nbytes), None)
        reuse = ws is not None and ws.tensor.numel() >= nbytes and ws.ready is not None
        if reuse:
            p.reserved = _lib.TPG_BUILD_TABLES_VALID
            if stream.cuda_stream != ws.stream:
                stream.wait_event(ws.ready)              # the tables were written on another stream (same stream: its order suffices)
        else:
            ws = TableWorkspace(key, torch.empty(nbytes, dtype=torch.uint8, device=device))
        out = _lib.ptr_table([arrays[n] for n in ARRAY_NAMES])
        _lib.check(lib.tpg_build_grid(C.byref(p), out, ws.tensor.data_ptr(), ws.tensor.numel(), C.c_void_p(stream.cuda_stream)))
        # the workspace must outlive the asynchronous kernels: tie its release to the stream
        ws.tensor.record_stream(stream)
        if not reuse:
            ws.ready, ws.stream = torch.cuda.Event(), stream.cuda_stream
            ws.ready.record(stream)
            with _live_lock:
                _live_workspaces.add(ws)
        Lz, zf, zc = _z_coordinate(z, Nz, Hz, dtype, device)
    ny = jend - jstart + 1
    return OrthogonalSphericalShellGrid(
        architecture=arch, Nx=Nx, Ny=ny, Nz=Nz, Hx=Hx, Hy=Hy, Hz=Hz, Lz=Lz, arrays=arrays,
        z_faces=zf, z_centers=zc, radius=float(radius),
        conformal_mapping=Tripolar(north_poles_latitude, first_pole_longitude, southernmost_latitude),
        topology=(PeriodicTopology, topology_y, Bounded), dtype=dtype,
        global_size=global_size, jrange=(jstart, jend) if global_size else None, z_spec=z,
        workspace=ws, tables_reused=reuse)


def TripolarGrid(arch=None, FT=torch.float64, *, size, southernmost_latitude=-80, halo=(4, 4, 4),
                 radius=R_Earth, z=(0, 1), north_poles_latitude=55, first_pole_longitude=70, _tables_from=None):
    """TripolarGrid(arch, FT; size, southernmost_latitude = -80, halo = (4, 4, 4), radius = R_Earth,
                 z = (0, 1), north_poles_latitude = 55, first_pole_longitude = 70)

    Reference: src/tripolar_grid.jl:59-333 (serial) and src/distributed_tripolar_grid.jl:24-110
    (arch::Distributed: latitude bands, x-partitioning rejected).  Returns an
    OrthogonalSphericalShellGrid{Periodic, RightConnected, Bounded} whose 20 metric arrays live in
    HBM; on a distributed architecture the rank's band jstart-Hy:jend+Hy is evaluated directly
    (the reference builds the whole globe on every rank and slices it).

    `_tables_from` (not a reference keyword; with_halo / reconstruct_global_grid pass it) names a TableWorkspace whose 1-D tables may be
    reused if they are of this geometry; without it the tables are computed -- unless the call sits inside a `share_tables()` scope, where a
    live grid of the same geometry is found by key.  Results are identical either way.
    """
    arch = GPU() if arch is None else arch
    dtype = _torch_dtype(FT)
    if getattr(arch, "is_distributed", False):
        workers = arch.ranks
        if workers[0] != 1:
            # src/distributed_tripolar_grid.jl:28-31
            raise ValueError("The tripolar grid is supported only on a Y-partitioning configuration")
        Nx, Ny, Nz = size
        jstart, jend = local_row_range(Ny, arch)
        LY = RightConnected if arch.local_rank == 0 else FullyConnected      # :75
        return _build(arch, dtype, size, halo, southernmost_latitude, radius, z, north_poles_latitude,
                      first_pole_longitude, jstart, jend, LY, global_size=tuple(size), tables_from=_tables_from)
    Nx, Ny, Nz = size
    return _build(arch, dtype, size, halo, southernmost_latitude, radius, z, north_poles_latitude,
                  first_pole_longitude, 1, Ny, RightConnected, tables_from=_tables_from)


def x_domain(grid):
    """x_domain(grid::TRG) = 0, 360   (src/tripolar_grid_extensions.jl:20)"""
    return 0, 360


def y_domain(grid):
    """y_domain(grid::TRG) = minimum(parent(grid.φᶠᶠᵃ)), 90   (:21)"""
    g = getattr(grid, "underlying_grid", grid)
    return float(g.arrays["phi_ff"].min()), 90


def with_halo(new_halo, old_grid):
    """with_halo(new_halo, grid)  (src/with_halo.jl:5-44): re-run the constructor from the stored
    Tripolar parameters with a different halo.  The old grid's 1-D tables are reused (TPG_BUILD_TABLES_VALID) when only Hx / Hz change;
    a new Hy -- or, distributed method, the default radius replacing a custom one -- changes the table key and recomputes them."""
    cm = old_grid.conformal_mapping
    kw = dict(z=old_grid.z_spec, halo=new_halo, north_poles_latitude=cm.north_poles_latitude,
              first_pole_longitude=cm.first_pole_longitude, southernmost_latitude=cm.southernmost_latitude,
              _tables_from=old_grid.workspace)
    if old_grid.global_size:
        # distributed method (src/with_halo.jl:25-44) does not forward `radius` (reference quirk kept)
        return TripolarGrid(old_grid.architecture, old_grid.dtype, size=old_grid.global_size, **kw)
    return TripolarGrid(old_grid.architecture, old_grid.dtype, size=old_grid.size, radius=old_grid.radius, **kw)


def reconstruct_global_grid(grid):
    """reconstruct_global_grid(grid::DistributedTripolarGrid)  (src/distributed_tripolar_grid.jl:201-226); the band's 1-D tables are
    those of the globe (they are indexed by global row), so the global build reuses them"""
    cm = grid.conformal_mapping
    return TripolarGrid(child_architecture(grid.architecture), grid.dtype, halo=grid.halo_size, _tables_from=grid.workspace,
                        size=grid.global_size if grid.global_size else grid.size, z=grid.z_spec,
                        north_poles_latitude=cm.north_poles_latitude,
                        first_pole_longitude=cm.first_pole_longitude,
                        southernmost_latitude=cm.southernmost_latitude)
