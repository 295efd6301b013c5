"""Zipper boundary condition metadata: host-side mirror of
src/zipper_boundary_condition.jl:8-64 and src/tripolar_grid_extensions.jl:25-53 of the reference.
Pure metadata -- the fold itself is the HIP kernel behind tpg_zipper_fill."""
from dataclasses import dataclass
from typing import Any, Optional


class Center:
    """Oceananigans.Grids.Center (cell-centre location)."""


class Face:
    """Oceananigans.Grids.Face (cell-interface location)."""


class AbstractBoundaryConditionClassification:
    pass


class Zipper(AbstractBoundaryConditionClassification):
    """struct Zipper <: AbstractBoundaryConditionClassification   (zipper_boundary_condition.jl:8)"""

    def __repr__(self):
        return "Zipper()"


class Periodic(AbstractBoundaryConditionClassification):
    def __repr__(self):
        return "Periodic()"


class HaloCommunication(AbstractBoundaryConditionClassification):
    """Oceananigans' DistributedCommunication classification: halo rows come from a neighbour rank."""

    def __repr__(self):
        return "HaloCommunication()"


@dataclass(frozen=True)
class BoundaryCondition:
    classification: Any
    condition: Any = None


def ZipperBoundaryCondition(sign=1):
    """ZipperBoundaryCondition(sign = 1) = BoundaryCondition(Zipper(), sign)   (:52)"""
    return BoundaryCondition(Zipper(), sign)


def PeriodicBoundaryCondition():
    return BoundaryCondition(Periodic(), None)


def HaloCommunicationBoundaryCondition(from_rank, to_rank):
    return BoundaryCondition(HaloCommunication(), (from_rank, to_rank))


def is_zipper(bc):
    """bc isa ZBC  (const ZBC = BoundaryCondition{<:Zipper}, :54)"""
    return isinstance(bc, BoundaryCondition) and isinstance(bc.classification, Zipper)


def bc_str(bc):
    """bc_str(zip::ZBC) = "Zipper"   (:56)"""
    if is_zipper(bc):
        return "Zipper"
    return type(bc.classification).__name__ if isinstance(bc, BoundaryCondition) else str(bc)


def validate_boundary_condition_location(bc, loc, side):
    """(:58-62) a Zipper classification is valid on the north side only."""
    cls = bc.classification if isinstance(bc, BoundaryCondition) else bc
    if isinstance(cls, Zipper) and side != "north":
        name = loc.__name__ if isinstance(loc, type) else type(loc).__name__
        raise ValueError(f"Cannot specify {side} boundary condition {cls!r} on a field at {name} (north only)!")
    return None


def apply_y_north_bc(Gc, loc, bc, *args):
    """@inline apply_y_north_bc!(Gc, loc, ::ZBC, args...) = nothing   (:64)"""
    return None


@dataclass
class FieldBoundaryConditions:
    west: Optional[BoundaryCondition] = None
    east: Optional[BoundaryCondition] = None
    south: Optional[BoundaryCondition] = None
    north: Optional[BoundaryCondition] = None
    bottom: Optional[BoundaryCondition] = None
    top: Optional[BoundaryCondition] = None
    immersed: Optional[BoundaryCondition] = None

    def validate(self, loc):
        for side in ("west", "east", "south", "north", "bottom", "top"):
            bc = getattr(self, side)
            if bc is not None:
                axis = {"west": 0, "east": 0, "south": 1, "north": 1, "bottom": 2, "top": 2}[side]
                validate_boundary_condition_location(bc, loc[axis], side)


def sign(LX, LY):
    """Location -> zipper sign table (tripolar_grid_extensions.jl:49-53):
    fields on edges are signed vectors (-1), fields on nodes and centres are scalars (+1)."""
    if LX is Face and LY is Center:
        return -1
    if LX is Center and LY is Face:
        return -1
    return 1


_ASSUMED_LOCATIONS = {"u": (Face, Center, Center), "v": (Center, Face, Center), "w": (Center, Center, Face)}


def assumed_field_location(field_name):
    """Oceananigans.BoundaryConditions.assumed_field_location [recalled]: u, v, w are face fields."""
    return _ASSUMED_LOCATIONS.get(field_name, (Center, Center, Center))


def regularize_field_boundary_conditions(bcs, grid, field_name, prognostic_names=None):
    """tripolar_grid_extensions.jl:25-44 / distributed_tripolar_grid.jl:129-155.
    north = ZipperBoundaryCondition(-1 for :u, :v else +1); on a distributed grid only the last
    rank gets the zipper, the others keep a neighbour-communication north side."""
    sgn = -1 if field_name in ("u", "v") else 1
    north = ZipperBoundaryCondition(sgn)
    arch = getattr(grid, "architecture", None)
    if getattr(arch, "is_distributed", False) and arch.local_rank != arch.ranks[1] - 1:
        north = HaloCommunicationBoundaryCondition(arch.local_rank, arch.local_rank + 1)
    return FieldBoundaryConditions(west=bcs.west, east=bcs.east, south=bcs.south, north=north,
                                   bottom=bcs.bottom, top=bcs.top, immersed=bcs.immersed)
