"""orthogonalsphericalshellgrids.jl_amd -- MI355X-native TripolarGrid metric precompute and Zipper
halo fill behind the surface of CliMA/OrthogonalSphericalShellGrids.jl.

The reference exports exactly `TripolarGrid` and `ZipperBoundaryCondition`
(src/OrthogonalSphericalShellGrids.jl:4); the other names are the Oceananigans-side pieces a caller
needs around them (Field constructors, fill_halo_regions!, locations, architectures).
All numerics run in hand-written HIP kernels (csrc/, C ABI in include/tripolar_hip.h).
"""
from .boundary_conditions import (BoundaryCondition, Center, Face, FieldBoundaryConditions, Zipper,
                                  ZipperBoundaryCondition, PeriodicBoundaryCondition, bc_str,
                                  apply_y_north_bc, regularize_field_boundary_conditions, sign,
                                  validate_boundary_condition_location, is_zipper)
from .grids import (CPU, GPU, convert_to_0_360, Distributed, Partition, OrthogonalSphericalShellGrid, R_Earth, Tripolar,
                    TripolarGrid, is_tripolar, local_row_range, local_sizes, reconstruct_global_grid, share_tables,
                    with_halo, x_domain, y_domain, RightConnected, FullyConnected, Bounded,
                    PeriodicTopology)
from .fields import (CenterField, Field, HaloFillPlan, XFaceField, YFaceField, ZFaceField, fill_halo_regions,
                     halo_fill_plan, interior, set_)
from .distributed import (LoopbackMailbox, PendingExchange, RcclComm, exchange_plan, exchange_y_halos, torch_distributed_transport)
from .geometry import convert_to_latlong_frame, convert_to_native_frame, nonorthogonality_angle

__all__ = ["TripolarGrid", "ZipperBoundaryCondition"]
