"""Geometry utilities over a TripolarGrid's arrays: host-side mirror of
compute_nonorthogonality_angle! (test/test_tripolar_grid.jl:8-34) and of convert_to_latlong_frame /
convert_to_native_frame (examples/convert_to_latlong_frame.jl:12-55).  The arithmetic runs in the HIP kernels of
csrc/tpg_geometry.hip."""
import torch

from . import _lib
from .boundary_conditions import Center
from .grids import is_tripolar


def _serial(grid):
    g = getattr(grid, "underlying_grid", grid)
    if not is_tripolar(g):
        raise TypeError("grid must be a TripolarGrid")
    return g


def nonorthogonality_angle(grid, immersed=None):
    """Angle (degrees, minus 90) between the two grid lines through every Face-Face node (i, j), i < Nx, j < Ny,
    as compute_nonorthogonality_angle! launched over (Nx-1, Ny-1) (test/test_tripolar_grid.jl:70); 0 elsewhere and
    where `immersed[j-1, i-1]` is true (immersed_cell, :28).  Returns a (Ny, Nx) Float64 tensor."""
    g = _serial(grid)
    dev = g.device
    angle = torch.empty((g.Ny, g.Nx), dtype=torch.float64, device=dev)
    mask = None
    if immersed is not None:
        mask = torch.as_tensor(immersed, device=dev).to(torch.uint8).contiguous()
        if tuple(mask.shape) != (g.Ny, g.Nx):
            raise ValueError(f"immersed must have shape {(g.Ny, g.Nx)}")
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().tpg_nonorthogonality_angle(
            g.arrays["lambda_ff"].data_ptr(), g.arrays["phi_ff"].data_ptr(), None if mask is None else mask.data_ptr(),
            angle.data_ptr(), g.Nx, g.Ny, g.Hx, g.Hy, _lib.ft_of(g.dtype), _lib.current_stream_ptr(dev)))
        if mask is not None:
            mask.record_stream(torch.cuda.current_stream(dev))
    return angle


def _convert(grid, u, v, to_native):
    from .fields import Field
    g = _serial(grid)
    for f in (u, v):
        if f.loc != (Center, Center, Center) or getattr(f.grid, "underlying_grid", f.grid) is not g:
            raise ValueError("frame conversion takes two (Center, Center, Center) fields of this grid")
    uo, vo = Field(u.loc, grid), Field(v.loc, grid)
    dev = g.device
    a = g.arrays
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().tpg_convert_frame(
            a["phi_cf"].data_ptr(), a["phi_fc"].data_ptr(), a["dy_cc"].data_ptr(), a["dx_cc"].data_ptr(),
            u.data.data_ptr(), v.data.data_ptr(), uo.data.data_ptr(), vo.data.data_ptr(), 1 if to_native else 0,
            u.Nx, u.Ny, u.Nz, u.Hx, u.Hy, u.Hz, _lib.ft_of(g.dtype), _lib.current_stream_ptr(dev)))
    return uo, vo


def convert_to_latlong_frame(grid, u, v):
    """(u, v) given along the grid's local axes -> (zonal, meridional) components, assuming local orthogonality
    (examples/convert_to_latlong_frame.jl:12-32): returns two new CenterFields (interior filled)."""
    return _convert(grid, u, v, False)


def convert_to_native_frame(grid, u, v):
    """inverse rotation (examples/convert_to_latlong_frame.jl:35-55)"""
    return _convert(grid, u, v, True)
