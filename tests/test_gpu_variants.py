"""Every kernel variant kept in the library (the simple thread-per-cell build kernel, the two marching
kernels, the LDS-tile default; the zipper's row / column / streaming forms) must produce identical bits: they differ only in how the same arithmetic is
scheduled.  The TPG_* knobs are read per call."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
KNOBS = ("TPG_CELLS_VARIANT", "TPG_BUILD_NT", "TPG_CELLS_STRIP", "TPG_ZIPPER_VARIANT")


@pytest.fixture
def knob():
    saved = {k: os.environ.get(k) for k in KNOBS}
    yield os.environ
    for k, v in saved.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v


@pytest.mark.parametrize("kw", [dict(size=(250, 100, 1)),
                                dict(size=(128, 64, 1), halo=(3, 2, 1), north_poles_latitude=65),
                                dict(size=(64, 20, 1), dtype=torch.float32)], ids=["250x100", "128x64", "64x20-f32"])
def test_build_kernel_variants_agree(osg, gpu, knob, kw):
    kw = dict(kw)
    dtype = kw.pop("dtype", torch.float64)
    results = {}
    for variant, nt, strip in ((2, 1, 0), (1, 1, 0), (0, 1, 0), (3, 1, 0), (3, 0, 0), (2, 0, 0), (2, 1, 7), (1, 1, 5)):
        knob["TPG_CELLS_VARIANT"], knob["TPG_BUILD_NT"], knob["TPG_CELLS_STRIP"] = str(variant), str(nt), str(strip)
        g = osg.TripolarGrid(osg.GPU(0), dtype, **kw)
        results[(variant, nt, strip)] = {n: getattr(g, n).cpu().numpy() for n in osg._lib.ARRAY_NAMES}
    ref = results[(0, 1, 0)]
    for key, arrs in results.items():
        for n, a in arrs.items():
            assert np.array_equal(a, ref[n], equal_nan=True), (key, n)


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32], ids=["f64", "f32"])
def test_zipper_kernel_variants_agree(osg, gpu, knob, dtype):
    size, halo = (256, 40, 4), (4, 4, 2)
    grid = osg.TripolarGrid(osg.GPU(0), dtype, size=size, halo=halo)
    rng = np.random.default_rng(4)
    specs = [(xl, yl, sg) for xl in (0, 1) for yl in (0, 1) for sg in (1, -1)]
    hosts = [rng.uniform(-1, 1, (4 + 4, 40 + 8, 256 + 8)).astype(np.float64 if dtype == torch.float64 else np.float32) for _ in specs]
    outs = {}
    for variant in (0, 1, 2, 3, 4):
        knob["TPG_ZIPPER_VARIANT"] = str(variant)
        fs = []
        for (xl, yl, sg), h in zip(specs, hosts):
            loc = (osg.Face if xl else osg.Center, osg.Face if yl else osg.Center, osg.Center)
            f = osg.Field(loc, grid, boundary_conditions=osg.FieldBoundaryConditions(north=osg.ZipperBoundaryCondition(sg)))
            f.data.copy_(torch.from_numpy(h))
            fs.append(f)
        osg.fill_halo_regions(fs)
        outs[variant] = [f.data.cpu().numpy() for f in fs]
    for variant, arrs in outs.items():
        for a, b in zip(arrs, outs[0]):
            assert np.array_equal(a, b), variant
