#!/usr/bin/env python3
"""Regenerates tests/golden/restatement_*.npz from the oracle (oracle/tpg_oracle.c).

These vectors are RESTATEMENT-DERIVED, not reference-derived: no Julia toolchain exists in the
build container, so the reference cannot be run.  They pin the oracle (and through
it the HIP path) against regressions; the reference-held known answers are in reference_kats.json.
Usage: python tests/golden/make_golden.py
"""
import os, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import oracle

CASES = {
    "restatement_60x30_f64": dict(size=(60, 30, 1), dtype=np.float64),
    "restatement_4x5_f32_poles75_35": dict(size=(4, 5, 1), dtype=np.float32, first_pole_longitude=75, north_poles_latitude=35),
    "restatement_20x12_h321_band5_8": dict(size=(20, 12, 1), dtype=np.float64, halo=(3, 2, 1), jstart=5, jend=8),
}
if __name__ == "__main__":
    for name, kw in CASES.items():
        g = oracle.build_grid(**kw)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **g)
        print(name, {k: v.shape for k, v in list(g.items())[:1]})
