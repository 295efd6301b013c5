#!/usr/bin/env python3
"""tools/band_halo5.py -- the LOCAL work of one latitude band's halo fill (BASELINE config 4: 3600 x 225 x 75, fields c/u/v/zeta, Float64 and Float32)
at the halo of the reference's distributed example, (5, 5, 5) (examples/distributed_bickley_jet.jl:23), with halo 4 measured by the same
method beside it: the periodic-x pass of a middle band, the whole local fill of the zipper band, and the seam pack / unpack.  Cold (a 1 GiB
read-only pass before every call), stream-event brackets around the C call, median of 10 after 2 dropped; per-byte ratios halo 5 / halo 4.
usage (GPU box): python tools/band_halo5.py [out.json]"""
import ctypes as C
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SPECS = [("c", 0, 0, 1), ("u", 1, 0, -1), ("v", 0, 1, -1), ("zeta", 1, 1, 1)]


def measure(torch, _lib, lib, tlib, dev, h, size=(3600, 225, 75), reps=12, dtype="f64"):
    from tools import testlib
    from bench_halo5 import fold_bytes, periodic_bytes
    nx, ny, nz = size
    geom = (nx, ny, nz, h, h, h)
    n = len(SPECS)
    tdt, ft, esz = (torch.float64, _lib.TPG_F64, 8) if dtype == "f64" else (torch.float32, _lib.TPG_F32, 4)
    fields = [torch.empty((nz + 2 * h, ny + 2 * h, nx + 2 * h), dtype=tdt, device=dev) for _ in SPECS]
    for fid, f in enumerate(fields):
        testlib.check(tlib.tpg_fill_synthetic(f.data_ptr(), 0xBA5D + fid, 12345.0, *geom, ft, None))
    pt = _lib.ptr_table(fields)
    xl = (C.c_int8 * n)(*[s[1] for s in SPECS]); yl = (C.c_int8 * n)(*[s[2] for s in SPECS]); sg = (C.c_int32 * n)(*[s[3] for s in SPECS])
    stream = _lib.current_stream_ptr(dev)
    elems = int(lib.tpg_y_halo_buffer_elems(n, *geom[:1], nz, h, h, h))
    buf = torch.empty(elems, dtype=tdt, device=dev)
    flush = torch.zeros(1 << 27, dtype=torch.float64, device=dev)

    def timed(call):
        ts = []
        for _ in range(reps):
            flush.sum()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); _lib.check(call()); e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        return statistics.median(ts[2:])

    out = {"halo": h,
           "middle_band_fill_us": timed(lambda: lib.tpg_fill_halo_regions(pt, n, xl, yl, sg, *geom, 0, ft, stream)),
           "zipper_band_fill_us": timed(lambda: lib.tpg_fill_halo_regions(pt, n, xl, yl, sg, *geom, 1, ft, stream)),
           "pack_north_us": timed(lambda: lib.tpg_pack_y_halo(pt, n, buf.data_ptr(), 1, *geom, ft, stream)),
           "unpack_north_us": timed(lambda: lib.tpg_unpack_y_halo(pt, n, buf.data_ptr(), 1, *geom, ft, stream)),
           "pack_south_us": timed(lambda: lib.tpg_pack_y_halo(pt, n, buf.data_ptr(), 0, *geom, ft, stream)),
           "periodic_bytes": periodic_bytes(ny, nz, (h, h, h), n, esz), "fold_bytes": fold_bytes(nx, nz, h, SPECS, esz),
           "message_bytes": elems * esz}
    out["zipper_band_bytes"] = out["periodic_bytes"] + out["fold_bytes"]
    del fields, buf, flush
    torch.cuda.empty_cache()
    return out


def main():
    import torch
    from orthogonalsphericalshellgrids.jl_amd import _lib
    from tools import testlib
    dev = torch.device("cuda:0")
    lib, tlib = _lib.lib(), testlib.lib()
    out = {"what": "local halo-fill work of one config-4 band (3600 x 225 x 75, c/u/v/zeta): halo 5 beside halo 4, cold, us"}
    for dtype in ("f64", "f32"):
        a, b = measure(torch, _lib, lib, tlib, dev, 4, dtype=dtype), measure(torch, _lib, lib, tlib, dev, 5, dtype=dtype)
        per_byte = lambda t, by: (b[t] / b[by]) / (a[t] / a[by])
        out[dtype] = {"halo4": a, "halo5": b,
                      "per_byte_halo5_over_halo4": {"middle_band_fill": per_byte("middle_band_fill_us", "periodic_bytes"),
                                                    "zipper_band_fill": per_byte("zipper_band_fill_us", "zipper_band_bytes"),
                                                    "pack_north": per_byte("pack_north_us", "message_bytes"),
                                                    "unpack_north": per_byte("unpack_north_us", "message_bytes"),
                                                    "pack_south": per_byte("pack_south_us", "message_bytes")}}
    text = json.dumps(out, indent=1)
    print(text)
    if len(sys.argv) > 1:
        open(sys.argv[1], "w").write(text + "\n")


if __name__ == "__main__":
    main()
