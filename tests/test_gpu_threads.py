"""include/tripolar_hip.h promises: calls are asynchronous on the given stream, re-entrant, thread-safe for distinct
streams; the only mutable state is a thread-local error string and immutable records published once.  Four host threads,
each on its own HIP stream, run grid builds and halo fills of different geometries at the same time; every result must be
what the same call gives alone (bit-exact vs the oracle), and an error raised on one thread must not leak its message
into another."""
import ctypes as C
import threading

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_concurrent_calls_on_distinct_streams(osg, oracle, gpu):
    lib = osg._lib.lib()
    jobs = [((60, 30, 2), (4, 4, 2), 1, 0, -1), ((128, 64, 3), (4, 4, 1), 0, 0, 1), ((250, 100, 1), (4, 4, 4), 0, 1, -1), ((64, 40, 2), (2, 3, 1), 1, 1, 1)]
    refs = [oracle.build_grid(size, halo=halo) for size, halo, *_ in jobs]
    results, errors = [None] * len(jobs), []
    barrier = threading.Barrier(len(jobs))

    def work(n):
        try:
            size, halo, xl, yl, sg = jobs[n]
            (Nx, Ny, Nz), (Hx, Hy, Hz) = size, halo
            torch.cuda.set_device(0)
            stream = torch.cuda.Stream()
            rng = np.random.default_rng(100 + n)
            h = rng.uniform(-1, 1, (Nz + 2 * Hz, Ny + 2 * Hy, Nx + 2 * Hx))
            want = h.copy()
            oracle.fill_halo_regions(want, xl, yl, sg, size, halo)
            barrier.wait()
            with torch.cuda.stream(stream):
                for rep in range(20):
                    grid = osg.TripolarGrid(osg.GPU(0), torch.float64, size=size, halo=halo)     # tpg_build_grid on this thread's stream
                    d = torch.from_numpy(h).to(gpu, non_blocking=False)
                    sp = C.c_void_p(stream.cuda_stream)
                    rc = lib.tpg_fill_halo_regions(osg._lib.ptr_table([d]), 1, (C.c_int8 * 1)(xl), (C.c_int8 * 1)(yl), (C.c_int32 * 1)(sg),
                                                   Nx, Ny, Nz, Hx, Hy, Hz, 1, 1, sp)
                    assert rc == 0
                    # a deliberate error on this thread: its message must be this thread's own
                    bad = lib.tpg_zipper_fill(osg._lib.ptr_table([d]), 1, (C.c_int8 * 1)(xl), (C.c_int8 * 1)(yl), (C.c_int32 * 1)(sg),
                                              Nx + 1 + 2 * n, Ny, Nz, Hx, Hy, Hz, 1, Nz, 1, sp)
                    assert bad == -2 and b"even" in lib.tpg_last_error()
                    stream.synchronize()
                    got = {k: getattr(grid, k).cpu().numpy() for k in ("lambda_ff", "dx_cc", "dy_fc", "az_ff")}
                    for k, v in got.items():
                        assert np.array_equal(v, refs[n][k], equal_nan=True), (n, rep, k)
                    assert np.array_equal(d.cpu().numpy(), want), (n, rep)
            results[n] = True
        except Exception as e:                                                  # noqa: BLE001
            errors.append(f"thread {n}: {type(e).__name__}: {e}")

    threads = [threading.Thread(target=work, args=(n,)) for n in range(len(jobs))]
    for t in threads:
        t.start()
    for t in threads:
        t.join(300)
    assert not errors, errors
    assert all(results)
