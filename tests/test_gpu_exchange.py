"""The C ABI's RCCL seam exchange on hardware.  A 1-GPU box cannot host two RCCL ranks (RCCL refuses two ranks on one
device), so the data path -- tpg_comm_init_rank, pack -> ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd -> unpack, and
the pack-free form sending the per-level seam windows directly -- runs on a communicator of ONE rank whose south and north
peers are the rank itself (tools/rccl_selftest.py), in a child process with a timeout (a mis-paired group would hang).
Two-rank protocol coverage: tests/test_distributed_gloo.py (world 2, 3 on CPU), tests/test_gpu_distributed.py (emulated
ranks, real kernels), tests/test_gpu_bench_contract.py::test_bench_two_ranks_rehearsal."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_rccl_seam_exchange_single_rank_loopback(gpu):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rccl_selftest.py")], cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-2000:])
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert d["ok"] and d["single_rank_chain_rc"] == 0 and d["from_torch"] == [0, 1]
    assert len(d["cases"]) == 10 and all(c["bit_exact"] for c in d["cases"])         # 5 geometries (the band at halo 4 and at halo 5) x packed / pack-free
    assert {c["packed"] for c in d["cases"]} == {True, False}
    assert [3600, 225, 75] in [c["size"] for c in d["cases"]]                       # BASELINE config 4's band geometry (ny = 225)
    # the whole band fill through ONE C call (tpg_fill_halo_regions_distributed_peers): a middle band and the zipper band
    forms = [("middle", False), ("north", False), ("middle", True), ("north", True)]
    assert [(tuple(c["halo"]), c["band"], c["pipelined"]) for c in d["distributed_fill"]] == \
        [(h, b, p) for h in ((4, 4, 2), (5, 5, 5)) for b, p in forms]                # the second halo: examples/distributed_bickley_jet.jl:23
    assert all(c["bit_exact"] and c["rc"] == 0 for c in d["distributed_fill"])
    # the pipelined packed exchange (stages of 1, 2, 3, all fields; one stream and two; two seams / south only / north only; called twice
    # on the same buffers) delivers exactly the monolithic result, incl. config 4's 3600 x 225 x 75 band
    assert d["pipelined"]["cases"] >= 50 and d["pipelined"]["all_bit_exact"], d["pipelined"]["failed"][:3]
    assert set(d["exchange_cost_config4_band_loopback"]) == {"monolithic", "pipelined_1", "pipelined_2"}
    assert d["two_streams_own_buffers_bit_exact"] is True
    # a failure injected after the RCCL group of stage 1 (test library): TPG_ERR_RCCL comes back, `stream` is nevertheless ordered after
    # everything on comm_stream (post-condition of include/tripolar_hip.h on error returns), and the same buffers then serve a good call
    lf = d["late_failure"]
    assert lf["rcs"] == [-7] * 5 and all(lf["comm_stream_idle_after_stream_sync"]) and lf["rc_after"] == 0 and lf["reuse_bit_exact"]
    assert len(lf["messages"]) == 1 and "injected failure after the RCCL group of stage 1" in lf["messages"][0]
    # the ordering events belong to their host thread (thread_local owner): threads that exchange and end leave the others untouched
    assert d["event_pool_threads"] == {"thread_rcs": [0, 0, 0], "main_rc_after": 0}
    # a capturing stream is refused instead of stalling, and the capture survives the refusal
    f = d["capture_fence"]
    assert f["rc"] == -5 and f["rc_pipelined"] == -5 and "captured" in f["message"] and f["periodic_rc_in_capture"] == 0 and f["replay_bit_exact"]


def test_exchange_argument_errors(osg, gpu):
    import ctypes as C
    import torch
    lib = osg._lib.lib()
    d = torch.zeros((1, 12, 12), dtype=torch.float64, device=gpu)
    ptr = osg._lib.ptr_table([d])
    assert lib.tpg_halo_exchange_y(None, 0, 2, ptr, 1, None, None, None, None, 4, 4, 1, 4, 4, 0, 1, None) == -1      # null communicator
    assert lib.tpg_halo_exchange_y(1 << 20, 2, 2, ptr, 1, None, None, None, None, 4, 4, 1, 4, 4, 0, 1, None) == -3   # rank outside the chain
    buf = torch.zeros(4 * 12, dtype=torch.float64, device=gpu)
    # packed exchange with a peer on the north side but no north buffers: refused before any RCCL call
    assert lib.tpg_halo_exchange_y(1 << 20, 0, 2, ptr, 1, buf.data_ptr(), None, buf.data_ptr(), None, 4, 4, 1, 4, 4, 0, 1, None) == -1
    assert b"message buffer" in lib.tpg_last_error()
