"""bench.py's output contract: exactly one JSON line on stdout with the driver's keys, the `roofline`
object of the dominant HBM-bound kernel and the `cpu_baseline` object (bounded oracle sample)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_prints_one_contract_line(gpu):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1"],
                       cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for key, typ in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                     ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str),
                     ("config", dict), ("roofline", dict), ("cpu_baseline", dict)):
        assert isinstance(d[key], typ), (key, d[key])
    assert d["vs_baseline"] is None and d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1
    assert d["unit"] == "cells/s" and d["dtype"] == "f64" and d["scaling"] == "weak" and d["higher_is_better"] is True
    assert "workload" in d["config"] and "model" not in d["config"]
    # value = cells of one step / step time
    assert abs(d["value"] - 3600 * 1800 / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and 0.2 < r["frac"] < 1.0
    assert r["algorithmic_bytes_per_launch"] == 73440000 and (r["traffic"] is None or r["traffic"] >= 0.95 * 73440000)
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["unit"] == "cells/s" and c["value"] > 1e5 and "oracle" in c["sample"]
    assert d["value"] > 100 * c["value"]            # sanity: the HIP path is the thing measured, not the oracle
    a = c["all_cores"]                               # single thread AND all cores, core count stated
    assert a["cores"] >= 1 and a["nproc"] >= a["cores"] and a["value"] > 0
    # cold and warm zipper launch durations, and the same-shape copy ceiling beside them
    assert 0 < d["zipper_warm_ms"] <= d["zipper_cold_dirty_ms"] * 1.5 and d["zipper_cold_ms"] > 0 and d["zipper_copy_ceiling_ms"] > 0
    assert r["copy_ceiling_ms"] == d["zipper_copy_ceiling_ms"]
    assert d["config2_quarter_degree_build"]["cells_per_s"] > 1e9
    assert "prewarm_steps" not in d                  # exactly W warm-up steps
    assert d["periodic_x_ms"] > 0 and d["exchange_ms"] is None and d["periodic_x"]["line_bytes"] == 3 * d["periodic_x"]["algorithmic_bytes"]
    # config 5: the fills of one baroclinic step at 1/24 degree x 100 levels
    fs = d["fill_step"]
    assert "skipped" in fs or (fs["fill3d_us"] > 0 and fs["substep_fills_us"] > 0 and fs["substeps"] == 30
                               and abs(fs["total_us"] - fs["fill3d_us"] - fs["substep_fills_us"]) < 1e-6 and fs["fields_GB"] > 160)


def test_bench_two_ranks_rehearsal(gpu):
    """The N > 1 code path of bench.py (latitude bands, zipper on the north rank, seam exchange on a side stream beside the
    build) as a FRESH child process: two ranks on this one GPU, seam messages host-staged over gloo (RCCL refuses two ranks on
    one device; the RCCL leg itself is tests/test_gpu_exchange.py).  Timings of such a run mean nothing; the contract does."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, TPG_BENCH_REHEARSE="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"]
    p = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert d["config"]["global_size"] == [3600, 3600, 75] and d["config"]["parallelism"] == "latitude-bands x2"
    assert isinstance(d["exchange_ms"], float) and d["exchange_ms"] > 0 and "gloo" in d["exchange_transport"]
    assert d["seam_GBps_per_direction"] > 0 and d["overlap"] is not None
    assert abs(d["value"] - 2 * 3600 * 1800 / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    assert "cpu_baseline" not in d and "fill_step" not in d           # rank 0 at N = 1 only
