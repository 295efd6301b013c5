"""bench.py's start-up at BASELINE config 4's own N, without a device: the launcher, the rendezvous, the band layout and the seam
pairing of `python bench.py --gpus 8` (src/distributed_tripolar_grid.jl:36-49,75,143-147; examples/distributed_bickley_jet.jl:8-25),
and the one-line diagnostic of a node that shows fewer devices than ranks.  The 8-rank run WITH kernels cannot be rehearsed on a
one-GPU box (its process guard allows 6 processes on the card: tests/test_gpu_bench_contract.py runs N <= 4 beside the test process, N = 6 was run by hand);
these tests run here."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _env(**kw):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", **kw)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    return env


def test_chain_layout_of_config4():
    import bench
    L = [bench.chain_layout(8, r) for r in range(8)]
    assert [x["ny"] for x in L] == [225] * 8 and [x["seams"] for x in L] == [1, 2, 2, 2, 2, 2, 2, 1]
    assert [(x["jstart"], x["jend"]) for x in L] == [(225 * r + 1, 225 * r + 225) for r in range(8)]
    assert [x["north_is_zipper"] for x in L] == [False] * 7 + [True]
    assert [(x["south_peer"], x["north_peer"]) for x in L] == [(-1, 1)] + [(r - 1, r + 1) for r in range(1, 7)] + [(6, -1)]
    assert all(x["gsize"] == (3600, 1800, 75) and x["strong"] for x in L)
    w = bench.chain_layout(4, 2, "weak")
    assert w["ny"] == 1800 and w["gsize"] == (3600, 7200, 75) and (w["jstart"], w["jend"]) == (3601, 5400)
    lb = bench.chain_layout(1, 0, "strong", (8, 7))                   # --loopback: band 7 of 8 on one rank, its peer is the rank itself
    assert (lb["south_peer"], lb["north_peer"], lb["north_is_zipper"], lb["ny"]) == (0, -1, True, 225)
    one = bench.chain_layout(1, 0)
    assert not one["chain"] and one["ny"] == 1800 and one["seams"] == 0 and one["north_is_zipper"]
    with pytest.raises(SystemExit):
        bench.chain_layout(7, 0)                                      # 1800 % 7 != 0: the remainder rule is unpinned, refused


def test_bench_eight_ranks_plan_rehearsal():
    """`python bench.py --gpus 8` (no launcher: bench.py starts its own 8 workers) under TPG_BENCH_REHEARSE=plan: every worker goes through
    the real start-up (environment, gloo rendezvous, chain_layout, local_row_range on Distributed(Partition(y = 8))), swaps seam-shaped
    host messages with its neighbours through the product's torch_distributed_transport and checks the tags it received."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1"], cwd=ROOT,
                       env=_env(TPG_BENCH_REHEARSE="plan"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["event"] == "bench_plan" and "metric" not in d and "value" not in d            # a plan, never a measurement
    assert d["n_gpus"] == 8 and d["rows_per_rank"] == 225 and d["global_size"] == [3600, 1800, 75]
    pr = d["per_rank"]
    assert [r["rank"] for r in pr] == list(range(8)) and [r["seams"] for r in pr] == [1, 2, 2, 2, 2, 2, 2, 1]
    assert [r["zipper"] for r in pr] == [False] * 7 + [True] and [r["rows"] for r in pr] == [[225 * r + 1, 225 * r + 225] for r in range(8)]
    assert all(r["seam_tags_ok"] for r in pr)
    assert pr[0]["seam_message_bytes_per_direction"] == 4 * 3608 * 4 * 83 * 8
    assert d["hbm_bytes_per_rank"] < 4e9                                                   # 2.9 GB of 288 per rank


def test_bench_too_few_devices_is_one_readable_line():
    """A node that shows 4 devices to an 8-rank job (stubbed count; this container shows 0): every worker leaves with status 7 before any
    device call, rank 0 prints ONE JSON diagnostic on stderr, the launcher passes the status on and prints no contract line."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1"], cwd=ROOT,
                       env=_env(TPG_BENCH_TEST_DEVICE_COUNT="4"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert p.returncode == 7 and not p.stdout.strip(), (p.returncode, p.stdout[-500:], p.stderr[-2000:])
    diag = [json.loads(l) for l in p.stderr.splitlines() if l.startswith("{") and "too_few_devices" in l]
    assert len(diag) == 1 and diag[0]["visible"] == 4 and diag[0]["requested"] == 8 and diag[0]["rank"] == 0
    assert "Traceback" not in p.stderr and "invalid device ordinal" not in p.stderr
