"""Host-side mirror of the reference's dispatch glue: zipper sign policy, validation, partition rule."""
import pytest


def test_zipper_boundary_condition_type(osg):
    bc = osg.ZipperBoundaryCondition()
    assert isinstance(bc.classification, osg.Zipper) and bc.condition == 1     # zipper_boundary_condition.jl:52
    assert osg.ZipperBoundaryCondition(-1).condition == -1
    assert osg.bc_str(bc) == "Zipper"                                          # :56
    assert osg.apply_y_north_bc(None, None, bc) is None                        # :64


@pytest.mark.parametrize("side", ["south", "west", "east", "top", "bottom"])
def test_zipper_is_north_only(osg, side):
    z = osg.ZipperBoundaryCondition()
    assert osg.validate_boundary_condition_location(z, osg.Center, "north") is None
    with pytest.raises(ValueError, match="north only"):                        # :58-62
        osg.validate_boundary_condition_location(z, osg.Face, side)


def test_sign_table(osg):
    C, F = osg.Center, osg.Face                                                # tripolar_grid_extensions.jl:49-53
    assert osg.sign(F, F) == 1 and osg.sign(C, C) == 1
    assert osg.sign(F, C) == -1 and osg.sign(C, F) == -1
    assert osg.sign(None, C) == 1


def test_regularize_by_name(osg):
    class G:  # serial grid stand-in
        architecture = None
    bcs = osg.FieldBoundaryConditions()
    for name, s in (("u", -1), ("v", -1), ("c", 1), ("T", 1), ("w", 1)):       # :32
        north = osg.regularize_field_boundary_conditions(bcs, G(), name).north
        assert osg.is_zipper(north) and north.condition == s


def test_regularize_distributed_only_last_rank(osg):
    class G:
        def __init__(self, r):
            self.architecture = osg.Distributed(osg.GPU(), osg.Partition(y=4), local_rank=r)
    bcs = osg.FieldBoundaryConditions()
    for r in range(4):                                                         # distributed_tripolar_grid.jl:143-147
        north = osg.regularize_field_boundary_conditions(bcs, G(r), "u").north
        assert osg.is_zipper(north) == (r == 3)


def test_partition_rule(osg):
    assert osg.local_sizes(1800, 8) == [225] * 8                               # config 4
    assert osg.local_sizes(10, 3) == [3, 3, 4]
    assert osg.local_sizes(10, 2, [4, 6]) == [4, 6]
    with pytest.raises(ValueError):
        osg.local_sizes(10, 2, [4, 5])
    arch = lambda r: osg.Distributed(osg.GPU(), osg.Partition(y=8), local_rank=r)
    assert osg.local_row_range(1800, arch(0)) == (1, 225)                      # :47-48
    assert osg.local_row_range(1800, arch(7)) == (1576, 1800)
    assert osg.local_row_range(1801, arch(7)) == (1576, 1801)                  # last rank always ends at Ny


def test_x_partitioning_is_rejected(osg):
    arch = osg.Distributed(osg.GPU(), osg.Partition(x=2, y=1), local_rank=0)
    with pytest.raises(ValueError, match="Y-partitioning"):                    # :28-31
        osg.TripolarGrid(arch, size=(60, 30, 1))


def test_odd_longitude_rejected_before_device(osg):
    with pytest.raises(ValueError, match="should be even"):
        osg.TripolarGrid(size=(61, 30, 1))


def test_exchange_plan(osg):
    from orthogonalsphericalshellgrids.jl_amd.distributed import SeamMessage, NORTH, SOUTH
    assert osg.exchange_plan(0, 1) == []
    assert osg.exchange_plan(0, 4) == [SeamMessage(NORTH, 1)]
    assert osg.exchange_plan(2, 4) == [SeamMessage(NORTH, 3), SeamMessage(SOUTH, 1)]
    assert osg.exchange_plan(3, 4) == [SeamMessage(SOUTH, 2)]                  # north side of the last rank = zipper


def test_unicode_property_aliases(osg):
    from orthogonalsphericalshellgrids.jl_amd.grids import _UNICODE_ALIASES
    import unicodedata
    assert _UNICODE_ALIASES[unicodedata.normalize("NFKC", "Δxᶜᶜᵃ")] == "dx_cc"
    assert _UNICODE_ALIASES[unicodedata.normalize("NFKC", "φᶠᶜᵃ")] == "phi_fc"
    assert _UNICODE_ALIASES[unicodedata.normalize("NFKC", "Azᶜᶠᵃ")] == "az_cf"
    assert len(_UNICODE_ALIASES) == 20


def test_halo_fill_plan_is_exported_and_validates_fields():
    """HaloFillPlan (the reusable form of fill_halo_regions!) is part of the host surface; building one
    touches no device memory, so its validation can be checked on CPU"""
    import orthogonalsphericalshellgrids.jl_amd as osg
    assert callable(osg.halo_fill_plan) and osg.HaloFillPlan is not None
    plan = osg.halo_fill_plan([])
    assert plan() is None and plan.fields == []


def test_loopback_mailbox_delivers_in_order(osg):
    """two-phase transport of the emulated-rank tests: every posted message reaches the peer's receive buffer, in posting order"""
    import torch
    from orthogonalsphericalshellgrids.jl_amd.distributed import NORTH, SOUTH
    box = osg.LoopbackMailbox()
    R = 3
    plans = [osg.exchange_plan(r, R) for r in range(R)]
    assert [[(m.side, m.peer) for m in p] for p in plans] == [[(NORTH, 1)], [(NORTH, 2), (SOUTH, 0)], [(SOUTH, 1)]]
    handles = []
    for batch in (0, 1):                                              # two batches per fill (more than TPG_MAX_FIELDS fields)
        for r, plan in enumerate(plans):
            send = {m.side: torch.full((4,), 100.0 * batch + 10 * r + m.side) for m in plan}
            recv = {m.side: torch.zeros(4) for m in plan}
            handles.append((r, batch, box.endpoint(r).post(plan, send, recv)))
    for r, batch, h in handles:
        box.endpoint(r).wait(h)
        plan, recv = h
        for m in plan:                                                # what the peer sent towards us: its side facing us
            want = 100.0 * batch + 10 * m.peer + (SOUTH if m.side == NORTH else NORTH)
            assert bool((recv[m.side] == want).all()), (r, batch, m)
    assert all(not q for q in box.box.values())


def test_rccl_comm_rejects_a_malformed_unique_id(osg):
    with pytest.raises(ValueError):
        osg.RcclComm.create(b"short", 0, 1)


def test_bench_watchdog_fires_with_one_json_line():
    """bench.py's deadline for the first contact with other ranks: on expiry ONE JSON line (rank, peers, phase) on stderr and
    os._exit(3) -- also when the main thread is blocked in a host call (here: a sleep)"""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, time; sys.path.insert(0, %r); import bench\n"
            "d = bench.Watchdog(0.3, {'rank': 2, 'world': 8, 'peers': {'south': 1, 'north': 3}})\n"
            "d.arm('first seam exchange: enqueue'); d.set_phase('first seam exchange: device')\n"
            "time.sleep(20); print('not reached')\n" % root)
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60)
    assert p.returncode == 3 and "not reached" not in p.stdout
    lines = [l for l in p.stderr.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["event"] == "bench_deadline_expired" and d["rank"] == 2 and d["peers"] == {"south": 1, "north": 3}
    assert d["phase"] == "first seam exchange: device" and d["deadline_s"] == 0.3
    # SOFT mode (the pipelined probe of an N > 1 run): on expiry the callback runs -- it prints what the run already holds -- and the
    # process leaves with status 9 (bench_chain.PROBE_FAILED: the line is out, but a product entry point did not come back) instead of 3,
    # also from under a blocked main thread
    code3 = ("import sys, time; sys.path.insert(0, %r); import bench\n"
             "d = bench.Watchdog(0.3, {'rank': 0})\n"
             "d.soft = lambda dog: print('LINE-IN-HAND after', dog.phase, flush=True)\n"
             "d.arm('pipelined probe (pipelined_1): first exchange'); time.sleep(20); print('not reached')\n" % root)
    p3 = subprocess.run([sys.executable, "-c", code3], capture_output=True, text=True, timeout=60)
    assert p3.returncode == 9 and "not reached" not in p3.stdout and "LINE-IN-HAND after pipelined probe (pipelined_1): first exchange" in p3.stdout
    assert "bench_deadline_expired" not in p3.stderr
    # a disarmed watchdog does nothing
    code2 = ("import sys, time; sys.path.insert(0, %r); import bench\n"
             "d = bench.Watchdog(0.2, {}); d.arm('x'); d.disarm(); time.sleep(0.6); print('alive')\n" % root)
    p2 = subprocess.run([sys.executable, "-c", code2], capture_output=True, text=True, timeout=60)
    assert p2.returncode == 0 and "alive" in p2.stdout


def test_bench_launches_its_own_workers(tmp_path):
    """`python bench.py --gpus N` without a launcher starts N workers itself: one child per rank with RANK /
    LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set, rank 0's ONE contract line relayed to stdout, everything else any rank
    prints sent to stderr, exit status = the children's.  The parent must not have imported torch (nothing of it may touch a GPU)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    worker = tmp_path / "worker.py"
    worker.write_text("import os, sys, json\n"
                      "r, w = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])\n"
                      "assert os.environ['LOCAL_RANK'] == str(r) and os.environ['MASTER_ADDR'] == '127.0.0.1' and int(os.environ['MASTER_PORT']) > 0\n"
                      "assert os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY') is not None\n"
                      "print('chatter from rank %d' % r)\n"
                      "if r == 0: print(json.dumps({'metric': 'm', 'n_gpus': w, 'argv': sys.argv[1:]}))\n"
                      "sys.exit(int(os.environ.get('FAIL_RANK', '-1')) == r and 7 or 0)\n")
    code = ("import sys, types; sys.path.insert(0, %r); import bench\n"
            "rc = bench.launch_workers(types.SimpleNamespace(gpus=3), ['--gpus', '3', '--steps', '2'], script=%r)\n"
            "assert 'torch' not in sys.modules, 'the launching parent imported torch'\n"
            "sys.exit(rc)\n" % (root, str(worker)))
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and json.loads(lines[0]) == {"metric": "m", "n_gpus": 3, "argv": ["--gpus", "3", "--steps", "2"]}
    assert all(f"chatter from rank {r}" in p.stderr for r in range(3))
    # a failing worker fails the job with its status
    p2 = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120, env=dict(os.environ, FAIL_RANK="2"))
    assert p2.returncode == 7
    # and the real thing on a box without a GPU: both workers start and stop at the device-count preflight -- status 7 and ONE readable
    # line from rank 0 (not a usage message, not a HIP traceback per rank)
    import torch
    if not torch.cuda.is_available():
        env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
        p3 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=300, env=env)
        assert p3.returncode == 7 and not p3.stdout.strip()
        diag = [json.loads(l) for l in p3.stderr.splitlines() if l.startswith("{") and "too_few_devices" in l]
        assert len(diag) == 1 and diag[0]["visible"] == 0 and diag[0]["requested"] == 2 and "launch N > 1 with" not in p3.stderr


def test_bench_refuses_a_strong_split_that_does_not_divide():
    """config 4 divides the 1800 rows evenly; the remainder rule for Ny % R != 0 is Oceananigans-internal (unpinned), so the
    strong-scaling bench refuses such an N before touching a device"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="7", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "7"], capture_output=True, text=True, env=env, timeout=300)
    assert p.returncode != 0 and "1800 % N == 0" in p.stderr and not p.stdout.strip()
    env2 = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}      # ... and so does the self-launching parent
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "7"], capture_output=True, text=True, env=env2, timeout=300)
    assert p.returncode != 0 and "1800 % N == 0" in p.stderr and not p.stdout.strip()


def test_convert_to_0_360(osg):
    """src/OrthogonalSphericalShellGrids.jl:24 with Julia's truncated `%`"""
    f = osg.convert_to_0_360
    assert f(0.0) == 0.0 and f(360.0) == 0.0 and f(-90.0) == 270.0 and f(725.5) == 5.5 and f(-725.5) == 354.5 and f(359.9999) == 359.9999


def test_traffic_json_is_keyed_per_kernel_source():
    """profiles/traffic.json: every kernel entry names the files its kernel is compiled from and one hash over them; bench.py reports a
    kernel's PMC traffic only while that hash matches the tree (the file used to be keyed to the zipper header only, so a
    changed cell kernel kept a stale figure).  A stale entry is dropped, never reported."""
    import json
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench_common as bench                                            # load_traffic / sources_sha16 live there (bench.py re-exports them)
    tj = json.load(open(os.path.join(root, "profiles", "traffic.json")))
    assert "kernel_source_sha16" not in tj                                   # the single global key is gone
    ks = tj["kernels"]
    for name in ("k_fill_merged", "k_zipper_cols", "k_periodic_x_vec", "k_cells_tile", "k_tables", "k_halos"):
        e = ks[name]
        assert e["sources"] and all(os.path.exists(os.path.join(root, s)) for s in e["sources"]) and len(e["sources_sha16"]) == 16
        assert e["hbm_bytes_per_launch"] == 2 * e["fetch_bytes_raw"] + e["write_bytes"]      # gfx950: FETCH_SIZE counts 128-B requests as 64 B
    assert any("tpg_grid.hip" in s for s in ks["k_cells_tile"]["sources"]) and any("tpg_zipper_kernels.hpp" in s for s in ks["k_fill_merged"]["sources"])
    live = bench.load_traffic()
    for name, e in ks.items():                                                # reported <=> the hash over its sources still matches
        assert (name in live) == (bench.sources_sha16(e["sources"]) == e["sources_sha16"])
    # a changed source drops exactly the kernels compiled from it
    saved = bench.sources_sha16
    try:
        bench.sources_sha16 = lambda srcs: "0" * 16 if any("tpg_grid.hip" in s for s in srcs) else saved(srcs)
        stale = bench.load_traffic()
        assert "k_cells_tile" not in stale and "k_tables" not in stale and ("k_fill_merged" in stale) == ("k_fill_merged" in live)
    finally:
        bench.sources_sha16 = saved


def test_every_rank_enters_the_plan_agreement_before_its_branch():
    """the agreement collective of a plan build used to sit inside the production branch only -- a rank whose fields were
    not uniformly zipped took the other branch and left its peers waiting in the collective.  It now runs for every group that has a
    communicator, before the branch is chosen.  It is deliberately NOT memoised (a rank that skips it on local knowledge strands the rank
    that brings a different layout: tests/test_distributed_gloo.py has that case) and plans are not cached behind fill_halo_regions (a
    plan reachable from its own fields is a reference cycle holding 32 GB tensors)."""
    import inspect
    import orthogonalsphericalshellgrids.jl_amd.fields as F
    src = inspect.getsource(F.HaloFillPlan.__init__)
    agree, branch = src.index("_agree_across_ranks(arch,"), src.index("if comm is not None and uniform and")
    assert agree < branch and "if comm is not None:" in src[:agree]
    body = inspect.getsource(F._agree_across_ranks)
    assert "_agreed" not in body and "all_gather_object" in body
    assert "_fill_plans" not in inspect.getsource(F.fill_halo_regions)
