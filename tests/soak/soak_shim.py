"""GPU box: randomised soak of the C ABI's seam exchange BETWEEN REAL PROCESSES on one GPU (test double of librccl, tools/nccl_shim, behind the
test library): W processes (default 4) draw the same random cases -- geometry, element type, 1-16 fields, exchange form (packed, pack-free,
pipelined in stages of 1..nf+1 on one stream or two, the one-call distributed fill monolithic / pipelined), which ranks take part as a chain --
and every rank checks its halo rows against what its neighbours hold (their data is reproducible from the seed) after the neighbours' own
local fill (oracle).  usage: python tests/soak/soak_shim.py [trials] [seed] [world]"""
import ctypes as C, os, socket, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import torch.distributed as dist
import torch.multiprocessing as mp


def worker(rank, world, port, trials, seed, out):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), TPG_RCCL_LIBRARY=os.path.join(ROOT, "tools", "nccl_shim", "libnccl_shim.so"),
                          TPG_SHIM_DEADLINE_S="30")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import orthogonalsphericalshellgrids.jl_amd as osg
        from orthogonalsphericalshellgrids.jl_amd import _lib
        from tools import testlib
        from oracle import oracle
        torch.cuda.set_device(0)
        dev = torch.device("cuda", 0)
        _lib._lib = testlib.lib()
        lib = _lib.lib()
        comm = osg.RcclComm.from_torch()
        stream = _lib.current_stream_ptr(dev)
        cs = torch.cuda.Stream(dev)
        rng = np.random.default_rng(seed)                               # the SAME stream of cases on every rank
        bad = 0
        parent = os.getppid()
        for t in range(trials):
            if os.getppid() != parent:                                  # the launcher is gone (killed by a time limit): do not linger on the GPU
                os._exit(2)
            Nx = 2 * int(rng.integers(2, 40)); Nz = int(rng.integers(1, 4))
            Hx = int(rng.integers(0, min(Nx, 4) + 1)); Hy = int(rng.integers(1, 5)); Hz = int(rng.integers(0, 3))
            Ny = int(rng.integers(2 * Hy + 1, 2 * Hy + 12))
            nf = int(rng.integers(1, 17))
            f64 = bool(rng.integers(0, 2))
            dt, tdt, ft = (np.float64, torch.float64, 1) if f64 else (np.float32, torch.float32, 0)
            form = int(rng.integers(0, 5))                              # 0 packed, 1 pack-free, 2 pipelined, 3 distributed fill, 4 distributed fill pipelined
            fps = int(rng.integers(0, nf + 2))
            two = bool(rng.integers(0, 2))
            chain = int(rng.integers(2, world + 1))                     # ranks 0 .. chain-1 form the latitude-band chain of this trial
            specs = [(int(rng.integers(0, 2)), int(rng.integers(0, 2)), int(rng.choice([1, -1]))) for _ in range(nf)]
            data_seed = int(rng.integers(0, 1 << 30))
            if form == 1 and nf * (Nz + 2 * Hz) > 24:
                form = 0                                                # pack-free sends one message per (field, level): keep the double's staging short
            if rank >= chain:
                continue
            shape = (Nz + 2 * Hz, Ny + 2 * Hy, Nx + 2 * Hx)
            mk = lambda r: [np.random.default_rng(data_seed + 97 * r + f).uniform(-1, 1, shape).astype(dt) for f in range(nf)]
            mine = mk(rank)
            devs = [torch.from_numpy(a).to(dev) for a in mine]
            ptrs = _lib.ptr_table(devs)
            nbuf = lib.tpg_y_halo_buffer_elems(nf, Nx, Nz, Hx, Hy, Hz)
            bufs = [torch.empty(max(int(nbuf), 1), dtype=tdt, device=dev) for _ in range(4)]
            bp = [b.data_ptr() for b in bufs]
            xl = (C.c_int8 * nf)(*[q[0] for q in specs]); yl = (C.c_int8 * nf)(*[q[1] for q in specs]); sg = (C.c_int32 * nf)(*[q[2] for q in specs])
            csp = C.c_void_p(cs.cuda_stream) if two else None
            g = (Nx, Ny, Nz, Hx, Hy, Hz)
            if form == 0:
                rc = lib.tpg_halo_exchange_y(comm.handle, rank, chain, ptrs, nf, *bp, *g, ft, stream)
            elif form == 1:
                rc = lib.tpg_halo_exchange_y(comm.handle, rank, chain, ptrs, nf, None, None, None, None, *g, ft, stream)
            elif form == 2:
                rc = lib.tpg_halo_exchange_y_pipelined(comm.handle, rank, chain, ptrs, nf, *bp, *g, ft, stream, csp, fps)
            elif form == 3:
                rc = lib.tpg_fill_halo_regions_distributed(comm.handle, rank, chain, ptrs, nf, xl, yl, sg, *bp, *g, ft, stream)
            else:
                rc = lib.tpg_fill_halo_regions_distributed_pipelined(comm.handle, rank, chain, ptrs, nf, xl, yl, sg, *bp, *g, ft, stream, csp, fps)
            torch.cuda.synchronize()
            if rc != 0:
                bad += 1; print(f"rank {rank} trial {t}: rc {rc} {lib.tpg_last_error().decode()}", flush=True); break
            local = form >= 3
            def filled(r):
                fs = mk(r) if r != rank else [a.copy() for a in mine]
                if local:
                    for f, a in enumerate(fs):
                        if r == chain - 1:
                            oracle.zipper_fill(a, specs[f][0], specs[f][1], specs[f][2], (Nx, Ny, Nz), (Hx, Hy, Hz))
                        oracle.periodic_x_fill(a, (Nx, Ny, Nz), (Hx, Hy, Hz))
                return fs
            want = filled(rank)
            if rank > 0:
                south = filled(rank - 1)
                for f in range(nf): want[f][:, :Hy] = south[f][:, Ny:Ny + Hy]
            if rank < chain - 1:
                north = filled(rank + 1)
                for f in range(nf): want[f][:, Ny + Hy:] = north[f][:, Hy:2 * Hy]
            if os.environ.get("TPG_SOAK_TEETH") == "1" and t == 3 and rank == 1:
                devs[0][0, 0, 0] += 1.0                                # self-check of the checker: exactly this must be reported (exit status 1)
            for f in range(nf):
                if not np.array_equal(devs[f].cpu().numpy(), want[f]):
                    bad += 1; print(f"MISMATCH rank {rank} trial {t} form {form} fps {fps} two {two} chain {chain} geom {g} nf {nf} f64 {f64} field {f}", flush=True); break
            if rank == 0 and t % 100 == 99:
                print(f"{t + 1} trials, rank 0: {bad} bad", flush=True)
        out.put((rank, bad))
        dist.barrier()
        comm.destroy()
        dist.destroy_process_group()
    except Exception as e:                                             # noqa: BLE001
        import traceback
        print(f"rank {rank}: {type(e).__name__}: {e}\n{traceback.format_exc()[-1200:]}", flush=True)
        out.put((rank, -1))
        os._exit(1)


if __name__ == "__main__":
    trials = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    world = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    assert 2 <= world <= 5, "at most 5 ranks + this launcher may hold the card (process guard: 6)"
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
    ctx = mp.get_context("spawn")
    out = ctx.SimpleQueue()
    procs = [ctx.Process(target=worker, args=(r, world, port, trials, seed, out)) for r in range(world)]
    for p in procs: p.start()
    for p in procs: p.join(3000)
    res = {}
    while not out.empty():
        r, b = out.get(); res[r] = b
    print("done:", trials, "trials x", world, "ranks, bad per rank:", res)
    sys.exit(0 if len(res) == world and all(v == 0 for v in res.values()) else 1)
