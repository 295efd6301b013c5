"""GPU box: randomised soak of the PIPELINED RCCL seam exchange (tpg_halo_exchange_y_pipelined_peers and the whole-fill entry point
tpg_fill_halo_regions_distributed_pipelined_peers) on a one-rank communicator whose peers are the rank itself: random geometry,
element type, number of fields (1..16), stage size (0..nfields+1), one stream or two, two seams / south only / north only, every call
issued twice on the same buffers.  Expected halos follow from the loop-back pairing (sent north -> received from the south, ...);
with the zipper (north band) the expected local part is the product's own local fill, which is bit-exact against the oracle elsewhere.
Also, in the same trials, the emulated-rank form: R ranks in this process, HaloFillPlan(fields_per_stage = k) over the loop-back
mailbox against the oracle's serial fill.  usage: soak_pipelined.py [trials] [seed]   (run as a child process under a timeout)"""
import ctypes as C, os, socket, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import torch.distributed as dist
import orthogonalsphericalshellgrids.jl_amd as osg
from orthogonalsphericalshellgrids.jl_amd import _lib
from oracle import oracle

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 4)
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
lib = _lib.lib()
with socket.socket() as sk:
    sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
comm = osg.RcclComm.from_torch()
comm_stream = torch.cuda.Stream(dev)
side = torch.cuda.Stream(dev)
SENT = 12345.0
bad = 0
for t in range(trials):
    Hx, Hy, Hz = int(rng.integers(1, 6)), int(rng.integers(1, 6)), int(rng.integers(0, 3))
    Ny = int(rng.integers(2 * Hy + 2, 2 * Hy + 14)); Nx = 2 * int(rng.integers(max(1, Hx) + 1, 70)); Nz = int(rng.integers(1, 5))
    dtype, tdt, ft = ((np.float64, torch.float64, 1), (np.float32, torch.float32, 0))[t % 2]
    nf = int(rng.integers(1, 17)); fps = int(rng.integers(0, nf + 2))
    south, north = ((0, 0), (0, -1), (-1, 0))[int(rng.integers(0, 3))]
    zipper = 1 if (north < 0 and rng.integers(0, 2)) else 0                    # the zipper band has no north seam
    two = bool(rng.integers(0, 2))
    shape = (Nz + 2 * Hz, Ny + 2 * Hy, Nx + 2 * Hx)
    specs = [(int(rng.integers(0, 2)), int(rng.integers(0, 2))) for _ in range(nf)]
    xl = (C.c_int8 * nf)(*[s[0] for s in specs]); yl = (C.c_int8 * nf)(*[s[1] for s in specs])
    sg = (C.c_int32 * nf)(*[-1 if s[0] != s[1] else 1 for s in specs])
    hosts = [rng.uniform(-1, 1, shape).astype(dtype) for _ in range(nf)]
    devs = [torch.from_numpy(h).to(dev) for h in hosts]
    refs = [d.clone() for d in devs]
    _lib.check(lib.tpg_fill_halo_regions(_lib.ptr_table(refs), nf, xl, yl, sg, Nx, Ny, Nz, Hx, Hy, Hz, zipper, ft, None))
    torch.cuda.synchronize()
    nbuf = lib.tpg_y_halo_buffer_elems(nf, Nx, Nz, Hx, Hy, Hz)
    bufs = [torch.full((nbuf,), float("nan"), dtype=tdt, device=dev) for _ in range(4)]
    csp = C.c_void_p(comm_stream.cuda_stream) if two else None
    with torch.cuda.stream(side):
        rc = lib.tpg_fill_halo_regions_distributed_pipelined_peers(comm.handle, south, north, zipper, _lib.ptr_table(devs), nf, xl, yl, sg,
                                                                   *[b.data_ptr() for b in bufs], Nx, Ny, Nz, Hx, Hy, Hz, ft,
                                                                   C.c_void_p(side.cuda_stream), csp, fps)
        rc2 = lib.tpg_halo_exchange_y_pipelined_peers(comm.handle, south, north, _lib.ptr_table(devs), nf, *[b.data_ptr() for b in bufs],
                                                      Nx, Ny, Nz, Hx, Hy, Hz, ft, C.c_void_p(side.cuda_stream), csp, fps)      # again: idempotent on these rows
    torch.cuda.synchronize()
    ok = rc == 0 and rc2 == 0
    for r, d in zip(refs, devs):
        want = r.clone()
        if south >= 0 and north >= 0:
            want[:, :Hy] = r[:, Ny:Ny + Hy]; want[:, Ny + Hy:] = r[:, Hy:2 * Hy]
        elif south >= 0:
            want[:, :Hy] = r[:, Hy:2 * Hy]
        else:
            want[:, Ny + Hy:] = r[:, Ny:Ny + Hy]
        ok = ok and bool(torch.equal(d, want))
    if not ok:
        bad += 1
        print("MISMATCH", t, (Nx, Ny, Nz), (Hx, Hy, Hz), "nf", nf, "fps", fps, "peers", (south, north), "zipper", zipper, "two", two, dtype.__name__,
              "rc", rc, rc2, lib.tpg_last_error().decode(), flush=True)
    # ---- emulated ranks: HaloFillPlan(fields_per_stage) over the mailbox against the oracle's serial fill (every 4th trial) ----
    if t % 4 == 0:
        R = int(rng.integers(2, 5)); ny = int(rng.integers(Hy + 1, Hy + 8)); NyG = ny * R; nfe = min(nf, 6)
        size, halo = (Nx, NyG, Nz), (Hx, Hy, Hz)
        globs = []
        for _ in range(nfe):
            g = rng.uniform(-1, 1, (Nz + 2 * Hz, NyG + 2 * Hy, Nx + 2 * Hx)).astype(dtype)
            g[:, :Hy] = SENT; g[:, Hy + NyG:] = SENT
            globs.append(g)
        ranks = []
        for r in range(R):
            grid = osg.TripolarGrid(osg.Distributed(osg.GPU(0), osg.Partition(y=R), local_rank=r), tdt, size=size, halo=halo)
            jstart, jend = grid.jrange
            fs = []
            for (fx, fy), g in zip(specs[:nfe], globs):
                f = osg.Field((osg.Face if fx else osg.Center, osg.Face if fy else osg.Center, osg.Center), grid)
                slab = g[:, jstart - 1:jend + 2 * Hy].copy()
                slab[:, :Hy] = SENT; slab[:, Hy + (jend - jstart + 1):] = SENT
                f.data.copy_(torch.from_numpy(slab)); fs.append(f)
            ranks.append((grid, fs))
        mailbox = osg.LoopbackMailbox()
        plans = [osg.halo_fill_plan(fs, exchange=mailbox.endpoint(r), fields_per_stage=min(fps, nfe)) for r, (grid, fs) in enumerate(ranks)]
        for plan in plans: plan.begin()
        for plan in plans: plan.finish()
        torch.cuda.synchronize()
        for (fx, fy), g in zip(specs[:nfe], globs):
            oracle.fill_halo_regions(g, fx, fy, -1 if fx != fy else 1, size, halo)
        for r, (grid, fs) in enumerate(ranks):
            jstart, jend = grid.jrange
            for f, g in zip(fs, globs):
                if not np.array_equal(f.data.cpu().numpy(), g[:, jstart - 1:jend + 2 * Hy]):
                    bad += 1; print("MISMATCH (emulated ranks)", t, size, halo, "R", R, "rank", r, f.loc, "stage", min(fps, nfe), flush=True); break
    if t % 100 == 99: print(f"{t + 1} trials, {bad} bad", flush=True)
print("done:", trials, "trials,", bad, "bad")
comm.destroy(); dist.destroy_process_group()
sys.exit(1 if bad else 0)
