"""GPU box: the torch.distributed (nccl = RCCL) calls bench.py makes for N > 1, on a world of one rank
(RCCL refuses two ranks on one device, so the point-to-point leg cannot be exercised on a 1-GPU box)."""
import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
torch.cuda.set_device(0); dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
t = torch.tensor([1.25], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX); dist.broadcast(t, src=0); dist.barrier(); torch.cuda.synchronize()
print("nccl world=1: all_reduce / broadcast / barrier ok", float(t.item()), "batch_isend_irecv importable:", callable(dist.batch_isend_irecv))
dist.destroy_process_group()
