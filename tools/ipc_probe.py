#!/usr/bin/env python3
"""GPU box: can two processes on ONE GPU map each other's device memory (HIP IPC, dmabuf mode) and see each other's stores
inside running kernels?  Parent allocates a tensor, hands it to a spawned child through torch.multiprocessing (hipIpcGetMemHandle /
hipIpcOpenMemHandle underneath); the child writes a pattern, the parent reads it back; then both directions with a flag word polled
by a bounded spin on the host side.  Prints one JSON line.  Bring-up probe for the peer-to-peer seam push."""
import json
import os
import sys
import time

import torch
import torch.multiprocessing as mp


def child(q_in, q_out):
    torch.cuda.set_device(0)
    t = q_in.get(timeout=60)                      # a tensor aliasing the parent's allocation
    flag = q_in.get(timeout=60)
    t[1::2] = 7.5                                  # child's stores into the parent's memory
    torch.cuda.synchronize()
    flag.fill_(1)
    torch.cuda.synchronize()
    mine = torch.arange(16, dtype=torch.float64, device="cuda")
    q_out.put(mine)
    # wait (bounded) for the parent to have written into OUR memory
    t0 = time.time()
    while float(mine[0]) != -1.0 and time.time() - t0 < 20:
        time.sleep(0.01)
    q_out.put(float(mine[0]))
    time.sleep(1.0)


def main():
    mp.set_start_method("spawn")
    torch.cuda.set_device(0)
    out = {"ok": False}
    q_in, q_out = mp.Queue(), mp.Queue()
    p = mp.Process(target=child, args=(q_in, q_out))
    p.start()
    t = torch.zeros(1 << 20, dtype=torch.float64, device="cuda")
    flag = torch.zeros(1, dtype=torch.int32, device="cuda")
    q_in.put(t); q_in.put(flag)
    t0 = time.time()
    while int(flag.item()) != 1 and time.time() - t0 < 30:
        time.sleep(0.01)
    out["child_flag_seen_after_s"] = round(time.time() - t0, 3)
    out["child_stores_visible"] = bool((t[1::2] == 7.5).all() and (t[0::2] == 0).all())
    theirs = q_out.get(timeout=60)
    theirs[0] = -1.0
    torch.cuda.synchronize()
    out["parent_store_seen_by_child"] = q_out.get(timeout=60) == -1.0
    p.join(30)
    out["child_exit"] = p.exitcode
    out["ok"] = out["child_stores_visible"] and out["parent_store_seen_by_child"] and p.exitcode == 0
    print(json.dumps(out))
    return 0 if out["ok"] else 1


if __name__ == "__main__":
    sys.exit(main())
