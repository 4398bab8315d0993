"""Rehearsal of the gradient all-reduce on a 1-GPU box: world_size ranks share cuda:0 and reduce through gloo
(RCCL refuses two ranks on one device).  Times device-tensor and host-tensor all-reduces of growing size, each
under the process group's timeout, so a stuck collective ends the rank with a non-zero exit instead of hanging.

    python tools/rehearse_allreduce.py [world_size] [max_mb]
"""
import datetime
import os
import socket
import sys
import time

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def worker(rank, world, port, max_mb):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))
    for mb in (1, 16, 64, max_mb):
        n = mb * 1024 * 1024 // 4
        for where in ("host", "device"):
            t = torch.full((n,), float(rank + 1), dtype=torch.float32, device="cuda" if where == "device" else "cpu")
            torch.cuda.synchronize()
            dist.barrier()
            t0 = time.perf_counter()
            dist.all_reduce(t)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            ok = bool((t == world * (world + 1) / 2).all())
            if rank == 0:
                print(f"all_reduce {mb:4d} MB {where:6s}: {dt * 1e3:8.1f} ms  {'ok' if ok else 'WRONG'}", flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    max_mb = int(sys.argv[2]) if len(sys.argv) > 2 else 236
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(worker, args=(world, port, max_mb), nprocs=world, join=True)
