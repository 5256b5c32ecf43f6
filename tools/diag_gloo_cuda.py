"""Does torch.distributed's gloo backend move CUDA (HIP) tensors on this image?  (2 ranks sharing cuda:0)"""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def worker(rank, world, port):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    t = torch.full((4,), float(rank + 1), dtype=torch.float64, device="cuda:0")
    try:
        dist.all_reduce(t)
        print(rank, "all_reduce ok", t.tolist(), flush=True)
    except Exception as e:  # noqa: BLE001
        print(rank, "all_reduce FAILED", repr(e)[:200], flush=True)
    a = torch.full((8,), float(rank), dtype=torch.float64, device="cuda:0")
    b = torch.empty(8, dtype=torch.float64, device="cuda:0")
    try:
        ops = [dist.P2POp(dist.irecv, b, 1 - rank), dist.P2POp(dist.isend, a, 1 - rank)]
        for w in dist.batch_isend_irecv(ops):
            w.wait()
        print(rank, "p2p ok", b.tolist()[:2], flush=True)
    except Exception as e:  # noqa: BLE001
        print(rank, "p2p FAILED", repr(e)[:200], flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    mp.spawn(worker, args=(2, 29611), nprocs=2)
