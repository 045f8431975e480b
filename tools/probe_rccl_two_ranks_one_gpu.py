"""Can RCCL form a 2-rank communicator with both ranks on the one GPU of a test box?  (NCCL refuses duplicate devices;
this records what RCCL on this image does, so that nobody has to wonder again.)"""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def worker(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    try:
        dist.init_process_group("nccl", rank=rank, world_size=world)
        t = torch.full((1024,), float(rank + 1), device="cuda")
        dist.all_reduce(t)
        torch.cuda.synchronize()
        print(f"rank {rank}: all_reduce ok -> {float(t[0])}", flush=True)
        dist.destroy_process_group()
    except Exception as e:          # noqa: BLE001
        print(f"rank {rank}: FAILED {type(e).__name__}: {str(e)[:400]}", flush=True)
        sys.exit(3)


if __name__ == "__main__":
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    try:
        mp.spawn(worker, args=(2, port), nprocs=2, join=True)
        print("RESULT: two RCCL ranks on one GPU work")
    except Exception as e:          # noqa: BLE001
        print("RESULT: two RCCL ranks on one GPU do NOT work:", str(e)[:300])
