"""Multi-GPU glue.  Frames are independent, so ranks exchange no data: the process group only
carries the start/stop barrier and the max-over-ranks wall time of a run (gloo; one process per GPU)."""
import os

import torch
import torch.distributed as dist


def init_from_env():
    """-> (rank, local_rank, world).  Initialises a gloo group when launched by torch.distributed.run."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    return rank, local_rank, world


def barrier():
    if dist.is_initialized():
        dist.barrier()


def max_over_ranks(value):
    if not dist.is_initialized():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t[0])


def sum_over_ranks(value):
    if not dist.is_initialized():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t[0])


def shutdown():
    if dist.is_initialized():
        dist.destroy_process_group()
