"""Data parallelism of the hot path (reference: trainer.py:209-300, DistributedDataParallel over NCCL).

One process per GPU, `torch.distributed` with backend "nccl" (= RCCL over xGMI on ROCm) or "gloo" (CPU tests).
Impressions are independent, so the only exchange is ONE all-reduce(sum) of the flat fp32 gradient buffer per step
(4*P bytes); the 1/world_size average is folded into the fused clip+Adam kernel.  Parameters are broadcast from
rank 0 once (DDP's constructor does the same).  Every rank keeps a full optimizer replica, as the reference does."""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Join the process group described by RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (torchrun contract)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', str(rank)))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29512')
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        if backend == 'nccl':
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, init_method='env://', world_size=world, rank=rank)
    return rank, local, world


def world_size():
    return dist.get_world_size() if dist.is_initialized() else 1


def broadcast_parameters(flat_params):
    if world_size() > 1:
        dist.broadcast(flat_params, src=0)


def allreduce_gradients(flat_grads):
    """Sum the flat gradient over ranks; returns the scale (1/world) the optimizer must apply."""
    w = world_size()
    if w > 1:
        dist.all_reduce(flat_grads, op=dist.ReduceOp.SUM)
    return 1.0 / w


def shard_batch(batch, rank, world):
    """Per-rank slice of a global batch: rank r takes samples r::world (DistributedSampler order, trainer.py:256-258)."""
    if world == 1:
        return batch
    return [t[rank::world].contiguous() for t in batch]


def barrier():
    if world_size() > 1:
        dist.barrier()
