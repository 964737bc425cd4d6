#!/usr/bin/env python3
"""Single-rank check of the torch.distributed 'nccl' (= RCCL) plumbing bench.py uses at N > 1: init, broadcast, all-reduce (SUM, MAX), barrier."""
import os, torch, torch.distributed as dist
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29544')
torch.cuda.set_device(0)
dist.init_process_group('nccl', init_method='env://', world_size=1, rank=0)
x = torch.arange(1 << 20, device='cuda', dtype=torch.float32)
dist.broadcast(x, src=0); dist.all_reduce(x); dist.barrier()
t = torch.tensor([1.5], device='cuda', dtype=torch.float64); dist.all_reduce(t, op=dist.ReduceOp.MAX)
torch.cuda.synchronize()
print('nccl single-rank ok', float(x[-1]), float(t))
dist.destroy_process_group()
