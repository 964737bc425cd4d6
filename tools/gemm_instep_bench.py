#!/usr/bin/env python3
"""GEMM tiles at the LIVE sizes of the batch-64 step (device-side token counts: ~70 k content / ~20 k title / ~15 k candidate rows
inside buffers of 409 600 / 102 400 / 40 960 rows), interleaved rounds in one process.  Usage: gemm_instep_bench.py [nt|tn]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nnr_amd import ops

d = torch.device('cuda')
ROUNDS = 5


def time_many(fns, iters):
    res = {k: [] for k in fns}
    for f in fns.values():
        f()
    torch.cuda.synchronize()
    for _ in range(ROUNDS):
        for k, f in fns.items():
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(iters):
                f()
            e.record()
            torch.cuda.synchronize()
            res[k].append(s.elapsed_time(e) / iters)
    return {k: sorted(v)[len(v) // 2] for k, v in res.items()}


def nt():
    tiles = [int(t) for t in os.environ.get('TILES', '0,5,15,16,9,17').split(',')]
    for name, live, cap, N, K, scatter in (('xw content', 70000, 409600, 1664, 300, False), ('xw title', 20000, 102400, 1664, 300, False),
                                           ('xw cand', 15000, 40960, 1664, 300, False), ('gate content', 70000, 409600, 400, 400, False),
                                           ('gate title', 20000, 102400, 400, 400, False), ('att content', 70000, 409600, 200, 400, False),
                                           ('dHt content', 70000, 409600, 400, 200, False), ('dX content', 70000, 409600, 300, 1664, True),
                                           ('dX title', 20000, 102400, 300, 1664, True), ('dX cand', 15000, 40960, 300, 1664, True)):
        a = torch.randn(cap, K, device=d); b = torch.randn(N, K, device=d) * 0.05
        dyn = torch.tensor([live], device=d, dtype=torch.int32)
        if scatter:
            c = torch.zeros(60000, N, device=d)
            idx = (torch.rand(cap, device=d) ** 3 * 59999).int()
            kw = dict(c_idx=idx, atomic=True, drop=(4, 0.2, 7, N))
        else:
            c = torch.empty(cap, N, device=d)
            kw = {}
        fns = {('t%d' % t): (lambda t=t: ops.gemm(a, b, c, M=cap, N=N, K=K, lda=K, ldb=K, ldc=N, dyn=dyn, dyn_dim=1, tile=t, **kw)) for t in tiles}
        r = time_many(fns, 8)
        fl = 2.0 * live * N * K
        print('%-14s live %6d ' % (name, live) + ' '.join('%s %6.1f us %5.1f TF' % (k, v * 1e3, fl / v / 1e9) for k, v in r.items()), flush=True)


def tn():
    tiles = [int(t) for t in os.environ.get('TN_TILES', '2,26,27').split(',')]
    for name, M, N, live, cap, gather in (('dW_ih content', 1664, 300, 70000, 409600, False), ('dW_ih title', 1664, 300, 20000, 102400, False),
                                          ('dW_hh content', 832, 200, 70000, 409600, True), ('dW_hh title', 832, 200, 20000, 102400, True),
                                          ('dW_H content', 400, 400, 70000, 409600, False), ('dW1 content', 200, 400, 70000, 409600, False),
                                          ('dW_H title', 400, 400, 20000, 102400, False), ('sue dW', 900, 900, 4352, 4352, False)):
        a = torch.randn(cap, M, device=d); b = torch.randn(cap, N, device=d) * 0.05; c = torch.zeros(M, N, device=d)
        dyn = torch.tensor([live], device=d, dtype=torch.int32)
        bidx = (torch.arange(cap, device=d, dtype=torch.int32) - 3200).clamp_min(-1) if gather else None
        fns = {}
        for t in tiles:
            if gather and t in (26,):
                continue
            bm = {2: 64, 26: 128, 27: 128, 30: 128}[t]
            bn = {27: 208, 30: 160}.get(t, 80)
            for target in (256, 512, 1024, 2048):
                sk = ops.split_for(M, N, cap, tile_m=bm, tile_n=bn, target_blocks=target)
                fns['t%d/%d' % (t, target)] = (lambda t=t, sk=sk: ops.gemm(a, b, c, M=M, N=N, K=cap, lda=M, ldb=N, ldc=N, trans_a=True, trans_b=True,
                                                                             split_k=sk, atomic=True, tile=t, b_idx=bidx, dyn=dyn, dyn_dim=2))
        r = time_many(fns, 8)
        fl = 2.0 * M * N * live
        best = {}
        for k, v in r.items():
            t = k.split('/')[0]
            if t not in best or v < best[t][0]:
                best[t] = (v, k.split('/')[1])
        print('%-14s live %6d ' % (name, live) + ' '.join('%s %6.1f us %5.1f TF (@%s)' % (t, v[0] * 1e3, fl / v[0] / 1e9, v[1]) for t, v in best.items()), flush=True)


if __name__ == '__main__':
    (tn if len(sys.argv) > 1 and sys.argv[1] == 'tn' else nt)()
