#!/usr/bin/env python3
"""A/B of the GEMM tiles on the shapes of the CNE+SUE step (run on the GPU box): interleaved rounds in one process, median
and best per (shape, tile).  Tiles 4/5/2 = register-staged gemm_kernel, 8..14 = LDS-DMA pipelined NT kernel."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nnr_amd import ops

d = torch.device('cuda')
ROUNDS = int(os.environ.get('ROUNDS', '5'))


def time_many(fns, iters):
    res = {k: [] for k in fns}
    for k, f in fns.items():
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
    return {k: (sorted(v)[len(v) // 2], min(v)) for k, v in res.items()}


def tn_main():
    shapes = [('dW_ih 1664x300xT', 1664, 300, 131072, 4), ('dW_hh 832x200xT', 832, 200, 131072, 6), ('dW_H 400x400xT', 400, 400, 131072, 8),
              ('dW1 200x400xT', 200, 400, 131072, 10), ('title dW_ih 1664x300x36000', 1664, 300, 36000, 10), ('sue 900x900x4352', 900, 900, 4352, 30)]
    tiles = [int(t) for t in os.environ.get('TN_TILES', '2,4,20,21,22,23,24,25').split(',')]
    for name, M, N, K, iters in shapes:
        a = torch.randn(K, M + 0, device=d); b = torch.randn(K, N, device=d) * 0.05; c = torch.zeros(M, N, device=d)
        gather = 'hh' in name
        bidx = (torch.arange(K, device=d, dtype=torch.int32) - 3200).clamp_min(-1) if gather else None      # previous time step's row; first rows: none
        dyn = torch.tensor([K - 1000], device=d, dtype=torch.int32)
        fl = 2.0 * M * N * K
        line = '%-28s' % name
        for tm, tn in ((64, 80),):
            pass
        fns = {}
        for t in tiles:
            bm = {2: 64, 4: 128, 20: 128, 21: 128, 22: 64, 23: 256, 24: 128, 25: 128, 26: 128, 27: 128, 28: 128, 29: 256, 30: 128, 32: 64, 37: 256, 38: 128, 39: 64}[t]
            bn = {24: 208, 25: 128, 27: 208, 30: 160, 32: 208, 37: 160, 38: 160, 39: 160}.get(t, 80)
            if gather and t in (26, 28, 29):
                continue
            for target in (512, 1024, 2048):
                sk = ops.split_for(M, N, K, tile_m=bm, tile_n=bn, target_blocks=target)
                fns['t%d/%d' % (t, target)] = (lambda t=t, sk=sk: ops.gemm(a, b, c, M=M, N=N, K=K, lda=M, ldb=N, ldc=N, trans_a=True, trans_b=True, split_k=sk, atomic=True, tile=t, b_idx=bidx, dyn=dyn, dyn_dim=2))
        r = time_many(fns, iters)
        best = {}
        for k, v in r.items():
            t = k.split('/')[0]
            tf = fl / v[0] / 1e9
            if t not in best or tf > best[t][0]:
                best[t] = (tf, k.split('/')[1])
        print(line + ' '.join('%s %6.1f(@%s)' % (t, v[0], v[1]) for t, v in best.items()), flush=True)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == 'tn':
        return tn_main()
    shapes = [('xw 131072x1664x300', 131072, 1664, 300, 6), ('gate 131072x400x400', 131072, 400, 400, 10), ('att 131072x200x400', 131072, 200, 400, 10),
              ('dX 131072x300x1664', 131072, 300, 1664, 6), ('dHt 131072x400x200', 131072, 400, 200, 10),
              ('sue 4352x900x900', 4352, 900, 900, 30), ('sue 6080x900x900', 6080, 900, 900, 30), ('title 36000x1664x300', 36000, 1664, 300, 10),
              ('cand 14000x400x400', 14000, 400, 400, 30), ('big 8192x8000x4096', 8192, 8000, 4096, 2)]
    tiles = [int(t) for t in os.environ.get('TILES', '5,2,8,13,15,16,17,18,19').split(',')]
    only = os.environ.get('SHAPES', '')
    shapes = [x for x in shapes if only in x[0]]
    out = {}
    for name, M, N, K, iters in shapes:
        a = torch.randn(M, K, device=d); b = torch.randn(N, K, device=d) * 0.05; c = torch.empty(M, N, device=d)
        fns = {}
        for t in tiles:
            fns['t%d' % t] = (lambda t=t: ops.gemm(a, b, c, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, tile=t))
        fns['rocblas'] = lambda: torch.mm(a, b.t(), out=c)
        r = time_many(fns, iters)
        fl = 2.0 * M * N * K
        out[name] = {k: round(fl / v[0] / 1e9, 1) for k, v in r.items()}
        print('%-24s' % name + ' '.join('%s %6.1f' % (k, fl / v[0] / 1e9) for k, v in r.items()), flush=True)
    json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'gpurun_out', 'gemm_pipe_bench.json'), 'w'), indent=1)


if __name__ == '__main__':
    main()
