#!/usr/bin/env python3
"""Micro-benchmark of the f32-MFMA GEMM variants the CNE step uses (run on the GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nnr_amd import ops

d = torch.device('cuda')


def timeit(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def tiles():
    M, E, NP2 = 131072, 300, 1664
    x = torch.randn(M, E, device=d); w = torch.randn(NP2, E, device=d) * 0.05; out = torch.empty(M, NP2, device=d)
    dg = torch.randn(M, NP2, device=d); dw = torch.zeros(NP2, E, device=d)
    h = torch.randn(M, 400, device=d); wh = torch.randn(400, 400, device=d); oh = torch.empty(M, 400, device=d)
    for tile in (4, 5):
        ms = timeit(lambda: ops.gemm(x, w, out, M=M, N=NP2, K=E, lda=E, ldb=E, ldc=NP2, tile=tile))
        print('tile %d NT 131072x1664x300  %7.3f ms %6.1f TF' % (tile, ms, 2.0 * M * NP2 * E / ms / 1e9))
        ms = timeit(lambda: ops.gemm(h, wh, oh, M=M, N=400, K=400, lda=400, ldb=400, ldc=400, tile=tile))
        print('tile %d NT 131072x400x400   %7.3f ms %6.1f TF' % (tile, ms, 2.0 * M * 400 * 400 / ms / 1e9))
        m2 = 13000
        ms = timeit(lambda: ops.gemm(h, wh, oh, M=m2, N=400, K=400, lda=400, ldb=400, ldc=400, tile=tile))
        print('tile %d NT 13000x400x400    %7.3f ms %6.1f TF' % (tile, ms, 2.0 * m2 * 400 * 400 / ms / 1e9))
        ms = timeit(lambda: ops.gemm(dg, w, x, M=M, N=E, K=NP2, lda=NP2, ldb=E, ldc=E, trans_b=True, tile=tile))
        print('tile %d NN 131072x300x1664  %7.3f ms %6.1f TF' % (tile, ms, 2.0 * M * NP2 * E / ms / 1e9))
        for sk in (14, 28, 56):
            ms = timeit(lambda: ops.gemm(dg, x, dw, M=NP2, N=E, K=M, lda=NP2, ldb=E, ldc=E, trans_a=True, trans_b=True, split_k=sk, atomic=True, tile=tile))
            print('tile %d TN 1664x300x131072 split %d %7.3f ms %6.1f TF' % (tile, sk, ms, 2.0 * M * NP2 * E / ms / 1e9))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == 'tiles':
        return tiles()
    M, E, NP2, V = 131072, 300, 1664, 60000
    x = torch.randn(M, E, device=d)
    emb = torch.randn(V, E, device=d)
    idx = torch.randint(0, V, (M,), device=d, dtype=torch.int32)
    w = torch.randn(NP2, E, device=d) * 0.05
    b = torch.randn(NP2, device=d)
    out = torch.empty(M, NP2, device=d)
    dyn = torch.tensor([M], device=d, dtype=torch.int32)
    fl = 2.0 * M * NP2 * E
    cases = {
        'NT plain 131072x1664x300': lambda: ops.gemm(x, w, out, M=M, N=NP2, K=E, lda=E, ldb=E, ldc=NP2),
        'NT +bias +dyn': lambda: ops.gemm(x, w, out, M=M, N=NP2, K=E, lda=E, ldb=E, ldc=NP2, bias=b, dyn=dyn, dyn_dim=1),
        'NT +gather': lambda: ops.gemm(emb, w, out, M=M, N=NP2, K=E, lda=E, ldb=E, ldc=NP2, bias=b, dyn=dyn, dyn_dim=1, a_idx=idx),
        'NT +gather +dropout': lambda: ops.gemm(emb, w, out, M=M, N=NP2, K=E, lda=E, ldb=E, ldc=NP2, bias=b, dyn=dyn, dyn_dim=1, a_idx=idx,
                                                 drop=(1, 0.2, 7, E)),
    }
    for k, f in cases.items():
        ms = timeit(f)
        print('%-34s %8.3f ms  %6.1f TF' % (k, ms, fl / ms / 1e9))
    # square-ish reference shapes
    for (m, n, k) in ((8192, 8000, 4096), (131072, 400, 400), (131072, 1600, 304)):
        a_ = torch.randn(m, k, device=d); b_ = torch.randn(n, k, device=d); c_ = torch.empty(m, n, device=d)
        ms = timeit(lambda: ops.gemm(a_, b_, c_, M=m, N=n, K=k, lda=k, ldb=k, ldc=n))
        print('NT plain %dx%dx%d  %8.3f ms  %6.1f TF' % (m, n, k, ms, 2.0 * m * n * k / ms / 1e9))
        ms = timeit(lambda: torch.mm(a_, b_.t(), out=c_))
        print('   torch.mm (rocBLAS/hipBLASLt) %8.3f ms  %6.1f TF' % (ms, 2.0 * m * n * k / ms / 1e9))
    # NN (dX) with scatter atomics
    dg = torch.randn(M, NP2, device=d)
    dtab = torch.zeros(V, E, device=d)
    fl = 2.0 * M * E * NP2
    ms = timeit(lambda: ops.gemm(dg, w, x, M=M, N=E, K=NP2, lda=NP2, ldb=E, ldc=E, trans_b=True))
    print('%-34s %8.3f ms  %6.1f TF' % ('NN plain 131072x300x1664', ms, fl / ms / 1e9))
    ms = timeit(lambda: ops.gemm(dg, w, dtab, M=M, N=E, K=NP2, lda=NP2, ldb=E, ldc=E, trans_b=True, c_idx=idx, atomic=True))
    print('%-34s %8.3f ms  %6.1f TF' % ('NN +scatter atomics (uniform ids)', ms, fl / ms / 1e9))
    ms = timeit(lambda: ops.gemm(dg, w, dtab, M=M, N=E, K=NP2, lda=NP2, ldb=E, ldc=E, trans_b=True, c_idx=idx, atomic=True, drop=(4, 0.2, 7, E)))
    print('%-34s %8.3f ms  %6.1f TF' % ('NN +scatter +dropout', ms, fl / ms / 1e9))
    # TN (dW) split-K
    dw = torch.zeros(NP2, E, device=d)
    sk = ops.split_for(NP2, E, M)
    ms = timeit(lambda: ops.gemm(dg, x, dw, M=NP2, N=E, K=M, lda=NP2, ldb=E, ldc=E, trans_a=True, trans_b=True, split_k=sk, atomic=True))
    print('%-34s %8.3f ms  %6.1f TF  (split %d)' % ('TN 1664x300x131072', ms, fl / ms / 1e9, sk))
    ms = timeit(lambda: ops.gemm(dg, emb, dw, M=NP2, N=E, K=M, lda=NP2, ldb=E, ldc=E, trans_a=True, trans_b=True, split_k=sk, atomic=True,
                                 b_idx=idx, drop=(2, 0.2, 7, E)))
    print('%-34s %8.3f ms  %6.1f TF' % ('TN +gather +dropout', ms, fl / ms / 1e9))


if __name__ == '__main__':
    main()
