import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nnr_amd import ops
from nnr_amd.synth import _zipf_ids
from tools.gemm_bench import timeit
d = torch.device('cuda')
M, E, NP2, V = 131072, 300, 1664, 60000
rng = np.random.default_rng(0)
uni = torch.randint(0, V, (M,), device=d, dtype=torch.int32)
zipf = torch.from_numpy(_zipf_ids(rng, M, 2, V, 1.1)).to(d)
print('zipf top-id share', float((zipf == zipf.mode().values).float().mean()))
dg = torch.randn(M, NP2, device=d); w = torch.randn(NP2, E, device=d) * 0.05; dtab = torch.zeros(V, E, device=d)
emb = torch.randn(V, E, device=d); out = torch.empty(M, NP2, device=d); dw = torch.zeros(NP2, E, device=d); bias = torch.randn(NP2, device=d)
x = torch.empty(M, E, device=d)
fl = 2.0 * M * E * NP2
for name, ids in (('uniform', uni), ('zipf', zipf)):
    for drop in (None, (4, 0.2, 7, E)):
        ms = timeit(lambda: ops.gemm(dg, w, dtab, M=M, N=E, K=NP2, lda=NP2, ldb=E, ldc=E, trans_b=True, c_idx=ids, atomic=True, drop=drop))
        print('NN scatter %-8s drop=%-5s %7.3f ms %6.1f TF' % (name, drop is not None, ms, fl / ms / 1e9))
    for drop in (None, (1, 0.2, 7, E)):
        ms = timeit(lambda: ops.gemm(emb, w, out, M=M, N=NP2, K=E, lda=E, ldb=E, ldc=NP2, a_idx=ids, bias=bias, drop=drop))
        print('NT gather  %-8s drop=%-5s %7.3f ms %6.1f TF' % (name, drop is not None, ms, fl / ms / 1e9))
    for drop in (None, (2, 0.2, 7, E)):
        ms = timeit(lambda: ops.gemm(dg, emb, dw, M=NP2, N=E, K=M, lda=NP2, ldb=E, ldc=E, trans_a=True, trans_b=True, split_k=56, atomic=True, b_idx=ids, drop=drop))
        print('TN gather  %-8s drop=%-5s %7.3f ms %6.1f TF' % (name, drop is not None, ms, fl / ms / 1e9))
ms = timeit(lambda: ops.gemm(dg, w, x, M=M, N=E, K=NP2, lda=NP2, ldb=E, ldc=E, trans_b=True))
print('NN plain store                  %7.3f ms %6.1f TF' % (ms, fl / ms / 1e9))
