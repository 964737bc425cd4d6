import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nnr_amd import ops
from tools.gemm_bench import timeit
d = torch.device('cuda')
cap, live, E, NP2 = 450560, 80000, 300, 1664
x = torch.randn(cap, E, device=d); w = torch.randn(NP2, E, device=d) * 0.05; out = torch.empty(cap, NP2, device=d)
dyn = torch.tensor([live], device=d, dtype=torch.int32)
fl = 2.0 * live * NP2 * E
ms = timeit(lambda: ops.gemm(x, w, out, M=live, N=NP2, K=E, lda=E, ldb=E, ldc=NP2))
print('static M = live            %7.3f ms %6.1f TF' % (ms, fl / ms / 1e9))
ms = timeit(lambda: ops.gemm(x, w, out, M=cap, N=NP2, K=E, lda=E, ldb=E, ldc=NP2, dyn=dyn, dyn_dim=1))
print('static M = capacity + dyn  %7.3f ms %6.1f TF' % (ms, fl / ms / 1e9))
h = torch.randn(cap, 400, device=d); wh = torch.randn(400, 400, device=d); oh = torch.empty(cap, 400, device=d)
fl = 2.0 * live * 400 * 400
ms = timeit(lambda: ops.gemm(h, wh, oh, M=live, N=400, K=400, lda=400, ldb=400, ldc=400))
print('400x400 static             %7.3f ms %6.1f TF' % (ms, fl / ms / 1e9))
ms = timeit(lambda: ops.gemm(h, wh, oh, M=cap, N=400, K=400, lda=400, ldb=400, ldc=400, dyn=dyn, dyn_dim=1))
print('400x400 capacity + dyn     %7.3f ms %6.1f TF' % (ms, fl / ms / 1e9))
