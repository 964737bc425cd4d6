"""Which kernels does the vendor library pick for the SUE shapes (names carry the macro-tile)?  Diagnostic only: nothing in the product calls a vendor BLAS."""
import torch
d = torch.device('cuda')
for M, N, K in ((4352, 900, 900), (6080, 900, 900), (131072, 400, 400)):
    a = torch.randn(M, K, device=d); b = torch.randn(N, K, device=d); c = torch.empty(M, N, device=d)
    for _ in range(3):
        torch.mm(a, b.t(), out=c)
    torch.cuda.synchronize()
