#!/bin/bash
# round 6, GPU call 8: which shapes of the rewritten bf16x3 kernel fail (no -x), and the headline parity test
mkdir -p gpurun_out
export PYTHONWARNINGS=ignore
(timeout 900 python -m pytest tests/test_hip_ops_gpu.py -m gpu -q --tb=line -k "bf16x3" 2>&1 | grep -v amdgpu.ids | tail -25) > gpurun_out/r06h_tests.log
cat gpurun_out/r06h_tests.log | cut -c1-250
python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06h_probe.txt
import torch, math
from nnr_amd import ops
d = torch.device('cuda')
for (M, N, K) in [(256, 80, 32), (256, 80, 64), (256, 80, 48), (256, 160, 64), (70000, 400, 400), (70000, 400, 384), (128, 80, 384), (300, 84, 96)]:
    g = torch.Generator().manual_seed(1)
    a = torch.randn(M, K, generator=g).to(d); b = (torch.randn(N, K, generator=g) * 0.2).to(d)
    img, stride, ldo = ops.bx3_images(b, N, K, K)
    out = torch.empty(M, N, device=d)
    ops.gemm(a, b, out, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, tile=50, b3=(img, stride, ldo))
    full = a.double() @ b.double().t()
    err = (out.double() - full).abs()
    rel = float(err.norm() / full.norm())
    bad = (err > 1e-3 * full.abs().max())
    print(M, N, K, 'rel', rel, 'bad rows', int(bad.any(1).sum()), 'bad cols', int(bad.any(0).sum()),
          'first bad rows', torch.nonzero(bad.any(1)).flatten()[:12].tolist(), 'first bad cols', torch.nonzero(bad.any(0)).flatten()[:12].tolist())
PY
