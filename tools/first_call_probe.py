import sys, ctypes as C
sys.path.insert(0, '/root/repo')
order = sys.argv[1]
if order == "lib_first":  # (nnr_amd._lib.lib() now imports torch itself before loading the library)
    from nnr_amd import _lib
    L = _lib.lib(); L.nnr_version()
import torch
from nnr_amd import _lib, ops
x = torch.ones(1 << 20, device='cuda')
s = torch.cuda.Stream()
for name, ctx in (('side', torch.cuda.stream(s)), ('default', torch.cuda.stream(torch.cuda.current_stream()))):
    with ctx:
        try:
            ops.fill_zero(x)
            torch.cuda.synchronize()
            print(order, name, 'fill_zero ok', float(x.sum()))
        except Exception as e:
            print(order, name, 'FAILED', e)
        x.fill_(1.0)
        y = ops.dropout(x, 0.5, 123)
        torch.cuda.synchronize()
        print(order, name, 'kernel launch ok', float(y.sum()) > 0)
        try:
            ops.fill_zero(x); torch.cuda.synchronize(); print(order, name, 'fill_zero after a launch ok')
        except Exception as e:
            print(order, name, 'fill after launch FAILED', e)
        x.fill_(1.0)
