"""Pure-torch stand-in for the two torch_scatter==2.0.9 entry points the reference calls
(userEncoders.py:88-89).  Used ONLY by tools/make_goldens.py in the build container, where the real
package is absent and cannot be installed; it follows torch_scatter's documented composite
(scatter_sum = zeros.scatter_add_; scatter_softmax = exp(x - group max) / group sum)."""
import torch


def scatter_sum(src, index, dim=-1, out=None, dim_size=None):
    if dim < 0:
        dim += src.dim()
    idx = index
    while idx.dim() < src.dim():
        idx = idx.unsqueeze(-1)
    idx = idx.expand_as(src)
    size = list(src.shape)
    size[dim] = int(index.max()) + 1 if dim_size is None else dim_size
    return torch.zeros(size, dtype=src.dtype, device=src.device).scatter_add_(dim, idx, src)


def scatter_softmax(src, index, dim=-1):
    if dim < 0:
        dim += src.dim()
    idx = index.expand_as(src)
    size = list(src.shape)
    size[dim] = int(index.max()) + 1
    gmax = torch.full(size, float('-inf'), dtype=src.dtype).scatter_reduce_(dim, idx, src.detach(), 'amax', include_self=True)
    e = torch.exp(src - gmax.gather(dim, idx))
    gsum = torch.zeros(size, dtype=src.dtype).scatter_add_(dim, idx, e)
    return e / gsum.gather(dim, idx)
