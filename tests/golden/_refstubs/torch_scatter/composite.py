"""pytorch-scatter 2.1.0 composite.scatter_softmax semantics for dim=0 and a 1-D index."""
import torch


def scatter_softmax(src, index, dim=0, eps=1e-12):
    if dim != 0:
        raise NotImplementedError
    idx = index
    while idx.dim() < src.dim():
        idx = idx.unsqueeze(-1)
    idx = idx.expand_as(src)
    groups = int(index.max()) + 1
    shape = (groups,) + tuple(src.shape[1:])
    gmax = torch.full(shape, float('-inf'), dtype=src.dtype, device=src.device)
    gmax = gmax.scatter_reduce(0, idx, src, 'amax', include_self=True)
    shifted = (src - gmax.gather(0, idx)).exp()
    gsum = torch.zeros(shape, dtype=src.dtype, device=src.device).scatter_add(0, idx, shifted)
    return shifted / gsum.gather(0, idx)
