import torch
from . import norm  # noqa: F401
from .norm import GraphNorm  # noqa: F401


def global_mean_pool(x, batch, size=None):
    """Segment mean of node rows by graph id (pyg 2.0.4 semantics)."""
    size = int(batch.max()) + 1 if size is None else int(size)
    total = torch.zeros(size, x.size(1), dtype=x.dtype, device=x.device).index_add_(0, batch, x)
    count = torch.zeros(size, dtype=x.dtype, device=x.device).index_add_(
        0, batch, torch.ones(batch.numel(), dtype=x.dtype, device=x.device))
    return total / count.clamp(min=1).unsqueeze(-1)


class MessagePassing(torch.nn.Module):
    def __init__(self, **kwargs):
        super().__init__()
