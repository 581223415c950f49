"""pyg 2.0.4 GraphNorm semantics when called with batch=None (whole input = one graph)."""
import torch


class GraphNorm(torch.nn.Module):
    def __init__(self, in_channels, eps=1e-5):
        super().__init__()
        self.in_channels = in_channels
        self.eps = eps
        self.weight = torch.nn.Parameter(torch.ones(in_channels))
        self.bias = torch.nn.Parameter(torch.zeros(in_channels))
        self.mean_scale = torch.nn.Parameter(torch.ones(in_channels))

    def forward(self, x, batch=None):
        if batch is not None:
            raise NotImplementedError('stub: only the batch=None call the reference makes')
        mean = x.mean(dim=0, keepdim=True)
        out = x - mean * self.mean_scale
        var = out.pow(2).mean(dim=0, keepdim=True)
        std = (var + self.eps).sqrt()
        return self.weight * out / std + self.bias


class LayerNorm(torch.nn.Module):
    pass
