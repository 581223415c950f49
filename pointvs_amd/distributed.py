"""Data-parallel gradient exchange: one process per GPU, one flat fp32 all-reduce per step.

The reference is single-process; graphs of a batch are independent, so the path
shards by graph across ranks with replicated weights and the only exchange step is the gradient
sum (RCCL over xGMI through torch.distributed's 'nccl' backend; 'gloo' on CPU in the tests).
Parameters whose gradient is None (the last layer's coord_mlp) are excluded on every
rank identically, so they stay None and Adam keeps skipping them as in the reference.
"""
import torch
import torch.distributed as dist


class GradAllReducer:
    """Flat-bucket all-reduce (sum, then 1/world) of the model's non-None gradients."""

    def __init__(self, params, process_group=None):
        self.params = list(params)
        self.group = process_group
        self._flat = None
        self._live = None

    def __call__(self):
        if not dist.is_available() or not dist.is_initialized():
            return
        world = dist.get_world_size(self.group)
        if world == 1:
            return
        live = [i for i, p in enumerate(self.params) if p.grad is not None]
        if self._live != live:
            # every rank must agree on the bucket layout (deterministic by construction; verified
            # once because a mismatch would silently mix parameters)
            sig = torch.tensor([len(live), sum(live), sum(self.params[i].numel() for i in live)],
                               dtype=torch.int64, device=self.params[0].device)
            lo, hi = sig.clone(), sig.clone()
            dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=self.group)
            dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=self.group)
            if not torch.equal(lo, hi):
                raise RuntimeError('ranks disagree on which parameters have gradients')
            self._live = live
            total = sum(self.params[i].numel() for i in live)
            self._flat = torch.empty(total, dtype=torch.float32, device=self.params[0].device)
        grads = [self.params[i].grad for i in live]
        # pack / unpack with one multi-tensor kernel each (34 tensors at cfg2: a per-tensor copy loop
        # would cost more launches than the whole all-reduce)
        torch.cat([g.reshape(-1) for g in grads], out=self._flat)
        dist.all_reduce(self._flat, op=dist.ReduceOp.SUM, group=self.group)
        self._flat.mul_(1.0 / world)
        views, offset = [], 0
        for g in grads:
            n = g.numel()
            views.append(self._flat[offset:offset + n].view_as(g))
            offset += n
        torch._foreach_copy_(grads, views)


def shard_range(n_items, rank, world):
    """Contiguous, balanced slice of n_items for `rank` (graphs are independent units)."""
    base, extra = divmod(n_items, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)
