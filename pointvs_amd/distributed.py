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


class OverlappedGradAllReducer:
    """The same exchange, started while the backward is still running (BASELINE config 4: "RCCL grad
    all-reduce overlapped with backward"). Parameters are cut into `n_buckets` groups in reverse
    order (the backward reaches the head and the last layers first); a post-accumulate hook counts
    the gradients of a bucket as they land and launches its asynchronous all-reduce when the
    bucket is complete, so only the first layers' bucket is exchanged after the backward ends.
    The first step runs the plain flat exchange: it learns which parameters receive gradients
    (None-gradient parameters never fire a hook) and checks that all ranks agree.

    Use: reducer = OverlappedGradAllReducer(params); ...; loss.backward(); reducer()   # = finish"""

    def __init__(self, params, n_buckets=2, process_group=None):
        self.params = list(params)
        self.group = process_group
        self._flat_fallback = GradAllReducer(self.params, process_group)
        self._buckets = None          # list of dicts: idx (param indices), flat, ready, handle
        self._bucket_of = {}
        self.n_buckets = max(1, int(n_buckets))
        self._hooks = [p.register_post_accumulate_grad_hook(self._make_hook(i))
                       for i, p in enumerate(self.params)]

    def _make_hook(self, i):
        def hook(_param):
            if self._buckets is None:
                return
            b = self._bucket_of.get(i)
            if b is None:
                return
            b['ready'] += 1
            # collectives must be issued in the same order on every rank: bucket k only after
            # buckets 0..k-1 (anything else waits for the end of the backward)
            k = self._buckets.index(b)
            if (b['ready'] == len(b['idx']) and b['handle'] is None
                    and all(prev['handle'] is not None for prev in self._buckets[:k])):
                self._launch(b)
        return hook

    def _launch(self, b):
        grads = [self.params[i].grad for i in b['idx']]
        torch.cat([g.reshape(-1) for g in grads], out=b['flat'])
        b['handle'] = dist.all_reduce(b['flat'], op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def _plan(self):
        live = self._flat_fallback._live
        order = list(reversed(live))                      # backward order: last parameters first
        per = -(-len(order) // self.n_buckets)
        dev = self.params[0].device
        self._buckets, self._bucket_of = [], {}
        for k in range(0, len(order), per):
            idx = order[k:k + per]
            b = dict(idx=idx, ready=0, handle=None,
                     flat=torch.empty(sum(self.params[i].numel() for i in idx), dtype=torch.float32, device=dev))
            self._buckets.append(b)
            for i in idx:
                self._bucket_of[i] = b

    def __call__(self):
        if not dist.is_available() or not dist.is_initialized():
            return
        world = dist.get_world_size(self.group)
        if world == 1:
            return
        if self._buckets is None:
            self._flat_fallback()        # first step: plain exchange + agreement check
            self._plan()
            return
        live_now = [i for i, p in enumerate(self.params) if p.grad is not None]
        if live_now != self._flat_fallback._live:
            raise RuntimeError('the set of parameters with gradients changed between steps')
        for b in self._buckets:
            if b['handle'] is None:      # incomplete at hook time (or hooks did not fire): exchange now
                self._launch(b)
        for b in self._buckets:
            b['handle'].wait()
            b['flat'].mul_(1.0 / world)
            views, offset = [], 0
            grads = [self.params[i].grad for i in b['idx']]
            for g in grads:
                n = g.numel()
                views.append(b['flat'][offset:offset + n].view_as(g))
                offset += n
            torch._foreach_copy_(grads, views)
            b['ready'], b['handle'] = 0, None


def shard_range(n_items, rank, world):
    """Contiguous, balanced slice of n_items for `rank` (graphs are independent units)."""
    base, extra = divmod(n_items, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)
