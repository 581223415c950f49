"""Data-parallel gradient exchange: one process per GPU, one flat fp32 all-reduce per step.

The reference is single-process; graphs of a batch are independent, so the path
shards by graph across ranks with replicated weights and the only exchange step is the gradient
sum (RCCL over xGMI through torch.distributed's 'nccl' backend; 'gloo' on CPU in the tests).
Parameters whose gradient is None (the last layer's coord_mlp) are excluded on every
rank identically, so they stay None and Adam keeps skipping them as in the reference.

Every exchanged buffer ends in two extra floats:
  weight  this rank's share of the global batch (its number of graphs; 1 when not given). Gradients
          are packed pre-multiplied by it and divided by the summed weights afterwards, so ranks
          with uneven shards (shard_range hands out base+1 / base graphs) still produce the gradient
          of the mean loss over the GLOBAL batch (each rank's loss is a mean over its own graphs).
  flag    1 when this rank's set of parameters with gradients differs from the agreed layout; the sum
          reaches every rank, so all ranks raise together instead of one raising and the others
          hanging in the next collective.
"""
import contextlib

import torch
import torch.distributed as dist


def _active(group, alone=False):
    """A process group exists and has someone to exchange with. alone=True: also a group of ONE rank runs
    its collectives (an RCCL smoke on a one-GPU box walks the very code of the multi-rank path)."""
    return dist.is_available() and dist.is_initialized() and (alone or dist.get_world_size(group) > 1)


class _Bucket:
    """One flat buffer [grads of `idx` ... | weight | flag] and its in-flight all-reduce."""

    def __init__(self, params, idx):
        self.params, self.idx = params, list(idx)
        dev = params[0].device
        self.numel = sum(params[i].numel() for i in self.idx)
        self.flat = torch.empty(self.numel + 2, dtype=torch.float32, device=dev)
        self.ready, self.handle = 0, None
        # [weight, flag] tails, resident on the device (no host-to-device copy per step)
        self._tail = {False: torch.tensor([1.0, 0.0], dtype=torch.float32, device=dev),
                      True: torch.tensor([1.0, 1.0], dtype=torch.float32, device=dev)}

    def launch(self, weight, flag, group):
        grads, missing = [], False
        for i in self.idx:
            g = self.params[i].grad
            if g is None:               # in the agreed layout, but no gradient on this rank this step
                g, missing = torch.zeros_like(self.params[i]), True
            grads.append(g.reshape(-1))
        tail = self._tail[bool(flag or missing)]
        # pack with one multi-tensor kernel (34 tensors at cfg2: a per-tensor copy loop would cost more
        # launches than the whole all-reduce)
        torch.cat(grads + [tail], out=self.flat)
        if weight != 1.0:
            self.flat[:self.numel + 1].mul_(float(weight))
        self.handle = dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group, async_op=True)

    def finish(self):
        """Waits, scales by 1 / sum of weights, copies back. Returns the summed mismatch flag."""
        self.handle.wait()
        self.handle, self.ready = None, 0
        self.flat[:self.numel].div_(self.flat[self.numel])
        grads, views, offset = [], [], 0
        for i in self.idx:
            n = self.params[i].numel()
            if self.params[i].grad is not None:
                grads.append(self.params[i].grad)
                views.append(self.flat[offset:offset + n].view_as(self.params[i]))
            offset += n
        if grads:
            torch._foreach_copy_(grads, views)
        return self.flat[self.numel + 1]


class GradAllReducer:
    """Flat-bucket all-reduce of the model's non-None gradients (weighted mean over ranks).

    reducer = GradAllReducer(params); ...; loss.backward(); reducer(weight=n_local_graphs)"""

    def __init__(self, params, process_group=None, exchange_when_alone=False):
        self.params = list(params)
        self.group = process_group
        self.alone = bool(exchange_when_alone)
        self._live = None
        self._bucket = None
        self._pending = None

    def _agree_on_layout(self, live):
        # every rank must agree on the bucket layout (deterministic by construction; verified on the
        # first exchange, which every rank reaches, because a mismatch would silently mix parameters)
        dev = self.params[0].device
        sig = torch.tensor([len(live), sum(live), sum(self.params[i].numel() for i in live)],
                           dtype=torch.int64, device=dev)
        lo, hi = sig.clone(), sig.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=self.group)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=self.group)
        if not torch.equal(lo, hi):     # the same verdict on every rank
            raise RuntimeError('ranks disagree on which parameters have gradients')
        self._live = live

    def _raise_if_flagged(self, flags):
        """The summed flags are read ONE CALL LATE on a GPU (pinned copy + event, like
        PreparedGraph.poll_status), so the step never waits for the host; every rank sees the same
        sums at the same call and raises together. `check()` drains the pending read."""
        self.check()
        total = torch.stack([f.reshape(()) for f in flags]).sum()
        if not total.is_cuda:
            self._pending = (None, total)
            return self.check()
        host = torch.empty((), dtype=torch.float32, pin_memory=True)
        host.copy_(total, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(total.device))
        self._pending = (ev, host)

    def check(self):
        pending, self._pending = getattr(self, '_pending', None), None
        if pending is None:
            return
        ev, host = pending
        if ev is not None:
            ev.synchronize()
        if float(host) != 0.0:
            raise RuntimeError('the set of parameters with gradients changed between steps (on at '
                               'least one rank; every rank raises this together)')

    def reset(self):
        """Forget the agreed layout: the next call learns (and cross-checks) it again. For a change of
        the set of parameters that receive gradients which every rank makes at the same step, e.g.
        MultitaskSatorrasEGNN switching from its pose head to its affinity head."""
        self.check()
        self._live = self._bucket = None

    def __call__(self, weight=1.0):
        if not _active(self.group, self.alone):
            return
        live = [i for i, p in enumerate(self.params) if p.grad is not None]
        if self._live is None:
            self._agree_on_layout(live)
            self._bucket = _Bucket(self.params, live)
        self._bucket.launch(weight, live != self._live, self.group)
        self._raise_if_flagged([self._bucket.finish()])


class OverlappedGradAllReducer:
    """The same exchange, started while the backward is still running (BASELINE config 4: "RCCL grad
    all-reduce overlapped with backward"). Parameters are cut into `n_buckets` groups in reverse
    order (the backward reaches the head and the last layers first); a post-accumulate hook counts
    the gradients of a bucket as they land and launches its asynchronous all-reduce when the
    bucket is complete. The LAST bucket (the first layers, complete only when the backward ends) is
    always launched by `reducer()`, which is also where a changed gradient set is flagged.
    The first step runs the plain flat exchange: it learns which parameters receive gradients
    (None-gradient parameters never fire a hook) and checks that all ranks agree.

    Gradient accumulation (several backward passes per optimiser step): wrap all but the last
    backward in `with reducer.no_sync():`. A second backward outside no_sync() is detected (a hook
    fires for a bucket that is already in flight) and that bucket is exchanged again from the
    accumulated gradients, so nothing is dropped - at the price of the wasted first exchange.

    Use: reducer = OverlappedGradAllReducer(params); ...; loss.backward(); reducer(weight=n_local)
    `weight` of the overlapped buckets is the one given to the PREVIOUS call or set_weight()."""

    def __init__(self, params, n_buckets=2, process_group=None, exchange_when_alone=False):
        self.params = list(params)
        self.group = process_group
        self.alone = bool(exchange_when_alone)
        self._flat_fallback = GradAllReducer(self.params, process_group, exchange_when_alone)
        self._buckets = None
        self._bucket_of = {}
        self.n_buckets = max(1, int(n_buckets))
        self._sync = True
        self._weight = 1.0
        self._hooks = [p.register_post_accumulate_grad_hook(self._make_hook(i))
                       for i, p in enumerate(self.params)]

    def set_weight(self, weight):
        """Share of the global batch held by this rank for the coming step(s) (its graph count)."""
        self._weight = float(weight)

    @contextlib.contextmanager
    def no_sync(self):
        """Backward passes inside accumulate into .grad without any exchange."""
        prev, self._sync = self._sync, False
        try:
            yield
        finally:
            self._sync = prev

    def _make_hook(self, i):
        def hook(_param):
            if self._buckets is None or not self._sync:
                return
            b = self._bucket_of.get(i)
            if b is None:
                return
            b.ready += 1
            k = self._buckets.index(b)
            # collectives must be issued in the same order on every rank: bucket k only after
            # buckets 0..k-1; the last bucket waits for reducer()
            if (b.ready == len(b.idx) and b.handle is None and k < len(self._buckets) - 1
                    and all(prev.handle is not None for prev in self._buckets[:k])):
                b.launch(self._weight, False, self.group)
        return hook

    def _plan(self):
        live = self._flat_fallback._live
        order = list(reversed(live))                      # backward order: last parameters first
        per = -(-len(order) // self.n_buckets)
        self._buckets = [_Bucket(self.params, order[k:k + per]) for k in range(0, len(order), per)]
        self._bucket_of = {i: b for b in self._buckets for i in b.idx}

    def __call__(self, weight=None):
        if not _active(self.group, self.alone):
            return
        if weight is not None:
            self._weight = float(weight)
        if self._buckets is None:
            self._flat_fallback(self._weight)        # first step: plain exchange + agreement check
            self._plan()
            return
        live_now = [i for i, p in enumerate(self.params) if p.grad is not None]
        changed = live_now != self._flat_fallback._live
        for k, b in enumerate(self._buckets):
            if b.handle is not None and b.ready > len(b.idx):
                # a further backward accumulated into .grad after this bucket was packed: drop that
                # exchange (p.grad still holds the local sums) and run it again
                b.handle.wait()
                b.handle = None
            if b.handle is None:      # incomplete at hook time, the last bucket, or hooks did not fire
                b.launch(self._weight, changed and k == len(self._buckets) - 1, self.group)
        self._flat_fallback._raise_if_flagged([b.finish() for b in self._buckets])


    def check(self):
        """Drains the pending (one call late) mismatch check."""
        self._flat_fallback.check()

    def reset(self):
        """Forget the bucket plan (see GradAllReducer.reset): the next step runs the flat exchange again,
        learns which parameters receive gradients now and re-plans the buckets. Call it on every rank at
        the same step (PointNeuralNetworkBase.set_task does when the task changes); an exchange must
        not be in flight."""
        for b in self._buckets or ():
            if b.handle is not None:
                b.handle.wait()
                b.handle = None
            b.ready = 0
        self._flat_fallback.reset()
        self._buckets, self._bucket_of = None, {}


def shard_range(n_items, rank, world):
    """Contiguous, balanced slice of n_items for `rank` (graphs are independent units)."""
    base, extra = divmod(n_items, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)
