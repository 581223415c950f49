"""`torch.optim.Adam` whose step (and the `clip_grad_value_` the reference runs just before it,
/root/reference/point_vs/models/point_neural_network_base.py:421-422) is ONE kernel launch for all
parameters (`pvs_adam_clip_step`) instead of torch's dozen multi-tensor
launches. Same update rule, same `state_dict` layout (`step`, `exp_avg`, `exp_avg_sq`), so
checkpoints stay interchangeable with the reference's Adam."""
import torch

from . import _lib


class FusedClipAdam(torch.optim.Adam):
    """Drop-in `torch.optim.Adam`; `step(clip_value=c)` first clamps every gradient to [-c, c] in
    place. Falls back to torch's own implementation whenever a feature the kernel does not cover is
    on (amsgrad, maximize, differentiable, a tensor learning rate, non-CUDA / non-fp32 parameters).
    `capturable=True` (round 5) keeps the step counters on the device exactly as torch's capturable Adam does (same
    `state_dict`), advances them with one foreach launch and lets the kernel form the bias corrections from them
    (`pvs_adam_clip_step_dev`): the whole step is then two launches that a hipGraph can replay."""

    def __init__(self, params, **kwargs):
        super().__init__(params, **kwargs)
        self._reset_transients()

    def _reset_transients(self):
        # pointer tables travel through a small ring of pinned buffers: a slot is only rewritten once
        # the copy that read it has run (the host may be several steps ahead of the device)
        self._ring, self._slot = [], 0
        self._capture_pool, self._capture_next, self._captured_tables = None, 0, []
        self._recent = {}              # (device, the table's rows) -> ring slot that holds that table on the device
        self._fast = None

    def __setstate__(self, state):
        """copy.deepcopy / pickle of an optimiser carry `defaults`, `state` and `param_groups` only (Optimizer.__getstate__):
        the copy starts with its own empty upload ring and no work list."""
        super().__setstate__(state)
        self._reset_transients()

    def _fusable(self):
        for group in self.param_groups:
            if group.get('amsgrad') or group.get('maximize') \
                    or group.get('differentiable') or group.get('decoupled_weight_decay') \
                    or isinstance(group['lr'], torch.Tensor):
                return False
            for p in group['params']:
                if p.grad is not None and (not p.is_cuda or p.dtype != torch.float32 or p.grad.is_sparse
                                           or not p.is_contiguous() or not p.grad.is_contiguous()):
                    return False
        return True

    def reserve_capture_tables(self, k):
        """Pinned pointer tables for `k` more captured steps (pinned memory cannot be allocated while a capture is open,
        and a captured upload must keep its source for as long as the graph is replayed)."""
        rows = max(64, sum(len(g['params']) for g in self.param_groups))
        left = 0 if self._capture_pool is None else self._capture_pool.shape[0] - self._capture_next
        if left < k:
            self._capture_pool = torch.empty((k, rows, 5), dtype=torch.int64).pin_memory()   # (tables handed out stay alive
            self._capture_next = 0                                                            # in _captured_tables)

    def zero_grad(self, set_to_none=True):
        """torch's zero_grad walks hooks, profiler ranges and foreach groups (~100 us on the host for 34 parameters);
        with set_to_none this is all it does."""
        if not set_to_none:
            return super().zero_grad(set_to_none=False)
        for group in self.param_groups:
            for p in group['params']:
                p.grad = None

    def load_state_dict(self, state_dict):
        """torch's own, then the step counters where THIS optimiser keeps them: on the host for a non-capturable group
        (a checkpoint read with `map_location=<device>` - as the reference's load_weights reads it,
        point_neural_network_base.py:532 - brings them in as device tensors, torch keeps them there, and torch's Adam
        then reads one counter per parameter back to the host every step; the fused step would refuse the state and fall
        back to exactly that), on the device as fp32 for a capturable one."""
        self._fast = None              # new state tensors, other step counts
        out = super().load_state_dict(state_dict)
        for group in self.param_groups:
            capturable = bool(group.get('capturable'))
            for p in group['params']:
                st = self.state.get(p)
                if not st or not torch.is_tensor(st.get('step')):
                    continue
                if capturable and (not st['step'].is_cuda or st['step'].dtype != torch.float32):
                    st['step'] = st['step'].to(device=p.device, dtype=torch.float32)
                elif not capturable and (st['step'].is_cuda or st['step'].dtype != torch.float32):
                    st['step'] = st['step'].detach().to(device='cpu', dtype=torch.float32)
        return out

    @torch.no_grad()
    def step(self, closure=None, clip_value=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        # Host time counts (small batches are bound by it: tools/host_profile.py): the work list - which parameters
        # have gradients, their state tensors, their common step count, the fusability verdict - is kept from step to
        # step and re-derived only when the set of parameters with gradients (or the groups, or the state) changes;
        # the step counters (one 0-dim CPU tensor per parameter: torch's state_dict layout) are views of one tensor and
        # advance with one add_; a pointer table already on the device is found again by the (parameter, gradient)
        # addresses alone.
        live = [p.grad is not None for group in self.param_groups for p in group['params']]
        fast = getattr(self, '_fast', None)
        capt = [bool(g.get('capturable')) for g in self.param_groups]
        if fast is None or fast['live'] != live or fast['capt'] != capt:
            fast = self._plan(live)
        keys = self._address_keys(fast) if fast['fusable'] else None
        if keys is None:
            self._fast = None
            if clip_value is not None:
                torch.nn.utils.clip_grad_value_([p for g in self.param_groups for p in g['params']], clip_value)
            super().step()
            return loss
        lib = _lib.lib()
        all_works = [work for works in fast['groups'] for work in works]
        # tensors at addresses not seen before are looked at (device, layout, sparsity) BEFORE anything is launched
        if any(self._recent.get(key) is None and not self._tensors_fusable(work) for work, key in zip(all_works, keys)):
            self._fast = None
            if clip_value is not None:
                torch.nn.utils.clip_grad_value_([p for g in self.param_groups for p in g['params']], clip_value)
            super().step()
            return loss
        keys = iter(keys)
        for group, works in zip(self.param_groups, fast['groups']):
            beta1, beta2 = group['betas']
            for work in works:            # one launch per distinct step count (one, unless the gradient set changed mid-run)
                key = next(keys)
                if work['steps_base'] is not None:
                    work['steps_base'].add_(1)               # (host counters: every parameter's `step` is a view of this tensor)
                else:
                    torch._foreach_add_(work['steps'], 1)    # (capturable: one device launch)
                work['step'] += 1
                step, n, dev = work['step'], work['n'], work['dev']
                capturing = work['on_device'] and torch.cuda.is_current_stream_capturing()
                if not self._ring or self._ring[0][0].shape[0] < n:
                    cap = max(n, 64)
                    self._ring = [[torch.empty((cap, 5), dtype=torch.int64).pin_memory(),
                                   torch.empty((cap, 5), dtype=torch.int64, device=dev), None] for _ in range(8)]
                    self._recent = {}
                if work['on_device'] and not capturing and self._capture_pool is None:
                    self.reserve_capture_tables(8)
                hit = None if capturing else self._recent.get(key)
                if hit is not None:
                    # addresses seen before (the caching allocator hands the gradients the same few sets of blocks,
                    # alternating from step to step): that table is still on the device - no upload, and nothing about
                    # these very tensors has to be checked or listed again
                    table_dev = self._ring[hit][1]
                else:
                    rows = [[p.data_ptr(), p.grad.data_ptr(), ea.data_ptr(), es.data_ptr(), p.numel()]
                            for p, ea, es in work['items']]
                    if capturing:
                        # a captured upload is replayed from its pinned source: that buffer belongs to the capture and is
                        # never rewritten (a ring slot would be, by the next eager step)
                        if self._capture_pool is None or self._capture_next >= self._capture_pool.shape[0]:
                            raise RuntimeError('FusedClipAdam: no pinned table left for this capture - call '
                                               'reserve_capture_tables(k) (or run one eager capturable step) before capturing')
                        table_host = self._capture_pool[self._capture_next]
                        self._capture_next += 1
                        table_host[:n] = torch.tensor(rows, dtype=torch.int64)
                        table_dev = torch.empty((n, 5), dtype=torch.int64, device=dev)
                        table_dev.copy_(table_host[:n], non_blocking=True)
                        self._captured_tables.append((table_host, table_dev))
                    else:
                        idx = self._slot
                        slot = self._ring[idx]
                        self._slot = (self._slot + 1) % len(self._ring)
                        if slot[2] is not None:
                            slot[2].synchronize()
                        # (the slot's previous table is forgotten before it is overwritten)
                        self._recent = {k: v for k, v in self._recent.items() if v != idx}
                        table_host, table_dev = slot[0], slot[1]
                        table_host[:n] = torch.tensor(rows, dtype=torch.int64)
                        table_dev[:n].copy_(table_host[:n], non_blocking=True)
                        slot[2] = torch.cuda.Event()
                        slot[2].record(torch.cuda.current_stream(dev))
                        self._recent[key] = idx
                if work['on_device']:
                    _lib.check(lib.pvs_adam_clip_step_dev(
                        _lib.ptr(table_dev), n, float(group['lr']), float(beta1), float(beta2),
                        float(group['eps']), float(group['weight_decay']), work['steps'][0].data_ptr(),
                        float(clip_value) if clip_value is not None else 0.0,
                        _lib.stream(dev)), 'pvs_adam_clip_step_dev')
                else:
                    _lib.check(lib.pvs_adam_clip_step(
                        _lib.ptr(table_dev), n, float(group['lr']), float(beta1), float(beta2),
                        float(group['eps']), float(group['weight_decay']), 1.0 - beta1 ** step, 1.0 - beta2 ** step,
                        float(clip_value) if clip_value is not None else 0.0,
                        _lib.stream(dev)), 'pvs_adam_clip_step')
                # the kernel wrote through raw pointers: tell autograd (and anything that caches by
                # version, e.g. ReceptorScreen) that the parameters changed, as an in-place op would
                torch.autograd.graph.increment_version(work['params'])
        return loss

    def _plan(self, live):
        """The work list of step(): per group, per distinct step count, the parameters with gradients and their state
        (created as torch.optim.Adam._init_group does, non-capturable flavour)."""
        groups, fusable = [], self._fusable()
        self._recent = {}      # (remembered tables hold the OLD plan's state-tensor addresses: exp_avg / exp_avg_sq are not in the key)
        if not fusable:       # (torch's own step creates whatever state its flavour - capturable, amsgrad ... - needs)
            self._fast = {'live': live, 'capt': [bool(g.get('capturable')) for g in self.param_groups], 'groups': [],
                          'fusable': False}
            return self._fast
        for group in self.param_groups:
            by_step = {}
            for p in group['params']:
                if p.grad is None:
                    continue
                state = self.state[p]
                capturable = bool(group.get('capturable'))
                if len(state) == 0:
                    state['step'] = (torch.zeros((), dtype=torch.float32, device=p.device) if capturable
                                     else torch.tensor(0.0, dtype=torch.float32))
                    state['exp_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    state['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                if (not isinstance(state['step'], torch.Tensor) or state['step'].dtype != torch.float32
                        or state['step'].is_cuda != capturable):
                    fusable = False        # (a state the other flavour left behind: torch's own step sorts it out)
                    continue
                # (capturable: one host read per parameter when the plan is made - never under capture, the plan exists
                # from the eager steps before it)
                by_step.setdefault((float(state['step']), capturable), []).append((p, state))
            works = []
            for (count, on_device), members in by_step.items():
                base = None
                if not on_device:
                    # host counters: torch's state_dict layout wants one 0-dim fp32 tensor per parameter; they become
                    # views of ONE tensor, so that a single add_ advances them all (78 counters of a 6-layer model cost
                    # 0.1 ms per step through _foreach_add_: CPU tensors take its slow path)
                    base = torch.full((len(members),), float(count), dtype=torch.float32)
                    for k, (_, st) in enumerate(members):
                        st['step'] = base[k]
                works.append({'items': [(p, st['exp_avg'], st['exp_avg_sq']) for p, st in members],
                              'steps': [st['step'] for _, st in members], 'step': int(count), 'n': len(members),
                              'dev': members[0][0].device, 'params': [p for p, _ in members], 'on_device': on_device,
                              'steps_base': base})
            groups.append(works)
        self._fast = {'live': live, 'capt': [bool(g.get('capturable')) for g in self.param_groups], 'groups': groups,
                      'fusable': fusable}
        return self._fast

    def _address_keys(self, fast):
        """Per work item the tuple of (parameter, gradient) addresses of this step - what the pointer table on the device
        is remembered by - or None when the fused kernel must not run: a group option it does not cover switched on
        mid-run (amsgrad, maximize, ... or a tensor learning rate), or a parameter / gradient that is not fp32 any more
        (`model.half()` / `.double()` after the first step: the kernel takes raw pointers as fp32 device memory). What
        else can change under an unchanged plan (device, layout, sparsity) is checked by `_tensors_fusable` whenever the
        addresses are new; tensors at addresses seen before ARE the tensors that passed it."""
        f32 = torch.float32
        keys = []
        for group, works in zip(self.param_groups, fast['groups']):
            if group.get('amsgrad') or group.get('maximize') or group.get('differentiable') \
                    or group.get('decoupled_weight_decay') or isinstance(group['lr'], torch.Tensor):
                return None
            for work in works:
                key = [work['dev']]
                for p in work['params']:
                    g = p.grad
                    if p.dtype is not f32 or g.dtype is not f32:
                        return None
                    key.append(p.data_ptr())
                    key.append(g.data_ptr())
                keys.append(tuple(key))
        return keys

    @staticmethod
    def _tensors_fusable(work):
        for p in work['params']:
            g = p.grad
            if (not p.is_cuda or not g.is_cuda or g.is_sparse or g.layout is not torch.strided
                    or not g.is_contiguous() or not p.is_contiguous()):
                return False
        return True
