"""torch.autograd.Function wrappers over the C ABI (include/pvs_egnn.h).

PyTorch supplies device memory, the current stream and the autograd tape; every arithmetic step
of the path runs in libpvs_egnn.so.
"""
import ctypes as C

import torch

from . import _lib


def _stream(dev):
    return _lib.stream(dev)


def _ws(nbytes, dev):
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=dev)


def _f32c(t):
    if t is None:
        return None
    if t.dtype != torch.float32:
        raise TypeError(f'pointvs_amd kernels are fp32 (got {t.dtype}); --double is not supported')
    return t.contiguous()


class _PermuteRows(torch.autograd.Function):
    """Rows between input edge order and CSR-sorted order (and back in the backward)."""

    @staticmethod
    def forward(ctx, src, perm, to_input):
        src = _f32c(src)
        _lib.require_hip(src)
        ctx.perm, ctx.to_input = perm, to_input
        width = src.shape[1] if src.dim() > 1 else 1
        dst = torch.empty_like(src)
        fn = _lib.lib().pvs_rows_to_input_order if to_input else _lib.lib().pvs_rows_to_sorted_order
        _lib.check(fn(_lib.ptr(src), _lib.ptr(dst), _lib.ptr(perm), src.shape[0], width,
                      _stream(src.device)), 'pvs_rows_permute')
        return dst

    @staticmethod
    def backward(ctx, g):
        return _PermuteRows.apply(g, ctx.perm, not ctx.to_input), None, None


def rows_to_input_order(src_sorted, pg):
    return _PermuteRows.apply(src_sorted, pg.perm, True)


def rows_to_sorted_order(src_input, pg):
    return _PermuteRows.apply(src_input, pg.perm, False)


class _EGNNLayerFn(torch.autograd.Function):
    """One EGNNLayer.forward (egnn_satorras.py:189-206) and its backward, sorted edge order."""

    @staticmethod
    def forward(ctx, h, x, m_prev, pg, desc_tuple, need_m, pstruct, *params):
        # pstruct: the layer's PvsLayerParams built ONCE for these very parameter tensors (EGNNLayer._params_cached:
        # fp32, contiguous, on the device - checked when it was built), or None (padded / ad-hoc parameter tensors).
        # Host time matters here: at the reference's default shape (32 graphs of 500 atoms) a training step is ~70
        # launches of a few microseconds and the step is bound by this Python (tools/host_profile.py).
        lib = _lib.lib()
        hidden, n_attr, flags, act = desc_tuple
        h, x, m_prev = _f32c(h), _f32c(x), _f32c(m_prev)
        if pstruct is None:
            params = tuple(_f32c(p) for p in params)
            _lib.require_hip(h, x, m_prev, *params)
            pstruct = _lib.PvsLayerParams(*[_lib.ptr(p) for p in params])
        else:
            _lib.require_hip(h, x, m_prev)
        dev = h.device
        n, e = pg.n_nodes, pg.n_edges
        if h.shape != (n, hidden) or x.shape != (n, 3):
            raise ValueError(f'h {tuple(h.shape)} / coord {tuple(x.shape)} do not match N={n}, '
                             f'H={hidden}')
        if n_attr != pg.n_edge_attr:
            raise ValueError(f'layer built with edges_in_d={n_attr} but edge_attr has '
                             f'{pg.n_edge_attr} columns')
        desc = _lib.PvsLayerDesc(hidden, n_attr, flags, act)
        eatt = bool(flags & _lib.EDGE_ATTENTION)
        natt = bool(flags & _lib.NODE_ATTENTION)
        eres = bool(flags & _lib.EDGE_RESIDUAL) and m_prev is not None
        h_out = torch.empty_like(h)
        x_out = torch.empty_like(x)
        m_out = torch.empty((max(e, 0), hidden), dtype=torch.float32, device=dev) if need_m else None
        att = torch.empty((max(e, 1),), dtype=torch.float32, device=dev) if eatt else None
        node_att = torch.empty((n,), dtype=torch.float32, device=dev) if natt else None
        saved = torch.empty(lib.pvs_egnn_layer_saved_floats(C.byref(desc), n, e),
                            dtype=torch.float32, device=dev)
        ws_bytes = lib.pvs_egnn_layer_workspace_bytes(C.byref(desc), n, e, 0)
        ws = _ws(ws_bytes, dev)
        rc = lib.pvs_egnn_layer_fwd(
            C.byref(desc), C.byref(pg.c), C.byref(pstruct), _lib.ptr(h), _lib.ptr(x),
            _lib.ptr(m_prev if eres else None), _lib.ptr(h_out), _lib.ptr(x_out), _lib.ptr(m_out),
            _lib.ptr(att), _lib.ptr(node_att), _lib.ptr(saved), _lib.ptr(ws), ws_bytes, _stream(dev))
        _lib.check(rc, 'pvs_egnn_layer_fwd')
        ctx.pg, ctx.desc_tuple, ctx.eres = pg, desc_tuple, eres
        ctx.n_params = len(params)
        ctx.pstruct = pstruct      # (the backward reads the same parameter tensors: saved below, so they cannot have moved)
        ctx.save_for_backward(h, x, m_prev if eres else None, att, saved, *params)
        ctx.set_materialize_grads(False)
        # absent outputs are None, not empty tensors (h.new_empty(0) cost 22 us apiece on the host: 0.4 ms per step of a
        # 6-layer model)
        if att is not None or node_att is not None:       # (one call: a second call replaces the first one's list)
            ctx.mark_non_differentiable(*[t for t in (att, node_att) if t is not None])
        return h_out, x_out, m_out, att, node_att

    @staticmethod
    def backward(ctx, g_h_out, g_x_out, g_m_out, _g_att, _g_natt):
        lib = _lib.lib()
        h, x, m_prev, att, saved, *params = ctx.saved_tensors
        pg = ctx.pg
        hidden, n_attr, flags, act = ctx.desc_tuple
        dev = h.device
        n, e = pg.n_nodes, pg.n_edges
        desc = _lib.PvsLayerDesc(hidden, n_attr, flags, act)
        pstruct = ctx.pstruct
        g_h_out = torch.zeros_like(h) if g_h_out is None else _f32c(g_h_out)
        g_x_out = _f32c(g_x_out)       # None => the caller never used x_out (last layer)
        g_m_out = _f32c(g_m_out) if (g_m_out is not None and g_m_out.numel()) else None
        need = ctx.needs_input_grad
        g_h = torch.empty_like(h)
        g_x = torch.empty_like(x) if need[1] else None
        g_m_prev = torch.empty_like(m_prev) if ctx.eres else None
        coord_live = bool(flags & _lib.UPDATE_COORDS) and g_x_out is not None
        live = {
            'coord_w1': coord_live, 'coord_b1': coord_live, 'coord_w2': coord_live,
            'edge_gate': ctx.eres and bool(flags & (_lib.REZERO | _lib.GATED_RESIDUAL)),
        }
        grads = []
        for name, p in zip(_lib.PARAM_FIELDS, params):
            if p is None or not live.get(name, True):
                grads.append(None)
            else:
                grads.append(torch.empty_like(p))
        gstruct = _lib.PvsLayerGrads(*[_lib.ptr(g) for g in grads])
        ws_bytes = lib.pvs_egnn_layer_workspace_bytes(C.byref(desc), n, e, 1)
        ws = _ws(ws_bytes, dev)
        rc = lib.pvs_egnn_layer_bwd(
            C.byref(desc), C.byref(pg.c), C.byref(pstruct), _lib.ptr(h), _lib.ptr(x),
            _lib.ptr(m_prev), _lib.ptr(att), _lib.ptr(saved), _lib.ptr(g_h_out), _lib.ptr(g_x_out),
            _lib.ptr(g_m_out), _lib.ptr(g_h), _lib.ptr(g_x), _lib.ptr(g_m_prev), C.byref(gstruct),
            _lib.ptr(ws), ws_bytes, _stream(dev))
        _lib.check(rc, 'pvs_egnn_layer_bwd')
        return (g_h, g_x, g_m_prev, None, None, None, None, *grads)


def egnn_layer(h, x, m_prev_sorted, pg, desc_tuple, need_m, params, pstruct=None):
    """Returns (h_out, x_out, m_sorted|None, att_sorted|None, node_att|None). pstruct: see _EGNNLayerFn.forward."""
    return _EGNNLayerFn.apply(h, x, m_prev_sorted, pg, desc_tuple, need_m, pstruct, *params)


def _align(count, to=64):
    return (int(count) + to - 1) // to * to


class StackPlan:
    """What the one-call layer stack (`pvs_egnn_stack_fwd/bwd`) needs about a model's EGNN layers, built once and kept
    while the layers' parameter structs stay the same objects: the C arrays of descriptors and parameter structs, the flat
    tuple of parameter tensors handed to autograd and, for each of them, (layer, slot of `_lib.PARAM_FIELDS`).
    `pstructs` are the layers' own cached PvsLayerParams (EGNNLayer._params_cached): held here, so `is` against a layer's
    current struct says whether any parameter has moved since."""

    def __init__(self, desc_tuples, param_tuples, pstructs):
        n = len(desc_tuples)
        self.n_layers, self.hidden = n, desc_tuples[0][0]
        self.desc_tuples, self.pstructs = tuple(desc_tuples), tuple(pstructs)
        self.descs = (_lib.PvsLayerDesc * n)(*[_lib.PvsLayerDesc(*d) for d in desc_tuples])
        self.params_c = (_lib.PvsLayerParams * n)(*pstructs)       # (copies of the structs: plain pointers)
        self.any_eatt = any(d[2] & _lib.EDGE_ATTENTION for d in desc_tuples)
        self.any_natt = any(d[2] & _lib.NODE_ATTENTION for d in desc_tuples)
        self.index, flat = [], []
        for layer, params in enumerate(param_tuples):
            for slot, p in enumerate(params):
                if p is not None:
                    self.index.append((layer, slot))
                    flat.append(p)
        self.params = tuple(flat)
        self._sizes, self._layouts = {}, {}

    def grad_layout(self, last_coords_live):
        """Where every parameter gradient of a backward sits in ONE flat buffer. Which gradients exist follows
        _EGNNLayerFn.backward: a layer's coord_mlp only when its coordinates fed something (every layer but the last; the
        last one when a gradient arrives for x_L), never the edge gate (no edge residual in a stack)."""
        got = self._layouts.get(last_coords_live)
        if got is None:
            got = self._layouts[last_coords_live] = _GradLayout(self, last_coords_live)
        return got

    def sizes(self, n, e):
        """(strides struct, forward workspace bytes, backward workspace bytes) for a graph of n nodes and e edges."""
        got = self._sizes.get((n, e))
        if got is None:
            lib = _lib.lib()
            saved = lib.pvs_egnn_layer_saved_floats(self.descs, n, e)
            st = _lib.PvsStackStrides(_align(n * self.hidden), _align(3 * n), _align(max(e, 1)), _align(n), _align(saved))
            got = (st, lib.pvs_egnn_stack_workspace_bytes(self.descs, self.n_layers, n, e, 0),
                   lib.pvs_egnn_stack_workspace_bytes(self.descs, self.n_layers, n, e, 1))
            if len(self._sizes) > 64:
                self._sizes.clear()
            self._sizes[(n, e)] = got
        return got


class _GradLayout:
    """`sizes`: the split of the flat buffer (gradients and the pads that keep each one 16-byte aligned); `take`: per
    parameter of `plan.params` (index into the split or -1 = no gradient, shape to view it as or None for 1-D);
    `structs(base)`: the PvsLayerGrads array for a flat buffer at address `base` (kept per address: the caching allocator
    alternates between a few)."""

    def __init__(self, plan, last_coords_live):
        fields, nl = _lib.PARAM_FIELDS, plan.n_layers
        self.sizes, self.take, self._offsets, self._by_base = [], [], [], {}
        off = 0
        for (layer, slot), p in zip(plan.index, plan.params):
            name = fields[slot]
            on = True
            if name in ('coord_w1', 'coord_b1', 'coord_w2'):
                on = bool(plan.desc_tuples[layer][2] & _lib.UPDATE_COORDS) and (layer < nl - 1 or last_coords_live)
            elif name == 'edge_gate':
                on = False
            if not on:
                self.take.append((-1, None))
                continue
            k = p.numel()
            self.take.append((len(self.sizes), None if p.dim() == 1 else tuple(p.shape)))
            self.sizes.append(k)
            self._offsets.append((layer, slot, 4 * off))
            off += k
            if k % 4:
                self.sizes.append(4 - k % 4)
                off += 4 - k % 4
        self.total, self.n_layers, self.n_fields = off, nl, len(fields)

    def structs(self, base):
        got = self._by_base.get(base)
        if got is None:
            rows = [[None] * self.n_fields for _ in range(self.n_layers)]
            for layer, slot, byte_off in self._offsets:
                rows[layer][slot] = base + byte_off
            got = (_lib.PvsLayerGrads * self.n_layers)(*[_lib.PvsLayerGrads(*r) for r in rows])
            if len(self._by_base) > 32:
                self._by_base.clear()
            self._by_base[base] = got
        return got


class _EGNNStackFn(torch.autograd.Function):
    """All EGNN layers of a model (the loop of SartorrasEGNN.get_embeddings, egnn_satorras.py:325-328) as ONE autograd
    node over `pvs_egnn_stack_fwd/bwd`: the same launches as one `_EGNNLayerFn` per layer, bit for bit the same values
    (tests/test_gpu_stack.py), a sixth of the host time for six layers. No edge residual (per-layer path)."""

    @staticmethod
    def forward(ctx, h0, x0, pg, plan, *params):
        lib = _lib.lib()
        h0, x0 = _f32c(h0), _f32c(x0)
        _lib.require_hip(h0, x0)
        n, e, hidden, nl = pg.n_nodes, pg.n_edges, plan.hidden, plan.n_layers
        if h0.shape != (n, hidden) or x0.shape != (n, 3):
            raise ValueError(f'h {tuple(h0.shape)} / coord {tuple(x0.shape)} do not match N={n}, H={hidden}')
        if plan.desc_tuples[0][1] != pg.n_edge_attr:
            raise ValueError(f'layers built with edges_in_d={plan.desc_tuples[0][1]} but edge_attr has '
                             f'{pg.n_edge_attr} columns')
        dev = h0.device
        st, ws_bytes, _ = plan.sizes(n, e)
        f32 = torch.float32
        h_out, x_out = torch.empty_like(h0), torch.empty_like(x0)
        h_mid = torch.empty((nl - 1, st.h_mid), dtype=f32, device=dev) if nl > 1 else None
        x_mid = torch.empty((nl - 1, st.x_mid), dtype=f32, device=dev) if nl > 1 else None
        att = torch.empty((nl, st.att), dtype=f32, device=dev) if plan.any_eatt else None
        natt = torch.empty((nl, st.node_att), dtype=f32, device=dev) if plan.any_natt else None
        saved = torch.empty((nl, st.saved), dtype=f32, device=dev)
        ws = _ws(ws_bytes, dev)
        rc = lib.pvs_egnn_stack_fwd(plan.descs, plan.params_c, nl, C.byref(pg.c), C.byref(st), _lib.ptr(h0), _lib.ptr(x0),
                                    _lib.ptr(h_mid), _lib.ptr(x_mid), _lib.ptr(h_out), _lib.ptr(x_out), _lib.ptr(att),
                                    _lib.ptr(natt), _lib.ptr(saved), _lib.ptr(ws), ws_bytes, _stream(dev))
        _lib.check(rc, 'pvs_egnn_stack_fwd')
        ctx.pg, ctx.plan = pg, plan
        ctx.save_for_backward(h0, x0, h_mid, x_mid, att, saved, *params)
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(*[t for t in (h_mid, x_mid, att, natt) if t is not None])
        return h_out, x_out, h_mid, x_mid, att, natt

    @staticmethod
    def backward(ctx, g_h_out, g_x_out, *_unused):
        lib = _lib.lib()
        h0, x0, h_mid, x_mid, att, saved, *params = ctx.saved_tensors
        pg, plan = ctx.pg, ctx.plan
        n, e, nl = pg.n_nodes, pg.n_edges, plan.n_layers
        dev = h0.device
        g_h_out = torch.zeros_like(h0) if g_h_out is None else _f32c(g_h_out)
        g_x_out = _f32c(g_x_out)        # None => nothing read the last layer's coordinates
        g_h0 = torch.empty_like(h0)
        g_x0 = torch.empty_like(x0) if ctx.needs_input_grad[1] else None
        # ONE allocation for all parameter gradients, handed out as views (16-byte aligned): 78 allocations less per step
        # of a 6-layer model, and the optimiser finds its pointer table again by one address instead of 78 that the
        # caching allocator shuffles from step to step (FusedClipAdam._recent)
        lay = plan.grad_layout(g_x_out is not None)
        flat = torch.empty((lay.total,), dtype=torch.float32, device=dev)
        parts = flat.split_with_sizes(lay.sizes)
        grads = [None if k < 0 else (parts[k] if shape is None else parts[k].view(shape)) for k, shape in lay.take]
        gstructs = lay.structs(flat.data_ptr())
        st, _, ws_bytes = plan.sizes(n, e)
        ws = _ws(ws_bytes, dev)
        rc = lib.pvs_egnn_stack_bwd(plan.descs, plan.params_c, nl, C.byref(pg.c), C.byref(st), _lib.ptr(h0), _lib.ptr(x0),
                                    _lib.ptr(h_mid), _lib.ptr(x_mid), _lib.ptr(att), _lib.ptr(saved), _lib.ptr(g_h_out),
                                    _lib.ptr(g_x_out), _lib.ptr(g_h0), _lib.ptr(g_x0), gstructs, _lib.ptr(ws), ws_bytes,
                                    _stream(dev))
        _lib.check(rc, 'pvs_egnn_stack_bwd')
        return (g_h0, g_x0, None, None, *grads)


def egnn_stack(h0, x0, pg, plan):
    """Returns (h_L, x_L, h_mid, x_mid, att, node_att): the last layer's outputs and the per-layer buffers (rows of
    `plan.sizes(n, e)[0]` strides; None where no layer needs them) the side attributes are read from."""
    return _EGNNStackFn.apply(h0, x0, pg, plan, *plan.params)


class _LinearFn(torch.autograd.Function):
    """y = x W^T + b  (PygLinearPass / head Linear; pnn_geometric_base.py:83-94)."""

    @staticmethod
    def forward(ctx, x, w, b):
        x, w, b = _f32c(x), _f32c(w), _f32c(b)
        _lib.require_hip(x, w, b)
        n, k = x.shape
        c = w.shape[0]
        y = torch.empty((n, c), dtype=torch.float32, device=x.device)
        _lib.check(_lib.lib().pvs_linear_fwd(_lib.ptr(x), _lib.ptr(w), _lib.ptr(b), _lib.ptr(y),
                                             n, k, c, _stream(x.device)), 'pvs_linear_fwd')
        ctx.save_for_backward(x, w)
        ctx.has_bias = b is not None
        return y

    @staticmethod
    def backward(ctx, g_y):
        x, w = ctx.saved_tensors
        g_y = _f32c(g_y)
        n, k = x.shape
        c = w.shape[0]
        lib = _lib.lib()
        g_x = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        g_w = torch.empty_like(w)
        g_b = torch.empty((c,), dtype=torch.float32, device=x.device) if ctx.has_bias else None
        ws_bytes = lib.pvs_linear_bwd_workspace_bytes(n, k, c)
        ws = _ws(ws_bytes, x.device)
        _lib.check(lib.pvs_linear_bwd(_lib.ptr(x), _lib.ptr(w), _lib.ptr(g_y), _lib.ptr(g_x),
                                      _lib.ptr(g_w), _lib.ptr(g_b), n, k, c, _lib.ptr(ws), ws_bytes,
                                      _stream(x.device)), 'pvs_linear_bwd')
        return g_x, g_w, g_b


def linear(x, weight, bias=None):
    squeeze = x.dim() == 1
    y = _LinearFn.apply(x.reshape(1, -1) if squeeze else x, weight, bias)
    return y.reshape(-1) if squeeze else y


class _MeanPoolFn(torch.autograd.Function):
    """global_mean_pool over contiguous per-graph node ranges (pnn_geometric_base.py:29-33)."""

    @staticmethod
    def forward(ctx, h, graph_ptr):
        h = _f32c(h)
        _lib.require_hip(h, graph_ptr)
        b = graph_ptr.numel() - 1
        pooled = torch.empty((b, h.shape[1]), dtype=torch.float32, device=h.device)
        _lib.check(_lib.lib().pvs_mean_pool_fwd(_lib.ptr(h), _lib.ptr(graph_ptr), _lib.ptr(pooled),
                                                b, h.shape[1], _stream(h.device)),
                   'pvs_mean_pool_fwd')
        ctx.graph_ptr, ctx.n = graph_ptr, h.shape[0]
        return pooled

    @staticmethod
    def backward(ctx, g):
        g = _f32c(g)
        b, width = g.shape
        g_h = torch.empty((ctx.n, width), dtype=torch.float32, device=g.device)
        _lib.check(_lib.lib().pvs_mean_pool_bwd(_lib.ptr(g), _lib.ptr(ctx.graph_ptr), _lib.ptr(g_h),
                                                b, ctx.n, width, _stream(g.device)),
                   'pvs_mean_pool_bwd')
        return g_h, None


def mean_pool(h, graph_ptr):
    return _MeanPoolFn.apply(h, graph_ptr)


class _PoolHeadFn(torch.autograd.Function):
    """global_mean_pool + the head's first Linear in one launch, their backward in one launch
    (pnn_geometric_base.py:29-36, egnn_multitask.py:158-166)."""

    @staticmethod
    def forward(ctx, h, graph_ptr, w, b):
        h, w, b = _f32c(h), _f32c(w), _f32c(b)
        _lib.require_hip(h, graph_ptr, w, b)
        n_graphs, width, n_out = graph_ptr.numel() - 1, h.shape[1], w.shape[0]
        pooled = torch.empty((n_graphs, width), dtype=torch.float32, device=h.device)
        y = torch.empty((n_graphs, n_out), dtype=torch.float32, device=h.device)
        _lib.check(_lib.lib().pvs_pool_head_fwd(_lib.ptr(h), _lib.ptr(graph_ptr), _lib.ptr(w), _lib.ptr(b),
                                                _lib.ptr(pooled), _lib.ptr(y), n_graphs, width, n_out,
                                                _stream(h.device)), 'pvs_pool_head_fwd')
        ctx.save_for_backward(pooled, w)
        ctx.graph_ptr, ctx.n, ctx.has_bias = graph_ptr, h.shape[0], b is not None
        return y

    @staticmethod
    def backward(ctx, g_y):
        pooled, w = ctx.saved_tensors
        g_y = _f32c(g_y)
        n_graphs, width = pooled.shape
        n_out = w.shape[0]
        dev = g_y.device
        g_h = torch.empty((ctx.n, width), dtype=torch.float32, device=dev) if ctx.needs_input_grad[0] else None
        g_w = torch.empty_like(w)
        g_b = torch.empty((n_out,), dtype=torch.float32, device=dev) if ctx.has_bias else None
        _lib.check(_lib.lib().pvs_pool_head_bwd(_lib.ptr(g_y), _lib.ptr(pooled), _lib.ptr(w), _lib.ptr(ctx.graph_ptr),
                                                _lib.ptr(g_h), _lib.ptr(g_w), _lib.ptr(g_b), n_graphs, ctx.n, width,
                                                n_out, _stream(dev)), 'pvs_pool_head_bwd')
        return g_h, None, g_w, g_b


POOL_HEAD_MAX_WIDTH = 1024


def pool_head(h, graph_ptr, weight, bias=None):
    """linear(mean_pool(h, graph_ptr), weight, bias) as one op (hidden widths up to POOL_HEAD_MAX_WIDTH)."""
    return _PoolHeadFn.apply(h, graph_ptr, weight, bias)


class _BceLogitsMeanFn(torch.autograd.Function):
    """nn.BCEWithLogitsLoss() (mean) as one launch forward (loss + the gradient factor), one backward."""

    @staticmethod
    def forward(ctx, x, target):
        shape = x.shape
        x, target = _f32c(x).reshape(-1), _f32c(target).reshape(-1)
        _lib.require_hip(x, target)
        n = x.numel()
        loss = torch.empty((), dtype=torch.float32, device=x.device)
        grad = torch.empty_like(x)
        _lib.check(_lib.lib().pvs_bce_logits_fwd(_lib.ptr(x), _lib.ptr(target), n, _lib.ptr(loss), _lib.ptr(grad),
                                                 _stream(x.device)), 'pvs_bce_logits_fwd')
        ctx.save_for_backward(grad)
        ctx.shape = shape
        return loss

    @staticmethod
    def backward(ctx, g_loss):
        grad, = ctx.saved_tensors
        if g_loss.data_ptr() == _unit_gradient_ptr(g_loss.device):
            # `loss.backward(gradient=unit_gradient(device))` (the harness's backprop): the upstream gradient is the
            # constant 1, so the factor saved by the forward IS the gradient - no scaling launch (and autograd did not
            # have to fill a ones tensor either)
            return grad.reshape(ctx.shape), None
        g_loss = _f32c(g_loss)
        out = torch.empty_like(grad)
        _lib.check(_lib.lib().pvs_scale_by_device_scalar(_lib.ptr(grad), _lib.ptr(g_loss), grad.numel(), _lib.ptr(out),
                                                         _stream(grad.device)), 'pvs_scale_by_device_scalar')
        return out.reshape(ctx.shape), None


_UNIT_GRADIENTS = {}


def unit_gradient(device):
    """A scalar 1.0 on `device`, made once: `loss.backward(gradient=unit_gradient(loss.device))` is `loss.backward()`
    without the ones_like fill autograd launches for the root gradient, and lets the fused loss skip its own scaling
    launch (two launches of ~4 us per training step). Never written to."""
    device = torch.device(device)
    if device.type == 'cuda' and device.index is None:
        device = torch.device('cuda', torch.cuda.current_device())
    unit = _UNIT_GRADIENTS.get(device)
    if unit is None:
        unit = _UNIT_GRADIENTS[device] = torch.ones((), dtype=torch.float32, device=device)
    return unit


def _unit_gradient_ptr(device):
    unit = _UNIT_GRADIENTS.get(device)
    return -1 if unit is None else unit.data_ptr()


def bce_with_logits_mean(y_pred, y_true):
    if y_pred.shape != y_true.shape:
        raise ValueError(f'Target size ({tuple(y_true.shape)}) must be the same as input size ({tuple(y_pred.shape)})')
    if y_pred.numel() == 0:
        raise ValueError('bce_with_logits_mean: empty input')
    return _BceLogitsMeanFn.apply(y_pred, y_true)


class _SegmentStatus:
    """The status word of one pvs_segment_reduce_fwd call, read WITHOUT blocking the stream (as
    graph.PreparedGraph.poll_status reads the graph preparation's): the call queues a copy into pinned memory and an
    event; every later segment_reduce call raises for whichever earlier words have landed, and the call's own backward
    waits for its word (it has long landed by then). The reference's scatter_add_ raises at once
    (egnn_satorras.py:336); here the error surfaces one call late - or in the backward - but it does surface."""
    pending = []

    def __init__(self, status):
        self.host = torch.empty(1, dtype=torch.int32, pin_memory=True)
        self.host.copy_(status, non_blocking=True)
        self.event = torch.cuda.Event()
        self.event.record(torch.cuda.current_stream(status.device))
        self.done = False
        _SegmentStatus.pending.append(self)

    def settle(self, wait):
        if self.done:
            return
        if not self.event.query():
            if not wait:
                return
            self.event.synchronize()
        self.done = True
        if self in _SegmentStatus.pending:
            _SegmentStatus.pending.remove(self)
        if int(self.host.item()) & 1:
            raise IndexError('unsorted_segment_sum / unsorted_segment_mean: segment_ids contains values outside '
                             '[0, num_segments)')

    @classmethod
    def poll(cls):
        for st in list(cls.pending):
            st.settle(wait=False)


class _SegmentReduceFn(torch.autograd.Function):
    """unsorted_segment_sum / unsorted_segment_mean (egnn_satorras.py:332-347)."""

    @staticmethod
    def forward(ctx, data, segment_ids, num_segments, mean):
        data = _f32c(data)
        _lib.require_hip(data, segment_ids)
        lib = _lib.lib()
        ids = segment_ids.long().contiguous()
        e, c = data.shape
        dev = data.device
        capturing = torch.cuda.is_current_stream_capturing()
        if not capturing:
            _SegmentStatus.poll()           # an earlier call's out-of-range ids raise here
        out = torch.empty((num_segments, c), dtype=torch.float32, device=dev)
        ptr = torch.empty(num_segments + 1, dtype=torch.int32, device=dev)
        status = torch.empty(1, dtype=torch.int32, device=dev)
        ws_bytes = lib.pvs_segment_workspace_bytes(e, num_segments)
        ws = _ws(ws_bytes, dev)
        _lib.check(lib.pvs_segment_reduce_fwd(
            _lib.ptr(data), _lib.ptr(ids), e, c, num_segments, 1 if mean else 0, _lib.ptr(out),
            _lib.ptr(ptr), _lib.ptr(status), _lib.ptr(ws), ws_bytes, _stream(dev)),
            'pvs_segment_reduce_fwd')
        ctx.status = None if capturing else _SegmentStatus(status)      # (no host-visible validation inside a capture)
        ctx.save_for_backward(ids, ptr)
        ctx.mean, ctx.shape, ctx.n_segments = mean, (e, c), num_segments
        return out

    @staticmethod
    def backward(ctx, g_out):
        ids, ptr = ctx.saved_tensors
        if ctx.status is not None and not torch.cuda.is_current_stream_capturing():
            ctx.status.settle(wait=True)
        g_out = _f32c(g_out)
        e, c = ctx.shape
        g_data = torch.empty((e, c), dtype=torch.float32, device=g_out.device)
        _lib.check(_lib.lib().pvs_segment_reduce_bwd(
            _lib.ptr(g_out), _lib.ptr(ids), _lib.ptr(ptr), e, c, ctx.n_segments, 1 if ctx.mean else 0,
            _lib.ptr(g_data), _stream(g_out.device)), 'pvs_segment_reduce_bwd')
        return g_data, None, None, None


def segment_reduce(data, segment_ids, num_segments, mean=False):
    return _SegmentReduceFn.apply(data, segment_ids, int(num_segments), bool(mean))


def segment_status_check():
    """Wait for every pending status word of segment_reduce and raise for the first bad one (call where a sync is fine)."""
    for st in list(_SegmentStatus.pending):
        st.settle(wait=True)


def dropout_adj(edge_index, edge_attr=None, p=0.5, force_undirected=True, training=True, seed=0, step=0):
    """torch_geometric's dropout_adj as the reference calls it (egnn_satorras.py:320-323: force_undirected=True):
    of every pair only the row <= col copy is drawn, survivors first, their reverses behind, attributes
    repeated. Identity when not training or p == 0. The draw is the library's own counter-based stream
    (pvs_dropout_adj_mark: Philox keyed on (seed, step)), reproducible but not torch's. One host read (the
    number of survivors sizes the outputs), like every data-dependent filter."""
    if not training or not p:
        return edge_index, edge_attr
    if not force_undirected:
        raise NotImplementedError('the reference only calls dropout_adj with force_undirected=True')
    _lib.require_hip(edge_index, edge_attr)
    lib = _lib.lib()
    edge_index = edge_index.long().contiguous()
    n_edges = int(edge_index.shape[1])
    if n_edges == 0:          # nothing to draw from (an empty tensor has no device pointer to hand over)
        return edge_index, edge_attr
    dev = edge_index.device
    n_attr = 0
    if edge_attr is not None:
        edge_attr = edge_attr.long().contiguous()
        n_attr = int(edge_attr.shape[1])
    stream = _lib.stream(dev)
    pos = torch.empty(n_edges + 1, dtype=torch.int32, device=dev)
    ws_bytes = lib.pvs_dropout_adj_workspace_bytes(n_edges)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    _lib.check(lib.pvs_dropout_adj_mark(_lib.ptr(edge_index), n_edges, float(p), int(seed) & (2 ** 64 - 1),
                                        int(step), _lib.ptr(pos), _lib.ptr(ws), ws_bytes, stream),
               'pvs_dropout_adj_mark')
    kept = int(pos[-1])
    out_index = torch.empty((2, 2 * kept), dtype=torch.int64, device=dev)
    out_attr = None if edge_attr is None else torch.empty((2 * kept, n_attr), dtype=torch.int64, device=dev)
    _lib.check(lib.pvs_dropout_adj_fill(_lib.ptr(edge_index), _lib.ptr(edge_attr), n_attr, n_edges, _lib.ptr(pos),
                                        kept, _lib.ptr(out_index), _lib.ptr(out_attr), stream),
               'pvs_dropout_adj_fill')
    return out_index, out_attr
