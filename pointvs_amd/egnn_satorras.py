"""EGNN layer and network with the reference's class surface, bodies in HIP.

Mirrors /root/reference/point_vs/models/geometric/egnn_satorras.py:
  EGNNLayer            :23-206   (constructor signature, forward signature, state_dict keys,
                                  side attributes att_val / node_att_val / intermediate_coords)
  SartorrasEGNN        :209-329  (build_net kwargs, .layers Sequential, get_embeddings)
The nn.Sequential members only hold parameters under the reference's names (so reference
checkpoints load and seeds give the same init); they are never called. The arithmetic of
forward/backward is `pvs_egnn_layer_fwd/bwd` in libpvs_egnn.so (include/pvs_egnn.h).
"""
import os

import torch
from torch import nn

from . import _lib
from . import functional as PF
from .graph import prepared_for
from .pnn_geometric_base import PNNGeometricBase, PygLinearPass


class GraphNorm(nn.Module):
    """Parameter holder for torch_geometric.nn.norm.GraphNorm (weight, bias, mean_scale).

    The reference places it inside an nn.Sequential, i.e. calls it without a batch vector, so the
    statistics run over all nodes of the mini-batch; that is what the node
    kernels implement (eps = 1e-5).
    """

    def __init__(self, in_channels, eps=1e-5):
        super().__init__()
        self.in_channels, self.eps = in_channels, eps
        self.weight = nn.Parameter(torch.ones(in_channels))
        self.bias = nn.Parameter(torch.zeros(in_channels))
        self.mean_scale = nn.Parameter(torch.ones(in_channels))


def _lazy_numpy(value):
    if value is None:
        return None
    return value().detach().cpu().numpy()


class EGNNLayer(nn.Module):
    """E(n)-equivariant message-passing layer (egnn_satorras.py:23-206), HIP-backed."""

    def __init__(self, input_nf, output_nf, hidden_nf, edges_in_d=0, act_fn=nn.SiLU(),
                 residual=True, edge_residual=False, edge_attention=False, normalize=False,
                 tanh=False, graphnorm=False, update_coords=True, permutation_invariance=False,
                 node_attention=False, attention_activation_fn='sigmoid', gated_residual=False,
                 rezero=False, softmax_attention=False):
        assert not (gated_residual and rezero), 'gated_residual and rezero are incompatible'
        super().__init__()
        if not (input_nf == output_nf == hidden_nf):
            raise NotImplementedError(
                'HIP EGNNLayer needs input_nf == output_nf == hidden_nf (build_net always passes '
                'k, k, k: egnn_satorras.py:283)')
        if not isinstance(act_fn, nn.SiLU):
            raise NotImplementedError('HIP EGNNLayer implements SiLU only (the reference never '
                                      'forwards another act_fn: SURVEY.md §5)')
        if attention_activation_fn not in ('sigmoid', 'tanh', 'relu', 'silu'):
            raise KeyError(attention_activation_fn)
        input_edge = input_nf if permutation_invariance else input_nf * 2
        self.gated_residual, self.rezero = gated_residual, rezero
        self.residual, self.edge_residual = residual, edge_residual
        self.edge_attention, self.normalize, self.tanh = edge_attention, normalize, tanh
        self.epsilon = 1e-8
        self.use_coords = update_coords
        self.permutation_invariance = permutation_invariance
        self.node_attention = node_attention
        self.hidden_nf = hidden_nf
        self.edges_in_d = edges_in_d
        self.graphnorm = graphnorm
        self.softmax_attention = softmax_attention
        self.attention_activation_name = 'identity' if softmax_attention else attention_activation_fn
        self.attention_activation = {
            'sigmoid': nn.Sigmoid, 'tanh': nn.Tanh, 'relu': nn.ReLU, 'silu': nn.SiLU,
            'identity': nn.Identity}[self.attention_activation_name]
        self._att_src = self._natt_src = self._coords_src = None

        # construction order follows the reference so a given torch seed yields the same init
        self.edge_mlp = nn.Sequential(
            nn.Linear(input_edge + 1 + edges_in_d, hidden_nf), act_fn,
            nn.Linear(hidden_nf, hidden_nf), act_fn)
        self.node_mlp = nn.Sequential(
            nn.Linear(hidden_nf + input_nf, hidden_nf),
            GraphNorm(hidden_nf) if graphnorm else nn.Identity(), act_fn,
            nn.Linear(hidden_nf, output_nf))
        last = nn.Linear(hidden_nf, 1, bias=False)
        torch.nn.init.xavier_uniform_(last.weight, gain=0.001)
        self.coord_mlp = nn.Sequential(
            nn.Linear(hidden_nf, hidden_nf), act_fn, last, nn.Tanh() if tanh else nn.Identity())
        if edge_attention:
            self.att_mlp = nn.Sequential(nn.Linear(hidden_nf, 1), self.attention_activation())
        if node_attention:
            self.node_att_mlp = nn.Sequential(nn.Linear(hidden_nf, 1), self.attention_activation())
        if rezero:
            if edge_residual:
                self.edge_gate_parameter = nn.Parameter(torch.zeros(1))
            if residual:
                self.node_gate_parameter = nn.Parameter(torch.zeros(1))
        elif gated_residual:
            if edge_residual:
                self.edge_gate_parameter = nn.Parameter(0.5 * torch.ones(1))
            if residual:
                self.node_gate_parameter = nn.Parameter(0.5 * torch.ones(1))

    # ---- side attributes: numpy on access, no device sync in forward ----
    @property
    def att_val(self):
        return _lazy_numpy(self._att_src)

    @att_val.setter
    def att_val(self, value):
        self._att_src = None if value is None else (lambda: torch.as_tensor(value))

    @property
    def node_att_val(self):
        return _lazy_numpy(self._natt_src)

    @node_att_val.setter
    def node_att_val(self, value):
        self._natt_src = None if value is None else (lambda: torch.as_tensor(value))

    @property
    def intermediate_coords(self):
        return _lazy_numpy(self._coords_src)

    @intermediate_coords.setter
    def intermediate_coords(self, value):
        self._coords_src = None if value is None else (lambda: torch.as_tensor(value))

    # ---- C ABI plumbing ----
    def _flags(self):
        f = 0
        for on, bit in ((self.residual, _lib.RESIDUAL), (self.edge_residual, _lib.EDGE_RESIDUAL),
                        (self.edge_attention, _lib.EDGE_ATTENTION), (self.normalize, _lib.NORMALIZE),
                        (self.tanh, _lib.TANH), (self.graphnorm, _lib.GRAPHNORM),
                        (self.use_coords, _lib.UPDATE_COORDS),
                        (self.permutation_invariance, _lib.PERM_INVARIANT),
                        (self.node_attention, _lib.NODE_ATTENTION),
                        (self.gated_residual, _lib.GATED_RESIDUAL), (self.rezero, _lib.REZERO),
                        (self.softmax_attention, _lib.SOFTMAX_ATT)):
            if on:
                f |= bit
        return f

    def _desc(self):
        return (self.hidden_nf, self.edges_in_d, self._flags(),
                _lib.ACT_CODES[self.attention_activation_name])

    def _params(self):
        gn = self.node_mlp[1] if self.graphnorm else None
        att = self.att_mlp[0] if self.edge_attention else None
        natt = self.node_att_mlp[0] if self.node_attention else None
        return (
            self.edge_mlp[0].weight, self.edge_mlp[0].bias, self.edge_mlp[2].weight,
            self.edge_mlp[2].bias, self.coord_mlp[0].weight, self.coord_mlp[0].bias,
            self.coord_mlp[2].weight,
            None if att is None else att.weight, None if att is None else att.bias,
            self.node_mlp[0].weight, self.node_mlp[0].bias, self.node_mlp[3].weight,
            self.node_mlp[3].bias,
            None if gn is None else gn.weight, None if gn is None else gn.bias,
            None if gn is None else gn.mean_scale,
            None if natt is None else natt.weight, None if natt is None else natt.bias,
            getattr(self, 'edge_gate_parameter', None), getattr(self, 'node_gate_parameter', None))

    # (container attribute, index inside it or None, parameter name) of the twenty slots of `_params()`, in its order
    _SLOTS = (('edge_mlp', '0', 'weight'), ('edge_mlp', '0', 'bias'), ('edge_mlp', '2', 'weight'),
              ('edge_mlp', '2', 'bias'), ('coord_mlp', '0', 'weight'), ('coord_mlp', '0', 'bias'),
              ('coord_mlp', '2', 'weight'), ('att_mlp', '0', 'weight'), ('att_mlp', '0', 'bias'),
              ('node_mlp', '0', 'weight'), ('node_mlp', '0', 'bias'), ('node_mlp', '3', 'weight'),
              ('node_mlp', '3', 'bias'), ('node_mlp', '1', 'weight'), ('node_mlp', '1', 'bias'),
              ('node_mlp', '1', 'mean_scale'), ('node_att_mlp', '0', 'weight'), ('node_att_mlp', '0', 'bias'),
              (None, None, 'edge_gate_parameter'), (None, None, 'node_gate_parameter'))

    def _slot_probes(self, params):
        """For every non-None slot of `params` the chain of plain dicts that leads from the layer to the tensor:
        (layer._modules, container name, container._modules, index, leaf._parameters, parameter name, tensor). Walking
        it is seven dict look-ups per slot and sees a replaced container, a replaced leaf module and a replaced
        parameter alike. A slot whose tensor is not a registered parameter (parametrize / weight_norm compute it on
        access) has no chain: returns None and nothing is cached."""
        probes = []
        for (cont, idx, name), p in zip(self._SLOTS, params):
            if p is None:
                continue
            if cont is None:
                if self._parameters.get(name) is not p:
                    return None
                probes.append((None, None, None, None, self._parameters, name, p))
                continue
            seq = self._modules.get(cont)
            leaf = None if seq is None else seq._modules.get(idx)
            if leaf is None or leaf._parameters.get(name) is not p:
                return None
            probes.append((self._modules, cont, seq, idx, leaf, name, p))
        return tuple(probes)

    def _params_cached(self):
        """(parameter tuple, PvsLayerParams) built once per layer: 35 module look-ups, 20 contiguity checks and a ctypes
        struct per call otherwise (small batches are host-bound: tools/host_profile.py). The parameters are updated in
        place (Adam, load_state_dict), so tensors and addresses stay. EVERY slot is validated on every call, by
        identity against the live modules (`_slot_probes`: plain dict look-ups, ~2 us per layer) and by address
        (`p.data = ...`): assigning a nested parameter (`layer.node_mlp[0].weight = ...`,
        `load_state_dict(assign=True)`), replacing a leaf module or a container all rebuild the struct. The cache is
        also dropped by `_apply` (.to / .cuda / .float) and is never pickled or deep-copied (`__getstate__`: a ctypes
        struct with pointers cannot be)."""
        cached = self.__dict__.get('_pcache')
        if cached is not None:
            params, pstruct, probes = cached
            for mods, cont, seq, idx, leaf, name, p, addr in probes:
                if mods is None:
                    if leaf.get(name) is not p or p.data_ptr() != addr:
                        break
                elif (mods.get(cont) is not seq or seq._modules.get(idx) is not leaf
                        or leaf._parameters.get(name) is not p or p.data_ptr() != addr):
                    break
            else:
                return params, pstruct
        params = self._params()
        ok = all(p is None or (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()) for p in params)
        if not ok:
            self.__dict__.pop('_pcache', None)
            return params, None
        pstruct = _lib.PvsLayerParams(*[_lib.ptr(p) for p in params])
        probes = self._slot_probes(params)
        if probes is None:            # computed weights (parametrizations): a fresh struct per call
            self.__dict__.pop('_pcache', None)
        else:
            self.__dict__['_pcache'] = (params, pstruct, tuple(pr + (pr[6].data_ptr(),) for pr in probes))
        return params, pstruct

    def _apply(self, fn, *args, **kwargs):
        self.__dict__.pop('_pcache', None)
        return super()._apply(fn, *args, **kwargs)

    def __setattr__(self, name, value):
        if isinstance(value, (torch.nn.Parameter, torch.nn.Module)):
            self.__dict__.pop('_pcache', None)
        super().__setattr__(name, value)

    def __getstate__(self):
        """copy.deepcopy (EMA / SWA `AveragedModel`, snapshots) and pickling: without the ctypes struct cache (pointers
        cannot be pickled, and a copy has its own tensors) and without the lazily evaluated side attributes (closures
        over device tensors of the last call; `att_val` etc. read None on the copy until its own first forward)."""
        state = self.__dict__.copy()
        state.pop('_pcache', None)
        for k in ('_att_src', '_natt_src', '_coords_src'):
            if k in state:
                state[k] = None
        return state

    _KERNEL_WIDTHS = (16, 32, 64, 128)

    def _fused_width_ok(self):
        """Hidden sizes 65..128 run on the fused 128-channel kernels (zero-padded: 65..127) - two launches of the
        f16x2 edge forward and the four-wave team backward, built for the MFMA path with up to 3 edge classes.
        Wider layers, PVS_WIDE=decomposed, the generic kernel family or more edge classes take the composition of
        the public sub-methods instead."""
        if self.hidden_nf <= 64:
            return True
        return (self.hidden_nf <= 128 and self.edges_in_d <= 3 and os.environ.get('PVS_WIDE') != 'decomposed'
                and os.environ.get('PVS_EGNN_KERNELS', '')[:1] != 'g')

    def _padded_call(self, pg, h, coord, m_prev_sorted, need_m, flags=None):
        """Hidden sizes the kernels are not built for run zero-padded to the next built width.
        Exact: a zero channel stays zero through SiLU, the products, GraphNorm (weight 0, bias 0)
        and the gates, so the extra channels contribute nothing; padding/slicing are autograd ops."""
        import torch.nn.functional as F
        hid = self.hidden_nf
        wide = next(w for w in self._KERNEL_WIDTHS if w >= hid)
        pad = wide - hid
        a = self.edges_in_d

        def rows(t):   # pad the output-channel dimension
            return None if t is None else F.pad(t, (0, 0, 0, pad)) if t.dim() == 2 else F.pad(t, (0, pad))

        def cols(t):   # pad the input-channel dimension
            return None if t is None else F.pad(t, (0, pad))

        p = list(self._params())
        w1 = p[0]
        blocks = [w1[:, :hid]] + ([] if self.permutation_invariance else [w1[:, hid:2 * hid]])
        tail = w1[:, (hid if self.permutation_invariance else 2 * hid):]
        w1p = torch.cat([cols(b) for b in blocks] + [tail], dim=1)
        wn1 = p[9]
        wn1p = torch.cat([cols(wn1[:, :hid]), cols(wn1[:, hid:])], dim=1)
        padded = [rows(w1p), rows(p[1]), rows(cols(p[2])), rows(p[3]), rows(cols(p[4])), rows(p[5]),
                  cols(p[6]), cols(p[7]), p[8], rows(wn1p), rows(p[10]), rows(cols(p[11])), rows(p[12]),
                  rows(p[13]), rows(p[14]), rows(p[15]), cols(p[16]), p[17], p[18], p[19]]
        desc = (wide, a) + self._desc()[2:]
        if flags is not None:
            desc = (wide, a, flags, desc[3])
        mp = None if m_prev_sorted is None else F.pad(m_prev_sorted, (0, pad))
        h_out, x_out, m_sorted, att, natt = PF.egnn_layer(
            F.pad(h, (0, pad)), coord, mp, pg, desc, need_m, tuple(padded))
        return (h_out[:, :hid], x_out, None if m_sorted is None else m_sorted[:, :hid], att, natt)

    def forward_prepared(self, pg, h, coord, m_prev_sorted=None, need_m=False, need_coords=True):
        """Layer on a PreparedGraph; edge tensors stay in CSR-sorted order (internal fast path).
        need_coords=False: the caller discards the returned coordinates (the last layer of a model:
        `x_L` feeds nothing), so the coordinate branch of the edge kernel is skipped and
        the input coordinates are returned; `intermediate_coords` is then evaluated on demand."""
        if not self.edge_residual:
            m_prev_sorted = None
        # PVS_EGNN_KEEP_DEAD_COORDS=1 evaluates the unused coordinate update anyway (like the reference)
        skip_coords = self.use_coords and not need_coords and not os.environ.get('PVS_EGNN_KEEP_DEAD_COORDS')
        desc = self._desc()
        if skip_coords:
            desc = (desc[0], desc[1], desc[2] & ~_lib.UPDATE_COORDS, desc[3])
        if not self._fused_width_ok():
            # wider than the fused kernels are built for: the layer as the composition of its public
            # sub-methods (dense products and segment reductions through the C ABI, per-edge glue as torch
            # ops on the HIP tensors), in CSR-sorted edge order
            return self._decomposed_call(pg, h, coord, m_prev_sorted, skip_coords)
        if self.hidden_nf not in self._KERNEL_WIDTHS:
            h_out, x_out, m_sorted, att, natt = self._padded_call(pg, h, coord, m_prev_sorted, need_m, desc[2])
        else:
            params, pstruct = self._params_cached()
            h_out, x_out, m_sorted, att, natt = PF.egnn_layer(
                h, coord, m_prev_sorted, pg, desc, need_m, params, pstruct)
        # (plain attributes through __dict__: nn.Module.__setattr__ costs 2-3 us a piece on the host, three per layer call)
        d = self.__dict__
        d['_att_src'] = None if att is None else (
            lambda: PF.rows_to_input_order(att.detach()[:pg.n_edges].reshape(-1, 1), pg))
        d['_natt_src'] = None if natt is None else (lambda: natt.detach().reshape(-1, 1))
        if skip_coords:
            def coords_on_demand(h=h.detach(), coord=coord.detach(),
                                 mp=None if m_prev_sorted is None else m_prev_sorted.detach()):
                with torch.no_grad():
                    return self.forward_prepared(pg, h, coord, mp, need_m=False)[1]
            d['_coords_src'] = coords_on_demand
        elif self.use_coords:
            d['_coords_src'] = lambda: x_out.detach()
        return h_out, x_out, m_sorted

    def _decomposed_call(self, pg, h, coord, m_prev_sorted, skip_coords):
        """EGNNLayer.forward (egnn_satorras.py:189-206) for hidden sizes above the fused kernels' 128 channels (any
        width: the reference accepts any --channels): coord2radial -> edge_model -> edge residual -> coord_model ->
        node_model on the prepared graph's sorted edge list. Not fused (the [E, H] intermediates live in HBM and
        autograd keeps them), so it runs at a fraction of the fused layers' rate; same values as the reference
        (oracle-checked at k = 96 ... 1100)."""
        e = pg.n_edges
        row, col = pg.t['row'][:e].long(), pg.t['col'][:e].long()
        edge_index = torch.stack([row, col])
        edge_attr = None
        if self.edges_in_d:
            edge_attr = torch.nn.functional.one_hot(pg.t['etype'][:e].long(), self.edges_in_d).to(h.dtype)
        radial, coord_diff = self.coord2radial(edge_index, coord)
        m = self.edge_model(h[row], h[col], radial, edge_attr)
        if self.edge_residual and m_prev_sorted is not None:
            if self.rezero:
                m = m_prev_sorted + self.edge_gate_parameter * m
            elif self.gated_residual:
                gate = torch.relu(self.edge_gate_parameter)
                m = gate * m + (1 - gate) * m_prev_sorted
            else:
                m = m + m_prev_sorted
        self._att_src = self._natt_src = None
        if skip_coords:
            x_out = coord
            self._coords_src = lambda h=h.detach(), coord=coord.detach(), mp=(
                None if m_prev_sorted is None else m_prev_sorted.detach()): self._decomposed_coords(pg, h, coord, mp)
        else:
            x_out = self.coord_model(coord, edge_index, coord_diff, m)
        h_out, _ = self.node_model(h, edge_index, m)
        if self._att_src is not None:      # node_model recorded the gates in sorted order
            att_sorted = self._att_src
            self._att_src = lambda: PF.rows_to_input_order(att_sorted(), pg)
        return h_out, x_out, m

    def _decomposed_coords(self, pg, h, coord, mp):
        with torch.no_grad():
            att, natt = self._att_src, self._natt_src
            x = self._decomposed_call(pg, h, coord, mp, skip_coords=False)[1]
            self._att_src, self._natt_src = att, natt
            return x

    def forward(self, h, edge_index, coord, edge_attr=None, edge_messages=None):
        """Same contract as the reference: returns (h, coord, edge_attr, edge_feat), edge_feat in
        the caller's edge order. `coord` is NOT modified in place (the reference's `coord += agg`
        aliasing quirk, SURVEY.md §8a Q2, is not reproduced; returned values are identical)."""
        if (edge_attr is None) != (self.edges_in_d == 0):
            raise ValueError(f'layer built with edges_in_d={self.edges_in_d}, edge_attr '
                             f'{"missing" if edge_attr is None else "given"}')
        pg = prepared_for(edge_index, edge_attr, h.size(0))
        m_prev = None
        if self.edge_residual and edge_messages is not None:
            m_prev = PF.rows_to_sorted_order(edge_messages, pg)
        h_out, x_out, m_sorted = self.forward_prepared(pg, h, coord, m_prev, need_m=True)
        return h_out, x_out, edge_attr, PF.rows_to_input_order(m_sorted, pg)

    # ---- the reference's public sub-methods (egnn_satorras.py:123-187) ----
    # Nothing in the reference calls these from outside `forward`; the layer itself runs fused
    # (pvs_egnn_layer_fwd/bwd). They are kept for code written against the decomposed API: the dense
    # products and the segment reductions go through the C ABI (pvs_linear_*, pvs_segment_reduce_*),
    # the per-edge gathers / concatenation / activations are plain torch ops on the HIP tensors, with
    # autograd. Same values as the fused layer up to fp32 summation order
    # (tests/test_gpu_properties.py::test_public_submethods_compose_to_the_fused_layer).
    @staticmethod
    def _mlp(seq, x):
        for mod in seq:
            if isinstance(mod, nn.Linear):
                if min(mod.in_features, mod.out_features) > 64 or mod.out_features > 64:
                    # a plain GEMM wider than the MFMA linear kernels are built for (hidden sizes above 64):
                    # the library GEMM (hipBLASLt through torch) instead of the generic one-row-per-lane kernel
                    _lib.require_hip(x)
                    x = torch.nn.functional.linear(x, mod.weight, mod.bias)
                else:
                    x = PF.linear(x, mod.weight, mod.bias)
            elif isinstance(mod, GraphNorm):      # no batch vector: one graph
                _lib.require_hip(x)
                out = x - x.mean(dim=0, keepdim=True) * mod.mean_scale
                x = mod.weight * out / (out.pow(2).mean(dim=0, keepdim=True) + mod.eps).sqrt() + mod.bias
            else:
                x = mod(x)
        return x

    def edge_model(self, source, target, radial, edge_attr):
        """:123-132: edge_mlp on [h_i, h_j, radial, edge_attr] (or [h_i + h_j, ...])."""
        _lib.require_hip(source, target, radial)
        inp = [source + target, radial] if self.permutation_invariance else [source, target, radial]
        if edge_attr is not None:
            inp.append(edge_attr.to(source.dtype))
        return self._mlp(self.edge_mlp, torch.cat(inp, dim=1))

    def node_model(self, x, edge_index, m_ij):
        """:134-166: attention gate, sum aggregation by row, node_mlp, node gate, residual variants.
        Returns (out, cat([x, agg]))."""
        _lib.require_hip(x, m_ij)
        row = edge_index[0]
        n = x.size(0)
        if self.edge_attention:
            att_val = self._mlp(self.att_mlp, m_ij)
            if self.softmax_attention:      # scatter_softmax over the edges that share a row
                idx = row.unsqueeze(-1)
                gmax = torch.full((n, 1), float('-inf'), dtype=att_val.dtype, device=att_val.device)
                gmax = gmax.scatter_reduce(0, idx, att_val.detach(), 'amax', include_self=True)
                shifted = (att_val - gmax[row]).exp()
                att_val = shifted / unsorted_segment_sum(shifted, row, n)[row]
            self._att_src = lambda: att_val.detach()
            agg = unsorted_segment_sum(att_val * m_ij, row, num_segments=n)
        else:
            agg = unsorted_segment_sum(m_ij, row, num_segments=n)
        agg = torch.cat([x, agg], dim=1)
        out = self._mlp(self.node_mlp, agg)
        if self.node_attention:
            natt = self._mlp(self.node_att_mlp, out)
            out = out * natt
            self._natt_src = lambda: natt.detach()
        if self.residual:
            if self.rezero:
                out = x + self.node_gate_parameter * out
            elif self.gated_residual:
                gate = torch.relu(self.node_gate_parameter)
                out = gate * out + (1 - gate) * x
            else:
                out = x + out
        return out, agg

    def coord_model(self, coord, edge_index, coord_diff, edge_feat):
        """:168-176: coord + mean over the row's edges of coord_diff * coord_mlp(m). Out of place
        (the reference adds in place; returned values are identical)."""
        if not self.use_coords:
            return coord
        _lib.require_hip(coord, coord_diff, edge_feat)
        trans = coord_diff * self._mlp(self.coord_mlp, edge_feat)
        out = coord + unsorted_segment_mean(trans, edge_index[0], num_segments=coord.size(0))
        self._coords_src = lambda: out.detach()
        return out

    def coord2radial(self, edge_index, coord):
        """:178-187: squared distance and (optionally normalised, norm detached) difference per edge."""
        _lib.require_hip(coord)
        row, col = edge_index[0], edge_index[1]
        coord_diff = coord[row] - coord[col]
        radial = torch.sum(coord_diff ** 2, 1).unsqueeze(1)
        if self.normalize:
            coord_diff = coord_diff / (torch.sqrt(radial).detach() + self.epsilon)
        return radial, coord_diff


class SartorrasEGNN(PNNGeometricBase):
    """Equivariant network based on EGNNLayer (egnn_satorras.py:209-329)."""

    def _layer_flags(self, idx, num_layers, kw):
        return kw['edge_attention'], kw['node_attention']

    def _build_layers(self, dim_input, k, num_layers, act_fn, kw):
        layers = [PygLinearPass(nn.Linear(dim_input, k), return_coords_and_edges=True)]
        for idx in range(num_layers):
            edge_att, node_att = self._layer_flags(idx, num_layers, kw)
            layers.append(EGNNLayer(
                k, k, k, edges_in_d=3, act_fn=act_fn, residual=kw['residual'],
                edge_attention=edge_att, normalize=kw['normalize'], graphnorm=kw['graphnorm'],
                tanh=kw['tanh'], update_coords=kw['update_coords'],
                permutation_invariance=kw['permutation_invariance'],
                attention_activation_fn=kw['attention_activation_fn'], node_attention=node_att,
                edge_residual=kw['edge_residual'], gated_residual=kw['gated_residual'],
                rezero=kw['rezero'], softmax_attention=kw['softmax_attention']))
        return layers

    def build_net(self, dim_input, k, dim_output, act_fn=nn.SiLU(), num_layers=4, residual=True,
                  edge_residual=False, edge_attention=False, normalize=True, tanh=True, dropout=0.0,
                  graphnorm=True, multi_fc=False, update_coords=True, permutation_invariance=False,
                  attention_activation_fn='sigmoid', node_attention=False, gated_residual=False,
                  rezero=False, model_task='classification', include_strain_info=False,
                  final_softplus=False, softmax_attention=False, **kwargs):
        assert not (gated_residual and rezero), 'gated_residual and rezero are incompatible'
        if not 0.0 <= float(dropout or 0.0) < 1.0:
            raise ValueError(f'dropout must be in [0, 1), got {dropout}')
        self.n_layers = num_layers
        self.dropout_p = dropout
        self.residual, self.edge_residual = residual, edge_residual
        self.gated_residual, self.rezero = gated_residual, rezero
        self.model_task = model_task
        self.include_strain_info = include_strain_info
        self.softmax_attention = softmax_attention
        kw = dict(residual=residual, edge_residual=edge_residual, edge_attention=edge_attention,
                  normalize=normalize, tanh=tanh, graphnorm=graphnorm, update_coords=update_coords,
                  permutation_invariance=permutation_invariance,
                  attention_activation_fn=attention_activation_fn, node_attention=node_attention,
                  gated_residual=gated_residual, rezero=rezero, softmax_attention=softmax_attention)
        kw.update({key: kwargs.get(key, False) for key in (
            'node_attention_final_only', 'edge_attention_final_only', 'node_attention_first_only',
            'edge_attention_first_only')})
        layers = self._build_layers(dim_input, k, num_layers, act_fn, kw)
        if include_strain_info:
            k += 1
        fc_dims = ((k, 32), (32, 16), (16, dim_output)) if multi_fc else ((k, dim_output),)
        head = []
        for idx, (n_in, n_out) in enumerate(fc_dims):
            head.append(nn.Linear(n_in, n_out))
            if idx < len(fc_dims) - 1:
                head.append(nn.SiLU())
        if final_softplus:
            head.append(nn.Softplus())
        self.feats_linear_layers = nn.Sequential(*head)
        return nn.Sequential(*layers)

    def edge_dropout(self, edges, edge_attributes):
        """egnn_satorras.py:320-323: dropout_adj(edges, edge_attributes, dropout, force_undirected=True,
        training=self.training) ahead of the layer stack; identity when dropout == 0 or in eval mode. The draw
        is keyed on (torch.initial_seed(), calls so far): reproducible under torch.manual_seed, not torch's
        stream (no parity vectors). Edge messages come back in the order of the DROPPED edge list,
        as in the reference."""
        p = float(getattr(self, 'dropout_p', 0.0) or 0.0)
        if p == 0.0 or not self.training:
            return edges, edge_attributes
        if int(edges.shape[1]) == 0:          # nothing to draw from (and no device pointer to hand to the library)
            return edges, edge_attributes
        # the step of the draw: epochs done (restored from a checkpoint) and optimiser steps taken in this run (+ the
        # calls since the last one, for several forwards per step), so that a resumed run does not replay the masks
        # of its first epoch; the rank goes into the seed, so that data-parallel ranks do not drop the same edge
        # indices
        base = ((getattr(self, 'p_epoch', 0) + getattr(self, 'a_epoch', 0)) << 24) + getattr(self, 'global_iter', 0)
        if base != getattr(self, '_dropout_base', None):
            self._dropout_base, self._dropout_calls = base, 0
        self._dropout_calls = getattr(self, '_dropout_calls', 0) + 1
        rank = 0
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            rank = torch.distributed.get_rank()
        seed = (torch.initial_seed() + 0x9E3779B97F4A7C15 * rank) & (2 ** 64 - 1)
        # (epochs, optimiser steps, calls) -> ONE 64-bit step by a mix, not by packing bit fields (with
        # fields, more than 255 forwards under one base - a loop that never bumps global_iter - carried into the next
        # base's range, and 2^24 optimiser steps into the epoch field: masks could repeat)
        return PF.dropout_adj(edges, edge_attributes, p, force_undirected=True, training=True,
                              seed=seed, step=_mix64(_mix64(int(base)) + self._dropout_calls))

    def embed_prepared(self, pg, feats, coords, need_messages=False, trace=None, need_coords=True):
        """Layer stack on a PreparedGraph. Edge messages stay in sorted order between layers and
        are only materialised where a consumer exists (edge_residual, or need_messages).
        need_coords=False: the caller ignores the returned coordinates, so the last layer skips its
        coordinate branch (nothing downstream reads x_L)."""
        embed = self.layers[0]
        feats = embed.embed(feats, coords)
        if trace is not None:
            trace['h0'], trace['x0'] = feats, coords
        m_sorted = None
        egnn_layers = list(self.layers)[1:]
        if not need_messages and egnn_layers:
            plan = self._stack_plan(pg, egnn_layers, need_coords or trace is not None)
            if plan is not None:
                return self._embed_stack(pg, plan, egnn_layers, feats, coords, trace)
        for idx, layer in enumerate(egnn_layers):
            last = idx == len(egnn_layers) - 1
            need_m = (need_messages and last) or (self.edge_residual and not last)
            feats, coords, m_sorted = layer.forward_prepared(
                pg, feats, coords, m_sorted, need_m=need_m,
                need_coords=need_coords or not last or trace is not None)
            if trace is not None:
                trace[f'h{idx + 1}'], trace[f'x{idx + 1}'] = feats, coords
        return feats, coords, m_sorted

    # ---- the layer loop as one call each way (pvs_egnn_stack_fwd / _bwd) ----
    # PVS_EGNN_STACK=0: always one autograd node and one C call per layer; =1: the one-call stack wherever it applies;
    # unset: the stack where the HOST sets the pace - batches of up to 2^26 edge-channels (E x hidden; the reference's
    # default shape has 6 x 10^6, a BASELINE batch 3 x 10^8). Above that the device is several times slower than the host
    # either way (profiles/r06_ab_layer_stack.txt: cfg2 / cfg3 within 0.5 % of each other) and the per-layer nodes stay:
    # they let a multi-rank run send the late layers' gradient bucket while the early layers' backward is still running
    # (distributed.py), and they free each layer's saved tensors as the backward passes it.
    _STACK_HOST_BOUND_EDGE_CHANNELS = 1 << 26

    def _stack_plan(self, pg, egnn_layers, need_coords):
        """The StackPlan for this call, or None where the per-layer path has to run: edge residual (messages travel
        between layers), widths the kernels are not built for (padded / decomposed layers), parameters that are not plain
        fp32 device tensors, layers of different width."""
        mode = os.environ.get('PVS_EGNN_STACK', '')
        if mode == '0' or os.environ.get('PVS_WIDE') == 'decomposed':
            return None
        hidden = egnn_layers[0].hidden_nf
        for layer in egnn_layers:
            if (not isinstance(layer, EGNNLayer) or layer.edge_residual or layer.hidden_nf != hidden
                    or hidden not in (16, 32, 64) or layer.edges_in_d != egnn_layers[0].edges_in_d):
                return None
        if mode != '1' and pg.n_edges * hidden > self._STACK_HOST_BOUND_EDGE_CHANNELS:
            return None
        last = egnn_layers[-1]
        skip_coords = last.use_coords and not need_coords and not os.environ.get('PVS_EGNN_KEEP_DEAD_COORDS')
        cached = self.__dict__.get('_stack_cache')
        pstructs, param_tuples = [], []
        for layer in egnn_layers:
            params, pstruct = layer._params_cached()
            if pstruct is None:
                return None
            pstructs.append(pstruct)
            param_tuples.append(params)
        # (the descriptors are re-derived every call, like the per-layer path does: a flag flipped on a built layer -
        # `layer.edge_attention = False` - must not meet a stale plan; ~2 us per layer)
        descs = [layer._desc() for layer in egnn_layers]
        if skip_coords:
            d = descs[-1]
            descs[-1] = (d[0], d[1], d[2] & ~_lib.UPDATE_COORDS, d[3])
        if cached is not None and cached[0] == skip_coords and len(cached[1].pstructs) == len(pstructs) \
                and all(a is b for a, b in zip(cached[1].pstructs, pstructs)) and cached[1].desc_tuples == tuple(descs):
            return cached[1]
        plan = PF.StackPlan(descs, param_tuples, pstructs)
        plan.skip_coords = skip_coords
        self.__dict__['_stack_cache'] = (skip_coords, plan)
        return plan

    def _embed_stack(self, pg, plan, egnn_layers, feats, coords, trace):
        n, e, hid = pg.n_nodes, pg.n_edges, plan.hidden
        h_out, x_out, h_mid, x_mid, att, natt = PF.egnn_stack(feats, coords, pg, plan)
        nl = plan.n_layers

        def h_of(k):       # output of layer k
            return h_out if k == nl - 1 else h_mid[k, :n * hid].view(n, hid)

        def x_of(k):
            return x_out if k == nl - 1 else x_mid[k, :3 * n].view(n, 3)

        for k, layer in enumerate(egnn_layers):
            d = layer.__dict__
            d['_att_src'] = None if not layer.edge_attention else (
                lambda k=k: PF.rows_to_input_order(att[k, :e].detach().reshape(-1, 1), pg))
            d['_natt_src'] = None if not layer.node_attention else (lambda k=k: natt[k, :n].detach().reshape(-1, 1))
            if k == nl - 1 and plan.skip_coords:
                def coords_on_demand(layer=layer, h=(feats if nl == 1 else h_of(nl - 2)).detach(),
                                     x=(coords if nl == 1 else x_of(nl - 2)).detach()):
                    with torch.no_grad():
                        return layer.forward_prepared(pg, h, x, None, need_m=False)[1]
                d['_coords_src'] = coords_on_demand
            elif layer.use_coords:
                d['_coords_src'] = lambda k=k: x_of(k).detach()
            if trace is not None:
                trace[f'h{k + 1}'], trace[f'x{k + 1}'] = h_of(k), x_of(k)
        return h_out, x_out, None

    def __getstate__(self):
        state = super().__getstate__() if hasattr(super(), '__getstate__') else self.__dict__.copy()
        state = dict(state)
        state.pop('_stack_cache', None)      # (ctypes arrays of pointers: never pickled or deep-copied; rebuilt on use)
        return state

    def get_embeddings(self, feats, edges, coords, edge_attributes, batch):
        """Reference signature (egnn_satorras.py:319-329): returns (feats, edge_messages) with
        edge_messages in the caller's edge order."""
        edges, edge_attributes = self.edge_dropout(edges, edge_attributes)
        pg = prepared_for(edges, edge_attributes, feats.size(0))
        if (batch is not None and torch.is_grad_enabled() and batch.numel() > 1 and not pg.c.graph_eptr
                and not torch.cuda.is_current_stream_capturing()):
            # the batch vector names the graphs: tiles of the fp16-split backward end where a graph ends (one host
            # sync for the number of graphs - the reference's forward has the same one, pnn_geometric_base.py:27).
            # Skipped when the PreparedGraph already knows its graphs (a cached one) and under hipGraph capture (a
            # data-dependent shape; only accuracy across graphs of very different gradient magnitude is at stake).
            # `batch` is the PyG batch vector: sorted, as Batch.from_data_list and the reference's loader build it.
            counts = torch.unique_consecutive(batch, return_counts=True)[1]
            pg.set_graph_ptr(torch.cat([counts.new_zeros(1), counts.cumsum(0)]))
        feats, _, m_sorted = self.embed_prepared(pg, feats, coords, need_messages=True, need_coords=False)
        edge_messages = None if m_sorted is None else PF.rows_to_input_order(m_sorted, pg)
        return feats, edge_messages


def _mix64(x):
    """splitmix64's finaliser: a bijection of 64-bit integers that spreads nearby inputs."""
    x = (x + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    return x ^ (x >> 31)


def unsorted_segment_sum(data, segment_ids, num_segments):
    """egnn_satorras.py:332-337: rows of `data` summed per segment id (HIP, deterministic order).
    Inside the layers this sum is fused into the edge kernels."""
    return PF.segment_reduce(data, segment_ids, num_segments, mean=False)


def unsorted_segment_mean(data, segment_ids, num_segments):
    """egnn_satorras.py:340-347: per-segment sum divided by max(count, 1)."""
    return PF.segment_reduce(data, segment_ids, num_segments, mean=True)
