"""CPU oracle for the PointVS EGNN hot path.  TEST INFRASTRUCTURE - NOT PRODUCT CODE.

A functional, `state_dict`-driven restatement in plain torch (CPU, fp32 or fp64, autograd for
the backward) of what the reference computes on the path SURVEY.md §8 scopes:

    EGNNLayer.forward        /root/reference/point_vs/models/geometric/egnn_satorras.py:189-206
      coord2radial           egnn_satorras.py:178-187
      edge_model             egnn_satorras.py:123-132
      edge residual block    egnn_satorras.py:194-202
      coord_model            egnn_satorras.py:168-176
      node_model             egnn_satorras.py:134-166
      unsorted_segment_sum   egnn_satorras.py:332-337
      unsorted_segment_mean  egnn_satorras.py:340-347
    SartorrasEGNN.get_embeddings / build_net   egnn_satorras.py:212-329
    MultitaskSatorrasEGNN.build_net / forward  egnn_multitask.py:14-166
    PNNGeometricBase.forward / PygLinearPass   pnn_geometric_base.py:24-41, 83-94
    get_loss / backprop                        point_neural_network_base.py:362-370, 417-429

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this
module, and only as the checker / the reported CPU baseline.  The product package
(`pointvs_amd`) never imports it and has no CPU fallback.

Pinning: checked against the golden vectors in `tests/golden/*.npz`, which were produced by
importing the real reference (`tests/golden/make_golden.py`).  Three third-party pieces the
reference calls are absent from `/root/reference` and from this image; they are restated here from
their published semantics and are only pinned as far as the reference's own tests pin them
(SURVEY.md §8c):
    torch_scatter.composite.scatter_softmax (pytorch-scatter 2.1.0)  -> segment_softmax
    torch_geometric.nn.norm.GraphNorm (pyg 2.0.4, no batch vector)    -> graph_norm
    torch_geometric.nn.global_mean_pool (pyg 2.0.4)                   -> mean_pool
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

# Defaults of SartorrasEGNN.build_net (egnn_satorras.py:212-238); CLI passes everything explicitly.
BUILD_NET_DEFAULTS = dict(
    num_layers=4, residual=True, edge_residual=False, edge_attention=False, normalize=True,
    tanh=True, dropout=0.0, graphnorm=True, multi_fc=False, update_coords=True,
    permutation_invariance=False, attention_activation_fn='sigmoid', node_attention=False,
    gated_residual=False, rezero=False, model_task='classification', final_softplus=False,
    softmax_attention=False, node_attention_final_only=False, edge_attention_final_only=False,
    node_attention_first_only=False, edge_attention_first_only=False)

_ATT_ACT = {'sigmoid': torch.sigmoid, 'tanh': torch.tanh, 'relu': torch.relu, 'silu': F.silu}


def segment_sum(data, ids, n):
    """egnn_satorras.py:332-337 - scatter-add of rows by segment id, in edge order."""
    out = data.new_zeros((n, data.size(1)))
    return out.index_add_(0, ids, data)


def segment_mean(data, ids, n):
    """egnn_satorras.py:340-347 - sum / max(count, 1)."""
    total = segment_sum(data, ids, n)
    count = segment_sum(torch.ones_like(data), ids, n)
    return total / count.clamp(min=1)


def segment_softmax(src, ids, n):
    """torch_scatter.composite.scatter_softmax over dim 0 (egnn_satorras.py:140-142)."""
    idx = ids.unsqueeze(-1).expand_as(src)
    gmax = torch.full((n, src.size(1)), float('-inf'), dtype=src.dtype)
    gmax = gmax.scatter_reduce(0, idx, src.detach(), 'amax', include_self=True)
    shifted = (src - gmax.gather(0, idx)).exp()
    gsum = segment_sum(shifted, ids, n)
    return shifted / gsum.gather(0, idx)


def graph_norm(x, weight, bias, mean_scale, eps=1e-5):
    """GraphNorm called inside nn.Sequential => no batch vector => one graph."""
    mean = x.mean(dim=0, keepdim=True)
    out = x - mean * mean_scale
    var = out.pow(2).mean(dim=0, keepdim=True)
    return weight * out / (var + eps).sqrt() + bias


def mean_pool(x, batch, n_graphs):
    total = x.new_zeros((n_graphs, x.size(1))).index_add_(0, batch, x)
    count = x.new_zeros((n_graphs,)).index_add_(0, batch, torch.ones_like(batch, dtype=x.dtype))
    return total / count.clamp(min=1).unsqueeze(-1)


def layer_flags(cfg, idx):
    """Per-layer attention placement (egnn_multitask.py:99-123); plain EGNN = same on all."""
    kw = dict(BUILD_NET_DEFAULTS)
    kw.update(cfg)
    n = kw['num_layers']

    def placed(on, first_only, final_only):
        if not on:
            return False
        if not first_only and not final_only:
            return True
        return (first_only and idx == 0) or (final_only and idx == n - 1)

    multitask = kw.get('_class', 'SartorrasEGNN') == 'MultitaskSatorrasEGNN'
    kw['edge_attention_here'] = placed(
        kw['edge_attention'], multitask and kw['edge_attention_first_only'],
        multitask and kw['edge_attention_final_only'])
    kw['node_attention_here'] = placed(
        kw['node_attention'], multitask and kw['node_attention_first_only'],
        multitask and kw['node_attention_final_only'])
    return kw


def egnn_layer(sd, pre, kw, h, edge_index, coord, edge_attr, edge_messages):
    """One EGNNLayer.forward (egnn_satorras.py:189-206). Returns h', coord', m, att, node_att."""
    row, col = edge_index[0], edge_index[1]
    n = h.size(0)
    # coord2radial :178-187
    diff = coord[row] - coord[col]
    radial = (diff ** 2).sum(1, keepdim=True)
    if kw['normalize']:
        diff = diff / (radial.sqrt().detach() + 1e-8)
    # edge_model :123-132 (edge_attr int64 one-hot promoted to float by cat)
    if kw['permutation_invariance']:
        parts = [h[row] + h[col], radial]
    else:
        parts = [h[row], h[col], radial]
    if edge_attr is not None:
        parts.append(edge_attr.to(h.dtype))
    z = torch.cat(parts, dim=1)
    z = F.silu(F.linear(z, sd[pre + 'edge_mlp.0.weight'], sd[pre + 'edge_mlp.0.bias']))
    m = F.silu(F.linear(z, sd[pre + 'edge_mlp.2.weight'], sd[pre + 'edge_mlp.2.bias']))
    # edge residual :194-202
    if kw['edge_residual'] and edge_messages is not None:
        if kw['rezero']:
            m = edge_messages + sd[pre + 'edge_gate_parameter'] * m
        elif kw['gated_residual']:
            gate = torch.relu(sd[pre + 'edge_gate_parameter'])
            m = gate * m + (1 - gate) * edge_messages
        else:
            m = m + edge_messages
    # coord_model :168-176 (mean aggregation; out of place here, values identical)
    if kw['update_coords']:
        s = F.silu(F.linear(m, sd[pre + 'coord_mlp.0.weight'], sd[pre + 'coord_mlp.0.bias']))
        s = F.linear(s, sd[pre + 'coord_mlp.2.weight'])
        if kw['tanh']:
            s = torch.tanh(s)
        coord = coord + segment_mean(diff * s, row, n)
    # node_model :134-166 (sum aggregation)
    att = None
    if kw['edge_attention_here']:
        att = F.linear(m, sd[pre + 'att_mlp.0.weight'], sd[pre + 'att_mlp.0.bias'])
        if kw['softmax_attention']:
            att = segment_softmax(att, row, n)
        else:
            att = _ATT_ACT[kw['attention_activation_fn']](att)
        agg = segment_sum(att * m, row, n)
    else:
        agg = segment_sum(m, row, n)
    out = F.linear(torch.cat([h, agg], dim=1),
                   sd[pre + 'node_mlp.0.weight'], sd[pre + 'node_mlp.0.bias'])
    if kw['graphnorm']:
        out = graph_norm(out, sd[pre + 'node_mlp.1.weight'], sd[pre + 'node_mlp.1.bias'],
                         sd[pre + 'node_mlp.1.mean_scale'])
    out = F.linear(F.silu(out), sd[pre + 'node_mlp.3.weight'], sd[pre + 'node_mlp.3.bias'])
    natt = None
    if kw['node_attention_here']:
        natt = F.linear(out, sd[pre + 'node_att_mlp.0.weight'], sd[pre + 'node_att_mlp.0.bias'])
        if not kw['softmax_attention']:  # Identity activation when softmax_attention (:66-71)
            natt = _ATT_ACT[kw['attention_activation_fn']](natt)
        out = out * natt
    if kw['residual']:
        if kw['rezero']:
            out = h + sd[pre + 'node_gate_parameter'] * out
        elif kw['gated_residual']:
            gate = torch.relu(sd[pre + 'node_gate_parameter'])
            out = gate * out + (1 - gate) * h
        else:
            out = h + out
    return out, coord, m, att, natt


def model_forward(sd, cfg, x, pos, edge_index, edge_attr, batch, n_graphs=None, trace=None):
    """PNNGeometricBase.forward / MultitaskSatorrasEGNN.forward. Returns per-graph outputs.

    sd: dict name -> tensor (leaf tensors with requires_grad for the backward).
    cfg: build_net kwargs (+ '_class').  trace: optional dict that receives per-layer tensors.
    """
    dtype = sd['layers.0.m.weight'].dtype
    h = F.linear(x.to(dtype), sd['layers.0.m.weight'], sd['layers.0.m.bias'])  # PygLinearPass
    coord = pos.to(dtype)
    if trace is not None:
        trace['h0'], trace['x0'] = h, coord
    kw0 = layer_flags(cfg, 0)
    m = None
    for li in range(kw0['num_layers']):
        kw = layer_flags(cfg, li)
        h, coord, m, att, natt = egnn_layer(
            sd, f'layers.{li + 1}.', kw, h, edge_index, coord, edge_attr, m)
        if trace is not None:
            trace[f'h{li + 1}'], trace[f'x{li + 1}'] = h, coord
            trace[f'att{li + 1}'], trace[f'natt{li + 1}'] = att, natt
            trace['m_last'] = m
    if n_graphs is None:
        n_graphs = int(batch.max()) + 1
    pooled = h.mean(dim=0) if n_graphs == 1 else mean_pool(h, batch, n_graphs)
    if kw0.get('_class') == 'MultitaskSatorrasEGNN':
        if 'classification' in kw0['model_task']:
            out = F.linear(pooled, sd['feats_linear_layers_pose.0.weight'],
                           sd['feats_linear_layers_pose.0.bias'])
        else:
            out = F.linear(pooled, sd['feats_linear_layers_affinity.0.weight'],
                           sd['feats_linear_layers_affinity.0.bias'])
            out = F.softplus(out) if kw0['final_softplus'] else torch.relu(out)
        return out
    idx, out = 0, pooled
    n_fc = 3 if kw0['multi_fc'] else 1
    for fc in range(n_fc):
        out = F.linear(out, sd[f'feats_linear_layers.{idx}.weight'],
                       sd[f'feats_linear_layers.{idx}.bias'])
        idx += 1
        if fc < n_fc - 1:
            out = F.silu(out)
            idx += 1
    if kw0['final_softplus']:
        out = F.softplus(out)
    return out


def loss_fn(cfg, y_pred, y_true):
    """get_loss (point_neural_network_base.py:362-370): BCE-with-logits or MSE."""
    task = dict(BUILD_NET_DEFAULTS, **cfg)['model_task']
    if task == 'classification':
        return F.binary_cross_entropy_with_logits(y_pred, y_true)
    return F.mse_loss(y_pred, y_true)


def forward_backward(sd_np, cfg, x, pos, edge_index, edge_attr, batch, y_true,
                     dtype=torch.float32, trace=None):
    """Forward + loss + autograd backward. Returns (y_pred, loss, grads dict or None per name)."""
    sd = {k: torch.as_tensor(v).to(dtype).clone().requires_grad_(True)
          for k, v in sd_np.items() if torch.as_tensor(v).is_floating_point()}
    y_pred = model_forward(sd, cfg, x, pos, edge_index, edge_attr, batch, trace=trace)
    y_pred = y_pred.reshape(-1)
    loss = loss_fn(cfg, y_pred, torch.as_tensor(y_true).to(dtype).reshape(-1))
    names = list(sd.keys())
    grads = torch.autograd.grad(loss, [sd[n] for n in names], allow_unused=True)
    return y_pred.detach(), loss.detach(), dict(zip(names, grads))


def adam_step(sd_np, grads, lr, wd, clip=1.0, eps=1e-8, b1=0.9, b2=0.999):
    """clip_grad_value_(1.0) + first torch.optim.Adam step with L2 weight decay
    (point_neural_network_base.py:83-85, 417-422). None-grad parameters are skipped."""
    out = {}
    for name, p in sd_np.items():
        p = torch.as_tensor(p)
        g = grads.get(name)
        if g is None:
            out[name] = p.clone()
            continue
        g = g.to(p.dtype).clamp(-clip, clip) + wd * p
        m = (1 - b1) * g
        v = (1 - b2) * g * g
        m_hat = m / (1 - b1)
        v_hat = v / (1 - b2)
        out[name] = p - lr * m_hat / (v_hat.sqrt() + eps)
    return out
