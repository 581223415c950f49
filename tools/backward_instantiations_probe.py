"""Every instantiation of the edge forward and backward (edge-residual kind none / sum / rezero / gated x edge attention off / sigmoid /
softmax, H = 32 and 64) on a multi-tile graph with random upstream gradients: run twice (bitwise reproducibility) and against
the exact-fp32-MFMA family (PVS_EGNN_BF16X3=0) of the same library. Prints the worst per-tensor distance of each; the GPU
test tests/test_gpu_stack.py::test_every_backward_instantiation_is_reproducible_and_close_to_the_exact_family holds the same
numbers to bounds.   usage (GPU box): python tools/backward_instantiations_probe.py [--big] [hidden sizes, default 32 64]"""
import os
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from tests import test_gpu_properties as t  # noqa: E402
from pointvs_amd.egnn_satorras import EGNNLayer  # noqa: E402
from pointvs_amd.graph import prepared_for  # noqa: E402


def probe(hidden, kind, att, n=1814, e_draw=45376, n_graphs=4, seed=116):
    """(worst run-to-run distance, tensor), (worst distance to the exact family, tensor); distances relative to the
    tensor's largest magnitude."""
    torch.manual_seed(3)
    kw = dict(edges_in_d=3, residual=True, edge_residual=kind != 'none')
    if kind == 'gated':
        kw['gated_residual'] = True
    if kind == 'rezero':
        kw['rezero'] = True
    if att:
        kw['edge_attention'] = True
        kw['softmax_attention'] = att == 'softmax'
    layer = EGNNLayer(hidden, hidden, hidden, **kw).cuda()
    if kind == 'rezero':
        with torch.no_grad():
            layer.edge_gate_parameter.fill_(0.3)
            layer.node_gate_parameter.fill_(0.2)
    g = t.random_graph(n, e_draw, seed=seed, n_graphs=n_graphs).to('cuda')
    pg = prepared_for(g.edge_index, g.edge_attr, n)
    e = pg.n_edges
    gen = torch.Generator(device='cuda').manual_seed(5)
    h = torch.randn(n, hidden, device='cuda', generator=gen)
    mp = torch.randn(e, hidden, device='cuda', generator=gen)
    gh = torch.randn(n, hidden, device='cuda', generator=gen)
    gx = torch.randn(n, 3, device='cuda', generator=gen)
    gm = torch.randn(e, hidden, device='cuda', generator=gen) * 0.1

    def run(env=None):
        os.environ.update(env or {})
        try:
            hh, xx, mm = h.clone().requires_grad_(), g.pos.clone().requires_grad_(), mp.clone().requires_grad_()
            layer.zero_grad()
            ho, xo, mo = layer.forward_prepared(pg, hh, xx, mm if kind != 'none' else None, need_m=True)
            (ho * gh).sum().add((xo * gx).sum()).add((mo * gm).sum()).backward()
            out = {'h_out': ho, 'x_out': xo, 'm_out': mo, 'g_h': hh.grad, 'g_x': xx.grad}
            if kind != 'none':
                out['g_m_prev'] = mm.grad
            out.update({'g_' + k: p.grad for k, p in layer.named_parameters() if p.grad is not None})
            return {k: v.detach().double().cpu() for k, v in out.items()}
        finally:
            for k_ in (env or {}):
                os.environ.pop(k_, None)

    a, b, ex = run(), run(), run({'PVS_EGNN_BF16X3': '0'})
    worst_rep, worst_ex = (0.0, ''), (0.0, '')
    for k in a:
        s = float(ex[k].abs().max()) or 1.0
        worst_rep = max(worst_rep, (float((a[k] - b[k]).abs().max()) / s, k))
        worst_ex = max(worst_ex, (float((a[k] - ex[k]).abs().max()) / s, k))
    return worst_rep, worst_ex


if __name__ == '__main__':
    # --big: one graph of 20,000 nodes and 3,000,000 edges - every wave of a full grid walks ~45 tiles (the default graph
    # gives a wave two)
    big = '--big' in sys.argv
    shape = dict(n=20000, e_draw=3000000, n_graphs=1) if big else {}
    for hidden in [int(v) for v in sys.argv[1:] if v != '--big'] or [32, 64]:
        for kind in ('none', 'sum', 'rezero', 'gated'):
            for att in (None, 'sigmoid', 'softmax'):
                rep, ex = probe(hidden, kind, att, **shape)
                print(f'H={hidden} residual kind {kind:7s} attention {str(att):8s} run-to-run {rep[0]:8.1e} {rep[1]:24s} '
                      f'vs exact family {ex[0]:8.1e} {ex[1]}', flush=True)
