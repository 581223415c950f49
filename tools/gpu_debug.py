"""GPU bring-up report: per-tensor errors of the HIP path against the golden vectors, without
stopping at the first mismatch. Usage: python tools/gpu_debug.py [case ...]"""
import sys
import traceback
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

from oracle import egnn_oracle as orc  # noqa: E402
from tests._golden import CASES, GoldenCase, rel_err  # noqa: E402
from tests.test_gpu_parity import build_model, make_batch  # noqa: E402


def check_prepare(c):
    from pointvs_amd.graph import prepare_graph
    ei = c.edge_index.cuda()
    pg = prepare_graph(ei, c.edge_attr.cuda(), c.x.shape[0])
    pg.check_status()
    rows = c.edge_index[0].numpy()
    cols = c.edge_index[1].numpy()
    perm = np.argsort(rows, kind='stable')
    t = {k: v.cpu().numpy() for k, v in pg.t.items()}
    E = len(rows)
    ok = (np.array_equal(t['perm'][:E], perm) and np.array_equal(t['row'][:E], rows[perm]) and
          np.array_equal(t['col'][:E], cols[perm]) and
          np.array_equal(t['etype'][:E], c.edge_type.numpy()[perm]))
    rowptr = np.searchsorted(rows[perm], np.arange(c.x.shape[0] + 1))
    ok = ok and np.array_equal(t['rowptr'], rowptr)
    scol = cols[perm]
    cedge = np.argsort(scol, kind='stable')
    ok = ok and np.array_equal(t['cedge'][:E], cedge)
    ok = ok and np.array_equal(t['colptr'], np.searchsorted(scol[cedge], np.arange(c.x.shape[0] + 1)))
    deg = np.diff(rowptr)
    ok = ok and np.allclose(t['inv_deg'], 1.0 / np.maximum(deg, 1))
    return ok


def report(name):
    c = GoldenCase(name)
    print(f'=== {name}: prepare ok = {check_prepare(c)}')
    model = build_model(c)
    g = make_batch(c)
    from pointvs_amd.graph import prepared_for
    feats, edges, coords, eattr, batch = model.unpack_graph(g)
    pg = prepared_for(edges, eattr, feats.size(0))
    trace = {}
    model.embed_prepared(pg, feats, coords, need_messages=True, trace=trace)
    n_layers = orc.layer_flags(c.cfg, 0)['num_layers']
    worst = 0.0
    for li in range(n_layers + 1):
        eh = rel_err(trace[f'h{li}'].detach().cpu().numpy(), c.out[f'h{li}'])
        ex = rel_err(trace[f'x{li}'].detach().cpu().numpy(), c.out[f'x{li}'])
        worst = max(worst, eh, ex)
        print(f'   layer {li}: h {eh:.2e}  x {ex:.2e}')
    for li, layer in enumerate(list(model.layers)[1:], start=1):
        if f'att{li}' in c.out:
            print(f'   att{li} {rel_err(layer.att_val, c.out[f"att{li}"]):.2e}', end='')
        if f'natt{li}' in c.out:
            print(f'   natt{li} {rel_err(layer.node_att_val, c.out[f"natt{li}"]):.2e}', end='')
    print()
    model.zero_grad()
    y_pred, _, _, _ = model.unpack_input_data_and_predict(make_batch(c))
    print(f'   logits {rel_err(y_pred.detach().cpu().numpy(), c.out["logits"]):.2e}')
    loss = model.get_loss(c.y_true.cuda(), y_pred)
    loss.backward()
    _, _, g64 = orc.forward_backward(c.sd, c.cfg, c.x, c.pos, c.edge_index, c.edge_attr, c.batch,
                                     c.y_true, dtype=torch.float64)
    for pname, p in model.named_parameters():
        ref = g64[pname]
        if p.grad is None or ref is None:
            flag = '' if (p.grad is None) == (ref is None) else '   <-- None mismatch'
            print(f'   grad {pname:40s} none={p.grad is None} ref_none={ref is None}{flag}')
            continue
        e = rel_err(p.grad.cpu().numpy(), ref.numpy())
        worst = max(worst, e)
        mark = '   <-- BAD' if e > 1e-5 else ''
        print(f'   grad {pname:40s} {e:.2e}  (max|ref| {ref.abs().max():.2e}){mark}')
    print(f'   WORST {worst:.2e}')


if __name__ == '__main__':
    names = sys.argv[1:] or ['c0_clidefault_g5batch', 'c1_testkwargs_g2', 'c3_all_on_k32_g5']
    for n in names:
        try:
            report(n)
        except Exception:
            traceback.print_exc()
