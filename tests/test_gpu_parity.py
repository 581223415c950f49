"""GPU parity: the HIP path (through the C ABI) against the golden vectors captured from the
reference and against the fp64 run of the oracle, on identical inputs.

Tolerance (BASELINE.json "within 1e-5 fp32"): per tensor
    max|gpu - ref| <= 1e-5 * max(1, max|ref|)                                   (round 1-3, kept)
and, since round 4, the STRICT form that a small tensor cannot pass as zeros (tests/_golden.py):
    max|gpu - ref64| <= 1e-5 * max|ref64| + 4 * max|ref32 - ref64| (+ 1e-12 * largest gradient of the case)
for every per-layer tensor, attention value and parameter gradient; ref64 = the oracle's fp64 run,
ref32 = the reference's own fp32 values from the golden file (the oracle's fp32 run where it has none).
"""
from pathlib import Path

import numpy as np
import pytest
import torch

from tests._golden import CASES, CaseLog, GoldenCase, assert_strict, edge_permutations, grad_floor, rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-5


def build_model(c, device='cuda'):
    from pointvs_amd.egnn_multitask import MultitaskSatorrasEGNN
    from pointvs_amd.egnn_satorras import SartorrasEGNN
    cls = SartorrasEGNN if c.meta['class'] == 'SartorrasEGNN' else MultitaskSatorrasEGNN
    model = cls(Path('/tmp/pvs_test'), c.meta['lr'], c.meta['wd'], None, None, silent=True,
                **c.meta['kwargs'])
    model.load_state_dict({k: torch.from_numpy(v) for k, v in c.sd.items()})
    return model.to(device).eval()


def make_batch(c, device='cuda'):
    from pointvs_amd.graph import Batch
    return Batch(x=c.x.clone(), edge_index=c.edge_index.clone(), edge_attr=c.edge_attr.clone(),
                 pos=c.pos.clone(), batch=c.batch.clone(), y=c.y_true.clone(),
                 lig_fname=['l'] * c.n_graphs, rec_fname=['r'] * c.n_graphs).to(device)


@pytest.mark.parametrize('name', CASES)
def test_forward_backward_match_reference(name):
    from oracle import egnn_oracle as orc
    from pointvs_amd.graph import prepared_for
    c = GoldenCase(name)
    model = build_model(c)
    g = make_batch(c)

    # the arbiters: the oracle's fp64 run (per-layer tensors and gradients) and, where the golden file
    # holds no reference fp32 value for a tensor, the oracle's fp32 run
    t64, t32 = {}, {}
    _, _, g64 = orc.forward_backward(c.sd, c.cfg, c.x, c.pos, c.edge_index, c.edge_attr, c.batch,
                                     c.y_true, dtype=torch.float64, trace=t64)
    _, _, g32 = orc.forward_backward(c.sd, c.cfg, c.x, c.pos, c.edge_index, c.edge_attr, c.batch,
                                     c.y_true, dtype=torch.float32, trace=t32)
    # two more samples of the reference arithmetic's fp32 noise: the same evaluation with the edges in another order
    g32p, t32p = [], []
    # (SIX more orders since round 6, two before: on a many-core host the oracle's fp32 sums are threaded and every sample
    # is a random draw; with three draws in all, the noise-dominated bound of a tiny tensor dipped under the GPU's - always
    # identical - error about once in twenty-five fresh processes: c3_all_on_k32_g5 `layers.1.att_mlp.0.weight`, error
    # 1.011e-09 every time, bound 7.8e-10 ... 3.3e-09. More draws can only raise the bound's maximum; profiles/r06_fuzz_campaign.txt)
    for perm in edge_permutations(c.edge_index.shape[1], count=6):
        tp = {}
        _, _, gp = orc.forward_backward(c.sd, c.cfg, c.x, c.pos, c.edge_index[:, perm], c.edge_attr[perm], c.batch,
                                        c.y_true, dtype=torch.float32, trace=tp)
        inv = torch.argsort(perm)
        for key in [k for k, v in tp.items() if v is not None and (k == 'm_last' or k.startswith('att'))]:
            tp[key] = tp[key][inv]             # per-edge tensors back in the caller's edge order
        g32p.append(gp)
        t32p.append(tp)
    log = CaseLog(name)

    def strict(got, key):
        r64 = t64[key].detach().numpy()
        samples = [t32[key].detach().numpy()] + [tp[key].detach().numpy() for tp in t32p]
        if key in c.out:
            samples.append(c.out[key])
        assert_strict(np.asarray(got).reshape(r64.shape), r64, samples, f'{name} {key}', log=log)

    # --- traced forward through the internal fast path ---
    feats, edges, coords, eattr, batch = model.unpack_graph(g)
    pg = prepared_for(edges, eattr, feats.size(0))
    pg.check_status()
    trace = {}
    _, _, m_sorted = model.embed_prepared(pg, feats, coords, need_messages=True, trace=trace)
    n_layers = orc.layer_flags(c.cfg, 0)['num_layers']
    for li in range(n_layers + 1):
        assert rel_err(trace[f'h{li}'].detach().cpu().numpy(), c.out[f'h{li}']) < TOL, f'h{li}'
        assert rel_err(trace[f'x{li}'].detach().cpu().numpy(), c.out[f'x{li}']) < TOL, f'x{li}'
        strict(trace[f'h{li}'].detach().cpu().numpy(), f'h{li}')
        strict(trace[f'x{li}'].detach().cpu().numpy(), f'x{li}')
    for li, layer in enumerate(list(model.layers)[1:], start=1):
        if f'att{li}' in c.out:
            assert layer.att_val.shape == c.out[f'att{li}'].shape
            assert rel_err(layer.att_val, c.out[f'att{li}']) < TOL, f'att{li}'
            strict(layer.att_val, f'att{li}')
        else:
            assert layer.att_val is None
        if f'natt{li}' in c.out:
            assert rel_err(layer.node_att_val, c.out[f'natt{li}']) < TOL, f'natt{li}'
            strict(layer.node_att_val, f'natt{li}')
        else:
            assert layer.node_att_val is None

    # --- public API: get_embeddings returns edge messages in the caller's edge order ---
    _, m_in = model.get_embeddings(feats, edges, coords, eattr, batch)
    m = m_in.detach().double().cpu().numpy()
    assert rel_err(m.sum(1), c.out['m_rowsum']) < TOL
    assert rel_err(m.sum(0), c.out['m_colsum']) < TOL
    assert rel_err(m[::16], c.out['m_rows16']) < TOL
    # every message element against the fp64 oracle, strict (the golden file keeps every 16th row of the
    # reference's fp32 values; the oracle's fp32 run stands in for the others)
    strict(m_in.detach().cpu().numpy(), 'm_last')

    # --- model forward + loss + backward through the reference-shaped entry points ---
    model.zero_grad()
    y_pred, _, _, _ = model.unpack_input_data_and_predict(make_batch(c))
    assert rel_err(y_pred.detach().cpu().numpy(), c.out['logits']) < TOL
    loss = model.get_loss(c.y_true.cuda(), y_pred)
    assert abs(float(loss.detach()) - float(c.out['loss'])) < TOL * max(1.0, abs(float(c.out['loss'])))
    loss.backward()
    got_none = sorted(n for n, p in model.named_parameters() if p.grad is None)
    if c.grads:
        assert got_none == sorted(c.meta['grad_none'])
    # fp64 oracle as arbiter of the gradients
    floor = grad_floor({k: (None if v is None else v.numpy()) for k, v in g64.items()})
    for pname, p in model.named_parameters():
        if p.grad is None:
            assert g64[pname] is None, pname
            continue
        got = p.grad.detach().cpu().numpy()
        ref64 = g64[pname].numpy()
        if (pname in c.grads and not rel_err(got, c.grads[pname]) < TOL) or not rel_err(got, ref64) < TOL:
            # (round 6: this check failed ONCE in sixteen runs of the whole suite - c3_all_on_k32_g5, never alone, never in
            # 1500 repeats under a concurrent GPU load - and the message said too little: say how far off, and whether a
            # second evaluation of the same step on the same inputs gives the same bits)
            first = {n: q.grad.detach().clone() for n, q in model.named_parameters() if q.grad is not None}
            model.zero_grad()
            y2, _, _, _ = model.unpack_input_data_and_predict(make_batch(c))
            model.get_loss(c.y_true.cuda(), y2).backward()
            moved = sorted(n for n, q in model.named_parameters() if q.grad is not None and not torch.equal(q.grad, first[n]))
            raise AssertionError(
                f'grad {pname}: {rel_err(got, ref64):.3e} from the fp64 oracle'
                + (f', {rel_err(got, c.grads[pname]):.3e} from the reference fp32' if pname in c.grads else '')
                + f' (bound {TOL}); a second evaluation of the step '
                + ('gave the same bits' if not moved else f'gave OTHER bits in {moved[:6]}: '
                   f'now {rel_err(p.grad.detach().cpu().numpy(), ref64):.3e} from the fp64 oracle'))
        samples = [g32[pname].numpy()] + [gp[pname].numpy() for gp in g32p]
        if pname in c.grads:
            samples.append(c.grads[pname])
        assert_strict(got, ref64, samples, f'{name} grad {pname}', floor=floor, log=log)
    log.finish()


def test_run_to_run_bitwise_reproducible():
    """Replaces the reference's test_consistency: no atomics => identical bits every run."""
    c = GoldenCase('c1_testkwargs_g2')
    model = build_model(c)
    outs, grads = [], []
    for _ in range(3):
        model.zero_grad()
        y, _, _, _ = model.unpack_input_data_and_predict(make_batch(c))
        model.get_loss(c.y_true.cuda(), y).backward()
        outs.append(y.detach().cpu().numpy().copy())
        grads.append(np.concatenate([p.grad.detach().cpu().numpy().ravel()
                                     for p in model.parameters() if p.grad is not None]))
    assert outs[0].tobytes() == outs[1].tobytes() == outs[2].tobytes()
    assert grads[0].tobytes() == grads[1].tobytes() == grads[2].tobytes()
    assert abs(float(torch.sigmoid(torch.from_numpy(outs[0]))[0])) > 1e-5


def test_e3_invariance():
    """test/test_invariance.py:35-43: |sigmoid(f(G)) - sigmoid(f(R G))| <= 3e-5."""
    a, b = GoldenCase('c1_testkwargs_g1'), GoldenCase('c1_testkwargs_g3rot')
    model = build_model(a)
    with torch.no_grad():
        ya = torch.sigmoid(model(make_batch(a))).item()
        yb = torch.sigmoid(model(make_batch(b))).item()
    assert ya == pytest.approx(yb, abs=3e-5)


def test_softmax_attention_sums_to_one():
    """test/test_attention.py:22-46 on the 2-graph batch."""
    c = GoldenCase('c1_testkwargs_g2')
    model = build_model(c)
    with torch.no_grad():
        model(make_batch(c))
    rows = c.edge_index[0].numpy()
    checked = False
    for layer in model.layers:
        if hasattr(layer, 'att_val') and layer.att_val is not None:
            sums = np.zeros(rows.max() + 1)
            np.add.at(sums, rows, layer.att_val.squeeze())
            np.testing.assert_allclose(sums, np.ones_like(sums), atol=1e-6)
            checked = True
    assert checked


def test_adam_step_matches_reference_backprop():
    """One reference `backprop()` step (clip 1.0 + Adam, lr 2e-3, wd 1e-4). The first Adam step is
    lr * g / (|g| + 1e-8): where |g| is far above eps the update is well conditioned and must
    match to 1e-5; where |g| ~ eps it amplifies fp32 gradient noise by 1/eps (the reference's own
    summation-order noise does the same), so there the step can only be bounded by 2*lr."""
    for name in ('c0_clidefault_g5batch', 'c2_sigatt_k32_g5batch'):
        c = GoldenCase(name)
        model = build_model(c).train()
        y_pred, _, _, _ = model.unpack_input_data_and_predict(make_batch(c))
        model.backprop(c.y_true.cuda(), y_pred)
        lr = c.meta['lr']
        for k, v in model.state_dict().items():
            if not np.issubdtype(c.adam[k].dtype, np.floating):
                continue
            got, ref = v.cpu().numpy(), c.adam[k]
            if k in c.grads:
                g_eff = np.clip(c.grads[k], -1, 1) + c.meta['wd'] * c.sd[k]
                solid = np.abs(g_eff) > 1e-5
                assert np.abs(got - ref)[solid].max(initial=0.0) < TOL, k
                assert np.abs(got - ref).max() <= 2 * lr * 1.001, k
            else:   # no gradient in the reference => Adam skipped it
                assert np.array_equal(got, c.sd[k]) and np.array_equal(ref, c.sd[k]), k
