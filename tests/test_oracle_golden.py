"""The oracle (oracle/egnn_oracle.py) against vectors captured from the real reference."""
import numpy as np
import pytest
import torch

from oracle import egnn_oracle as orc
from tests._golden import CASES, GOLDEN_DIR as GOLDEN, GoldenCase, rel_err

TOL = 2e-6   # fp32 CPU vs fp32 CPU, same op order up to fused/unfused cat-linear


@pytest.mark.parametrize('name', CASES)
def test_oracle_matches_reference_vectors(name):
    torch.set_num_threads(1)
    c = GoldenCase(name)
    trace = {}
    y, loss, grads = orc.forward_backward(
        c.sd, c.cfg, c.x, c.pos, c.edge_index, c.edge_attr, c.batch, c.y_true, trace=trace)
    assert rel_err(y.numpy(), c.out['logits']) < TOL
    assert abs(float(loss) - float(c.out['loss'])) < 1e-5 * max(1.0, abs(float(c.out['loss'])))
    n_layers = orc.layer_flags(c.cfg, 0)['num_layers']
    for li in range(n_layers + 1):
        assert rel_err(trace[f'h{li}'].detach().numpy(), c.out[f'h{li}']) < TOL, f'h{li}'
        assert rel_err(trace[f'x{li}'].detach().numpy(), c.out[f'x{li}']) < TOL, f'x{li}'
        for key in (f'att{li}', f'natt{li}'):
            if key in c.out:
                assert rel_err(trace[key].detach().numpy(), c.out[key]) < TOL, key
            else:
                assert trace.get(key) is None, key
    m = trace['m_last'].detach().double().numpy()
    assert rel_err(m.sum(1), c.out['m_rowsum']) < 1e-5
    assert rel_err(m.sum(0), c.out['m_colsum']) < 1e-5
    assert rel_err(m[::16], c.out['m_rows16']) < TOL
    none_names = sorted(k for k, g in grads.items() if g is None)
    assert none_names == sorted(c.meta['grad_none'])
    for k, g in c.grads.items():
        assert rel_err(grads[k].numpy(), g) < 1e-5, k
    if c.adam:
        new = orc.adam_step(c.sd, grads, c.meta['lr'], c.meta['wd'])
        for k, v in c.adam.items():
            if np.issubdtype(v.dtype, np.floating):
                assert rel_err(new[k].numpy(), v) < 1e-5, k


def test_fp64_arbiter_agrees_with_fp32_reference():
    """fp64 run of the oracle is the arbiter for GPU parity; it must sit within fp32 noise."""
    c = GoldenCase('c0_clidefault_g5batch')
    y64, _, _ = orc.forward_backward(c.sd, c.cfg, c.x, c.pos, c.edge_index, c.edge_attr,
                                     c.batch, c.y_true, dtype=torch.float64)
    assert rel_err(y64.numpy(), c.out['logits']) < 1e-5


# ---- radius-graph builder oracle -------------------------------------------
def test_generate_edges_oracle_matches_reference_test_vectors():
    """The reference's own expected arrays (test/test_preprocessing_fns.py:32-71)."""
    import json
    from oracle.generate_edges_oracle import generate_edges
    d = json.loads((GOLDEN / 'generate_edges_reference_tests.json').read_text())
    xyz = np.stack([d['struct']['x'], d['struct']['y'], d['struct']['z']], axis=1).astype(np.float64)
    for key, prune in (('no_prune', False), ('prune', True)):
        keep, (rows, cols), attrs = generate_edges(xyz, d['struct']['bp'], d['inter_radius'],
                                                   d['intra_radius'], prune=prune)
        assert rows.tolist() == d[key]['rows'] and cols.tolist() == d[key]['cols']
        assert attrs.tolist() == d[key]['attrs']
    assert keep.tolist() == list(range(8))


@pytest.mark.parametrize('name', ['edges_small', 'edges_small_prune', 'edges_default_radii', 'edges_r10',
                                  'edges_bonds', 'edges_no_inter'])
def test_generate_edges_oracle_matches_reference_outputs(name):
    """Outputs of the reference function itself on seeded random structures
    (tests/golden/make_golden_edges.py)."""
    from oracle.generate_edges_oracle import generate_edges
    z = np.load(GOLDEN / f'{name}.npz')
    keep, (rows, cols), attrs = generate_edges(z['xyz'], z['bp'], float(z['inter']), float(z['intra']),
                                               prune=bool(z['prune']))
    assert np.array_equal(keep, z['keep'])
    assert np.array_equal(rows, z['rows']) and np.array_equal(cols, z['cols'])
    assert np.array_equal(attrs, z['attrs'])
