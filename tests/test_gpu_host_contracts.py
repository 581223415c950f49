"""Host-side contracts around the HIP path that earlier rounds left untested:
the per-layer parameter cache against copies and nested parameter swaps, the standalone segment operators' status word,
the 64-bit-offset instantiation of the edge forward, and graph preparation of more graphs than a grid's y extent."""
import copy
import pickle

import numpy as np
import pytest
import torch

from tests.test_gpu_properties import gpu_run, make_model, random_graph

pytestmark = pytest.mark.gpu


def _step(model, g):
    model.zero_grad()
    y, _, _, _ = model.unpack_input_data_and_predict(g)
    model.get_loss(torch.ones_like(y), y).backward()
    return y.detach().clone()


def test_model_copies_after_a_forward_and_nested_parameter_swaps():
    """EGNNLayer caches a ctypes struct of its parameters' addresses after the first forward. (1) copy.deepcopy and
    pickle of a model that has run must work (EMA / SWA snapshots) and the copy must run on ITS OWN tensors; (2) a
    parameter replaced INSIDE a leaf module - `layer.node_mlp[0].weight = ...`, `load_state_dict(assign=True)` - or a
    replaced leaf module must be picked up by the next forward: the new tensor is used and receives the gradient."""
    model, _ = make_model(seed=3, edge_attention=True, node_attention=True, residual=True)
    model.train()
    g = random_graph(200, 3000, seed=1).to('cuda')
    y0 = _step(model, g)
    layer = model.layers[1]
    assert layer.__dict__.get('_pcache') is not None

    twin = copy.deepcopy(model)
    assert twin.layers[1].__dict__.get('_pcache') is None
    blob = pickle.dumps(model.layers[1])
    assert pickle.loads(blob).__dict__.get('_pcache') is None
    with torch.no_grad():
        for p in model.parameters():
            p.mul_(0.5)                         # the original changes; the copy must not follow it
    y_twin = _step(twin, g)
    assert torch.equal(y_twin, y0)
    assert not torch.equal(_step(model, g), y0)
    assert all(p.grad is not None for n, p in twin.named_parameters() if 'layers.2.coord_mlp' not in n)

    # nested parameter assignment: the layer's own __setattr__ never sees it
    model, _ = make_model(seed=3)
    model.train()
    y0 = _step(model, g)
    lin = model.layers[1].node_mlp[0]
    old = lin.weight
    lin.weight = torch.nn.Parameter(old.detach().clone() * 1.5)
    y1 = _step(model, g)
    assert not torch.equal(y1, y0)
    assert lin.weight.grad is not None and float(lin.weight.grad.abs().max()) > 0
    # load_state_dict(assign=True) replaces every parameter object
    sd = {k: v.detach().clone() for k, v in make_model(seed=11)[0].state_dict().items()}
    model.load_state_dict(sd, assign=True)
    ref, _ = make_model(seed=11)
    ref.train()
    assert torch.equal(_step(model, g), _step(ref, g))
    # a replaced leaf module
    new_lin = torch.nn.Linear(32, 32).cuda()
    model.layers[2].edge_mlp[2] = new_lin
    _step(model, g)
    assert new_lin.weight.grad is not None and float(new_lin.weight.grad.abs().max()) > 0


def test_segment_operators_report_out_of_range_ids():
    """unsorted_segment_sum with an id == num_segments: the reference's scatter_add_ raises (egnn_satorras.py:336). The
    kernels stay in bounds (the forward sums such a row into segment 0, the backward reads segment 0's gradient) and
    the host raises IndexError - at the latest in the call's own backward, or at the next segment call."""
    from pointvs_amd import functional as PF
    from pointvs_amd.egnn_satorras import unsorted_segment_mean, unsorted_segment_sum
    data = torch.randn(1000, 5, device='cuda', requires_grad=True)
    ids = torch.randint(0, 40, (1000,), device='cuda')
    bad = ids.clone()
    bad[17] = 40
    out = unsorted_segment_sum(data, bad, 40)
    with pytest.raises(IndexError):
        out.sum().backward()
    PF.segment_status_check()                       # (settled: nothing pending)
    out = unsorted_segment_mean(data, bad, 40)      # never differentiated: the next call reports it
    torch.cuda.synchronize()
    with pytest.raises(IndexError):
        unsorted_segment_sum(data, ids, 40)
    bad[17] = -3
    unsorted_segment_sum(data.detach(), bad, 40)
    with pytest.raises(IndexError):
        PF.segment_status_check()
    # good ids: values and gradients as before, nothing raised
    data.grad = None
    out = unsorted_segment_sum(data, ids, 40)
    out.sum().backward()
    PF.segment_status_check()
    ref = torch.zeros(40, 5, device='cuda').index_add_(0, ids, data.detach())
    assert torch.allclose(out.detach(), ref, atol=1e-5)
    assert torch.equal(data.grad, torch.ones_like(data))


@pytest.mark.parametrize('changes', [dict(), dict(k=64, edge_attention=True), dict(k=128, softmax_attention=True, edge_attention=True)])
def test_forward_with_64_bit_offsets_equals_the_scalar_base_form(changes, monkeypatch):
    """The edge forward addresses node rows as scalar base + 32-bit lane offset; tables beyond 4 GB (N * 8H >= 2^32) or
    E >= 2^30 take the 64-bit instantiation of the same kernel (it used to refuse). PVS_FWD_SADDR=0 forces
    that instantiation: outputs and gradients bit-identical."""
    g = random_graph(700, 30000, seed=4, n_graphs=3).to('cuda')
    model, _ = make_model(seed=2, **changes)
    a = gpu_run(model, g)
    monkeypatch.setenv('PVS_FWD_SADDR', '0')
    b = gpu_run(model, g)
    for x, y in zip(a, b):
        if isinstance(x, dict):
            assert x.keys() == y.keys()
            for k in x:
                assert (x[k] is None and y[k] is None) or np.array_equal(x[k], y[k]), k
        else:
            assert np.array_equal(np.asarray(x), np.asarray(y))


@pytest.mark.parametrize('placement', ['2', '0'])
def test_graph_preparation_of_more_graphs_than_a_grid_has_rows(placement, monkeypatch):
    """70,000 graphs in the reference loader's layout: the per-graph colptr kernel used one grid row (blockIdx.y) per
    graph and a launch cannot have more than 65,535. Same arrays as the general sort."""
    from pointvs_amd.graph import Batch, prepare_graph, runs_layout
    from pointvs_amd.synthetic import synthetic_graph
    from tests.test_gpu_baseline_parity import _replicate_on_device
    monkeypatch.setenv('PVS_CSC_TILES', placement)
    items = [synthetic_graph(700 + k, n_nodes=6 + k % 5, n_lig=2, edge_radius=30.0) for k in range(70)]
    big = _replicate_on_device(Batch.from_data_list(items).to('cuda'), 1000)
    assert big.num_graphs == 70000
    layout = runs_layout(big)
    assert layout is not None
    n = int(big.x.shape[0])
    a = prepare_graph(big.edge_index, big.edge_attr, n, need_backward=True, layout=layout)
    b = prepare_graph(big.edge_index, big.edge_attr, n, need_backward=True)
    a.check_status(); b.check_status()
    for name in ('rowptr', 'row', 'col', 'etype', 'perm', 'colptr', 'cedge'):
        assert torch.equal(a.t[name], b.t[name]), name
