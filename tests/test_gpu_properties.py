"""GPU property and edge-case tests of the HIP path beyond the golden fixtures: ragged / degenerate
graphs against the CPU oracle, tile- and chunk-boundary sizes, and size-independent properties
(E(3) equivariance, edge-order invariance, batch = union of graphs) at BASELINE config sizes."""
import os
from pathlib import Path

import numpy as np
import pytest
import torch

from tests._golden import CaseLog, assert_strict, grad_floor, needs_caching_allocator, rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-5

BASE_KW = dict(dim_input=12, k=32, dim_output=1, num_layers=2, residual=False, edge_residual=False,
               edge_attention=False, normalize=False, tanh=False, dropout=0.0, graphnorm=False,
               update_coords=True, permutation_invariance=False, node_attention=False,
               gated_residual=False, rezero=False, softmax_attention=False,
               model_task='classification')


def make_model(seed=0, **changes):
    from pointvs_amd.egnn_satorras import SartorrasEGNN
    torch.manual_seed(seed)
    kw = dict(BASE_KW, **changes)
    model = SartorrasEGNN(Path('/tmp/pvs_prop'), 2e-3, 1e-4, silent=True, **kw)
    return model.cuda().eval(), kw


def random_graph(n, edges, seed, n_graphs=1):
    """edges: int (random pairs, duplicates allowed) or an explicit [2,E] array."""
    from pointvs_amd.graph import Batch
    rng = np.random.default_rng(seed)
    if isinstance(edges, int):
        ei = rng.integers(0, n, size=(2, edges))
    else:
        ei = np.asarray(edges)
    e = ei.shape[1]
    x = np.zeros((n, 12), dtype=np.float32)
    x[np.arange(n), rng.integers(0, 11, n)] = 1.0
    x[:, 11] = rng.integers(0, 2, n)
    batch = np.sort(rng.integers(0, n_graphs, n)) if n_graphs > 1 else np.zeros(n, dtype=np.int64)
    if n_graphs > 1:   # every graph id must occur; edges stay inside a graph
        batch = np.repeat(np.arange(n_graphs), int(np.ceil(n / n_graphs)))[:n]
        same = batch[ei[0]] == batch[ei[1]]
        ei = ei[:, same]
        e = ei.shape[1]
    et = rng.integers(0, 3, e)
    return Batch(
        x=torch.from_numpy(x), pos=torch.from_numpy(rng.normal(size=(n, 3)).astype(np.float32) * 3),
        edge_index=torch.from_numpy(ei.astype(np.int64)),
        edge_attr=torch.nn.functional.one_hot(torch.from_numpy(et), 3),
        batch=torch.from_numpy(batch.astype(np.int64)), y=torch.ones(n_graphs),
        lig_fname=['l'] * n_graphs, rec_fname=['r'] * n_graphs, num_graphs=n_graphs)


def oracle_run(model, kw, g, dtype=torch.float32):
    from oracle import egnn_oracle as orc
    sd = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    cfg = dict(kw, _class='SartorrasEGNN')
    return orc.forward_backward(sd, cfg, g.x, g.pos, g.edge_index, g.edge_attr, g.batch,
                                torch.ones(int(g.batch.max()) + 1), dtype=dtype)


def oracle_fp32_draws(model, kw, g, draws=3):
    """{parameter: [fp32 gradient of `draws` oracle evaluations]}: on a many-core host the oracle's CPU sums are threaded and
    every evaluation is ONE draw of the reference arithmetic's rounding noise; a strict bound formed from a single draw is a
    random variable (round 6: profiles/r06_fuzz_campaign.txt, the golden case that failed once in 25-60 fresh processes)."""
    runs = [oracle_run(model, kw, g, dtype=torch.float32)[2] for _ in range(draws)]
    return {k: (None if runs[0][k] is None else [r[k].numpy() for r in runs]) for k in runs[0]}


def gpu_run(model, g):
    import copy
    gg = copy.copy(g)
    gg.__dict__ = dict(g.__dict__)
    gg = gg.to('cuda')
    model.zero_grad()
    y = model(gg).reshape(-1)
    loss = model.get_loss(torch.ones_like(y), y)
    loss.backward()
    grads = {n: (None if p.grad is None else p.grad.detach().cpu().numpy())
             for n, p in model.named_parameters()}
    return y.detach().cpu().numpy(), grads


CASES = {
    'tile_31': (40, 31), 'tile_32': (40, 32), 'tile_33': (40, 33), 'tile_64': (40, 64),
    'tile_65': (40, 65), 'single_edge': (5, 1), 'isolated_nodes': (200, 150),
    'dense_small': (24, 24 * 23), 'medium': (700, 20000),
}


@pytest.mark.parametrize('name', sorted(CASES))
@pytest.mark.parametrize('flags', ['default', 'att_res'])
def test_ragged_graphs_match_oracle(name, flags):
    n, e = CASES[name]
    changes = {} if flags == 'default' else dict(
        edge_attention=True, node_attention=True, residual=True, normalize=True, tanh=True,
        edge_residual=True, num_layers=3)
    model, kw = make_model(seed=1, **changes)
    g = random_graph(n, e, seed=sum(name.encode()) % 1000)
    y, grads = gpu_run(model, g)
    y_ref, _, g_ref = oracle_run(model, kw, g, dtype=torch.float64)
    g_ref32 = oracle_fp32_draws(model, kw, g)
    assert rel_err(y, y_ref.numpy()) < TOL
    log = CaseLog(f'ragged_{name}_{flags}')
    floor = grad_floor({k: (None if v is None else v.numpy()) for k, v in g_ref.items()})
    for pname, gr in grads.items():
        if gr is None:
            assert g_ref[pname] is None, pname
        else:
            assert rel_err(gr, g_ref[pname].numpy()) < TOL, pname
            # (strict per-tensor form, tests/_golden.py: a small gradient tensor is compared at its own magnitude)
            assert_strict(gr, g_ref[pname].numpy(), g_ref32[pname], f'{log.case} grad {pname}', floor=floor, log=log)
    log.finish()


@pytest.mark.parametrize('k', [96, 128])
@pytest.mark.parametrize('flags', ['default', 'att_res'])
def test_wide_layers_run_decomposed_and_match_oracle(k, flags, monkeypatch):
    """Hidden sizes above 64 channels (the reference accepts any --channels) can run as the composition of the
    layer's public sub-methods on the prepared graph (EGNNLayer._decomposed_call; PVS_WIDE=decomposed selects it
    since round 3, when the fused 128-channel kernels became the default): same logits and gradients as the oracle,
    attention values and coordinates still available."""
    monkeypatch.setenv('PVS_WIDE', 'decomposed')
    _check_wide(k, flags)


@pytest.mark.parametrize('k,flags', [(160, 'att_res'), (256, 'default'), (200, 'softmax'), (300, 'rezero'),
                                     (1100, 'default')])
def test_layers_wider_than_the_fused_kernels_match_oracle(k, flags):
    """The reference accepts any --channels (parse_args.py:56). Above the fused kernels' 128 channels the layer runs
    as the composition of its public sub-methods: wide plain products on the library GEMM, the per-channel kernels
    (mean pool, column reductions, the one-output linears of the heads and gates) in chunks."""
    _check_wide(k, flags)


@pytest.mark.parametrize('k,flags', [(96, 'default'), (128, 'default'), (128, 'att_res'), (72, 'att_res'),
                                     (128, 'softmax'), (100, 'rezero')])
def test_wide_layers_run_fused_and_match_oracle(k, flags):
    """Round 3: 64 < hidden <= 128 on the fused kernels at 128 channels (zero-padded below 128): the f16x2 edge
    forward as two launches (one split weight matrix in LDS each, the messages handed over through memory) and the
    four-wave team backward on three-term fp16 products (edge_bwd_wide.hip) with coord_mlp.0's weight read from global
    memory. Oracle: fp64 autograd."""
    _check_wide(k, flags)


@pytest.mark.parametrize('k,flags', [(128, 'att_res'), (80, 'default')])
def test_wide_layers_fp32_family_matches_oracle(k, flags, monkeypatch):
    """PVS_EGNN_BF16X3=0 at 128 channels: the four-wave team backward on exact fp32 MFMAs (coord_mlp.0's weight read
    from global memory) - the arithmetic cross-check of the f16x2 team backward (the forward has no fp32 form at this
    width: two fp32 128x128 matrices do not fit in LDS)."""
    monkeypatch.setenv('PVS_EGNN_BF16X3', '0')
    _check_wide(k, flags)


def _check_wide(k, flags):
    changes = dict(k=k, num_layers=2) if flags == 'default' else dict(
        k=k, num_layers=2, edge_attention=True, node_attention=True, residual=True, normalize=True, tanh=True,
        edge_residual=True, graphnorm=True)
    if flags == 'softmax':
        changes = dict(k=k, num_layers=2, edge_attention=True, softmax_attention=True, node_attention=True, residual=True)
    if flags == 'rezero':
        changes = dict(k=k, num_layers=3, edge_residual=True, rezero=True, residual=True, edge_attention=True)
    model, kw = make_model(seed=3, **changes)
    g = random_graph(300, 6000, seed=21, n_graphs=3)
    y, grads = gpu_run(model, g)
    y_ref, _, g_ref = oracle_run(model, kw, g, dtype=torch.float64)
    g_ref32 = oracle_fp32_draws(model, kw, g)
    assert rel_err(y, y_ref.numpy()) < TOL
    log = CaseLog(f'wide_{k}_{flags}')
    floor = grad_floor({k: (None if v is None else v.numpy()) for k, v in g_ref.items()})
    for pname, gr in grads.items():
        if gr is None:
            assert g_ref[pname] is None, pname
        else:
            assert rel_err(gr, g_ref[pname].numpy()) < TOL, pname
            # (strict per-tensor form, tests/_golden.py: a small gradient tensor is compared at its own magnitude)
            assert_strict(gr, g_ref[pname].numpy(), g_ref32[pname], f'{log.case} grad {pname}', floor=floor, log=log)
    log.finish()
    if flags == 'att_res':
        n_edges = int(g.edge_index.shape[1])
        for layer in list(model.layers)[1:]:      # (the last one evaluates its dead coordinate update on demand)
            assert layer.att_val.shape == (n_edges, 1) and layer.node_att_val.shape == (300, 1)
            assert layer.intermediate_coords.shape == (300, 3)


def test_star_graph_rows_longer_than_a_chunk():
    """One destination with 20,000 incoming edges (a row far longer than a wave's chunk) plus the
    reverse edges: exercises multi-tile rows, chunk alignment and the column gather."""
    n = 20001
    hub = np.zeros(n - 1, dtype=np.int64)
    leaves = np.arange(1, n)
    ei = np.concatenate([np.stack([hub, leaves]), np.stack([leaves, hub])], axis=1)
    model, kw = make_model(seed=2, num_layers=2)
    g = random_graph(n, ei, seed=5)
    y, grads = gpu_run(model, g)
    y_ref, _, g_ref = oracle_run(model, kw, g, dtype=torch.float64)
    _, _, g_ref32 = oracle_run(model, kw, g, dtype=torch.float32)
    assert rel_err(y, y_ref.numpy()) < TOL
    # This case checks STRUCTURE (a row spanning hundreds of tiles and several chunks); a 20,000-term
    # fp32 sum of un-normalised activations is ill-conditioned (the CPU fp32 run itself is 2e-6 off
    # the fp64 one, a different summation tree lands at 2-3e-5), so the bound here is 1e-4: indexing
    # mistakes show up as O(1) errors.
    for pname, gr in grads.items():
        if gr is not None:
            ref64 = g_ref[pname].numpy()
            noise = rel_err(g_ref32[pname].numpy(), ref64)
            assert rel_err(gr, ref64) < max(1e-4, 2 * noise), pname


def test_mfma_and_generic_kernels_agree():
    """The two kernel families on the same inputs (PVS_EGNN_KERNELS=generic selects the generic)."""
    model, kw = make_model(seed=3, edge_attention=True, num_layers=3, residual=True)
    g = random_graph(1500, 60000, seed=9, n_graphs=3)
    os.environ.pop('PVS_EGNN_KERNELS', None)
    y_a, g_a = gpu_run(model, g)
    os.environ['PVS_EGNN_KERNELS'] = 'generic'
    try:
        y_b, g_b = gpu_run(model, g)
    finally:
        os.environ.pop('PVS_EGNN_KERNELS', None)
    assert rel_err(y_a, y_b) < TOL
    for pname in g_a:
        if g_a[pname] is not None:
            assert rel_err(g_a[pname], g_b[pname]) < TOL, pname


def _cfg2_graph(seed=2000):
    from pointvs_amd.graph import Batch
    from pointvs_amd.synthetic import CONFIGS, synthetic_graph
    return Batch.from_data_list([synthetic_graph(seed, **CONFIGS['cfg2']['graph'])])


def test_e3_equivariance_at_config_size():
    """Rotation + translation of a 2000-atom, r=10 A graph: logits and h invariant, x equivariant."""
    from pointvs_amd.graph import prepared_for
    from pointvs_amd.synthetic import CONFIGS
    model, _ = make_model(seed=4, **{k: v for k, v in CONFIGS['cfg2']['model'].items()
                                     if k in BASE_KW})
    g = _cfg2_graph().to('cuda')
    rng = np.random.default_rng(0)
    q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
    rot = torch.tensor(q, dtype=torch.float32, device='cuda')
    shift = torch.tensor([3.0, -2.0, 1.0], device='cuda')
    with torch.no_grad():
        feats, edges, coords, eattr, _ = model.unpack_graph(g)
        pg = prepared_for(edges, eattr, feats.size(0))
        h1, x1, _ = model.embed_prepared(pg, feats, coords)
        h2, x2, _ = model.embed_prepared(pg, feats, coords @ rot.T + shift)
    scale = max(1.0, float(h1.abs().max()))
    assert float((h1 - h2).abs().max()) < 2e-5 * scale   # fp32 noise of a rotated input
    assert float((x1 @ rot.T + shift - x2).abs().max()) < 1e-4


def test_edge_order_invariance_and_message_order():
    """Shuffling the COO changes nothing but the order of the returned edge messages."""
    model, _ = make_model(seed=5)
    g = _cfg2_graph(2001).to('cuda')
    perm = torch.randperm(g.edge_index.shape[1], device='cuda',
                          generator=torch.Generator('cuda').manual_seed(1))
    with torch.no_grad():
        feats, edges, coords, eattr, batch = model.unpack_graph(g)
        h1, m1 = model.get_embeddings(feats, edges, coords, eattr, batch)
        h2, m2 = model.get_embeddings(feats, edges[:, perm].contiguous(), coords,
                                      eattr[perm].contiguous(), batch)
    scale = max(1.0, float(h1.abs().max()))
    assert float((h1 - h2).abs().max()) < 1e-5 * scale
    assert float((m1[perm] - m2).abs().max()) < 1e-5 * max(1.0, float(m1.abs().max()))


def test_batch_equals_union_of_graphs():
    from pointvs_amd.graph import Batch
    from pointvs_amd.synthetic import synthetic_graph
    model, _ = make_model(seed=6)
    gs = [synthetic_graph(s, n_nodes=400, n_lig=20, edge_radius=7.0) for s in (11, 12, 13)]
    with torch.no_grad():
        y_batch = model(Batch.from_data_list(gs).to('cuda')).reshape(-1).cpu()
        y_single = torch.cat([model(Batch.from_data_list([gi]).to('cuda')).reshape(-1).cpu()
                              for gi in gs])
    assert float((y_batch - y_single).abs().max()) < 1e-5 * max(1.0, float(y_single.abs().max()))


def test_invalid_inputs_raise():
    from pointvs_amd.graph import prepare_graph
    ei = torch.tensor([[0, 1, 7], [1, 2, 0]], device='cuda')
    ea = torch.nn.functional.one_hot(torch.tensor([0, 1, 2]), 3).cuda()
    with pytest.raises(IndexError):
        prepare_graph(ei, ea, 3).check_status()
    bad = ea.clone()
    bad[1] = torch.tensor([1, 1, 0])
    with pytest.raises(ValueError):
        prepare_graph(torch.tensor([[0, 1, 2], [1, 2, 0]], device='cuda'), bad, 3).check_status()


@pytest.mark.parametrize('k', [8, 24, 40])
def test_hidden_sizes_between_built_widths_are_zero_padded(k):
    model, kw = make_model(seed=8, k=k, edge_attention=True, node_attention=True, residual=True,
                           graphnorm=True, normalize=True, tanh=True)
    g = random_graph(120, 1500, seed=3)
    y, grads = gpu_run(model, g)
    y_ref, _, g_ref = oracle_run(model, kw, g, dtype=torch.float64)
    assert rel_err(y, y_ref.numpy()) < TOL
    for pname, gr in grads.items():
        if gr is None:
            assert g_ref[pname] is None, pname
        else:
            assert gr.shape == tuple(g_ref[pname].shape)
            assert rel_err(gr, g_ref[pname].numpy()) < TOL, pname


def test_unsorted_segment_sum_and_mean():
    """Module-level functions of egnn_satorras.py:332-347 as HIP operators, forward and backward."""
    from oracle import egnn_oracle as orc
    from pointvs_amd.egnn_satorras import unsorted_segment_mean, unsorted_segment_sum
    rng = np.random.default_rng(0)
    data = torch.from_numpy(rng.normal(size=(5000, 7)).astype(np.float32))
    ids = torch.from_numpy(rng.integers(0, 300, 5000))
    ids[:50] = 7          # a long segment; some segments stay empty
    for fn, ref in ((unsorted_segment_sum, orc.segment_sum), (unsorted_segment_mean, orc.segment_mean)):
        d_gpu = data.cuda().requires_grad_(True)
        out = fn(d_gpu, ids.cuda(), 320)
        d_ref = data.double().requires_grad_(True)
        out_ref = ref(d_ref, ids, 320)
        assert rel_err(out.detach().cpu().numpy(), out_ref.detach().numpy()) < 1e-6
        w = torch.from_numpy(rng.normal(size=(320, 7)).astype(np.float32))
        (out * w.cuda()).sum().backward()
        (out_ref * w.double()).sum().backward()
        assert rel_err(d_gpu.grad.cpu().numpy(), d_ref.grad.numpy()) < 1e-6


@pytest.mark.parametrize('width,n_out', [(32, 1), (64, 3), (100, 1), (33, 2), (128, 128), (1024, 2)])
def test_pool_and_head_as_one_op(width, n_out):
    """feats_linear_layers(global_mean_pool(feats, batch)) (pnn_geometric_base.py:29-36) as PF.pool_head - one launch
    forward, one backward - against fp64 torch on CPU: ragged graphs, one empty graph, one single-node graph."""
    from pointvs_amd import functional as PF
    rng = np.random.default_rng(width + n_out)
    counts = np.array([37, 0, 1, 500, 2000, 3, 64])
    ptr = np.concatenate([[0], np.cumsum(counts)])
    n = int(ptr[-1])
    h = torch.from_numpy(rng.normal(size=(n, width)).astype(np.float32))
    w = torch.from_numpy((rng.normal(size=(n_out, width)) / np.sqrt(width)).astype(np.float32))
    b = torch.from_numpy(rng.normal(size=(n_out,)).astype(np.float32))
    up = torch.from_numpy(rng.normal(size=(len(counts), n_out)).astype(np.float32))
    for bias in (b, None):
        hg, wg = h.cuda().requires_grad_(True), w.cuda().requires_grad_(True)
        bg = None if bias is None else bias.cuda().requires_grad_(True)
        y = PF.pool_head(hg, torch.from_numpy(ptr).int().cuda(), wg, bg)
        (y * up.cuda()).sum().backward()
        hr, wr = h.double().requires_grad_(True), w.double().requires_grad_(True)
        br = None if bias is None else bias.double().requires_grad_(True)
        pooled = torch.stack([hr[ptr[g]:ptr[g + 1]].sum(0) / max(int(counts[g]), 1) for g in range(len(counts))])
        yr = torch.nn.functional.linear(pooled, wr, br)
        (yr * up.double()).sum().backward()
        assert rel_err(y.detach().cpu().numpy(), yr.detach().numpy()) < 1e-6
        assert rel_err(hg.grad.cpu().numpy(), hr.grad.numpy()) < 1e-6
        assert rel_err(wg.grad.cpu().numpy(), wr.grad.numpy()) < 1e-6
        if bias is not None:
            assert rel_err(bg.grad.cpu().numpy(), br.grad.numpy()) < 1e-6
        # the unfused pair gives the same values (same summation orders)
        h2, w2 = h.cuda().requires_grad_(True), w.cuda().requires_grad_(True)
        y2 = PF.linear(PF.mean_pool(h2, torch.from_numpy(ptr).int().cuda()), w2, None if bias is None else bias.cuda())
        assert rel_err(y.detach().cpu().numpy(), y2.detach().cpu().numpy()) < 1e-6


@pytest.mark.parametrize('n', [1, 32, 257, 5000])
def test_bce_with_logits_as_one_op(n):
    """nn.BCEWithLogitsLoss() (point_neural_network_base.py:74, :365) as PF.bce_with_logits_mean against torch in fp64,
    logits from -90 to 90 (both tails of the stable form), hard and soft targets, an upstream gradient other than 1."""
    from pointvs_amd import functional as PF
    rng = np.random.default_rng(n)
    x = torch.from_numpy(np.concatenate([rng.normal(size=n) * 3, [-90.0, 90.0, 0.0]])[:max(n, 1)].astype(np.float32))
    if n >= 3:
        x[:3] = torch.tensor([-90.0, 90.0, 0.0])
    t = torch.from_numpy((rng.random(x.numel()) < 0.5).astype(np.float32))
    t[::7] = 0.3
    xg = x.cuda().requires_grad_(True)
    loss = PF.bce_with_logits_mean(xg, t.cuda())
    (2.5 * loss).backward()
    xr = x.double().requires_grad_(True)
    loss_ref = torch.nn.BCEWithLogitsLoss()(xr, t.double())
    (2.5 * loss_ref).backward()
    assert abs(float(loss.detach()) - float(loss_ref.detach())) <= 1e-6 * max(1.0, abs(float(loss_ref.detach())))
    assert rel_err(xg.grad.cpu().numpy(), xr.grad.numpy()) < 1e-6
    # the 2-D form the models produce ([B, 1] logits against [B, 1] labels), and a shape mismatch
    l2 = PF.bce_with_logits_mean(x.cuda().reshape(-1, 1), t.cuda().reshape(-1, 1))
    assert float(l2) == float(loss.detach())
    with pytest.raises(ValueError):
        PF.bce_with_logits_mean(x.cuda().reshape(-1, 1), t.cuda())


@needs_caching_allocator
def test_hipgraph_captured_step_matches_eager():
    """The whole training step (prepare + forward + loss + backward + clip + Adam) captured in a
    hipGraph and replayed gives the same parameters as the eager step (the C ABI allocates nothing
    and never synchronises, so it is capturable)."""
    from pointvs_amd import graph as pgraph
    from pointvs_amd.graph import Batch
    from pointvs_amd.synthetic import synthetic_graph
    gs = [synthetic_graph(s, n_nodes=300, n_lig=20, edge_radius=6.0) for s in (21, 22, 23, 24)]
    batch = Batch.from_data_list(gs).to('cuda')
    y_true = batch.y.float()

    def run(captured):
        model, _ = make_model(seed=7, num_layers=2)
        model.train()
        params = list(model.parameters())
        model.optimiser = torch.optim.Adam(params, lr=2e-3, weight_decay=1e-4, capturable=True)

        def step():
            y = model(batch).reshape(-1)
            loss = model.get_loss(y_true, y)
            model.optimiser.zero_grad()
            loss.backward()
            torch.nn.utils.clip_grad_value_(params, 1.0)
            model.optimiser.step()
            return loss

        old = pgraph.CACHE_ENABLED
        pgraph.CACHE_ENABLED = False
        try:
            stream = torch.cuda.Stream()
            with torch.cuda.stream(stream):
                step()
                step()
                if captured:
                    torch.cuda.synchronize()
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, stream=stream):
                        step()
                    for _ in range(3):
                        g.replay()
                else:
                    for _ in range(3):
                        step()
            torch.cuda.synchronize()
        finally:
            pgraph.CACHE_ENABLED = old
        return {k: v.detach().cpu().numpy().copy() for k, v in model.state_dict().items()}

    eager, graphed = run(False), run(True)
    for k in eager:
        assert rel_err(graphed[k], eager[k]) < 1e-6, k


@pytest.mark.parametrize('flags', [
    dict(), dict(edge_residual=True, residual=True),
    dict(edge_residual=True, gated_residual=True, residual=True, edge_attention=True,
         node_attention=True, normalize=True, tanh=True, graphnorm=True),
    dict(edge_attention=True, softmax_attention=True, residual=True)])
def test_layer_public_api_input_gradients(flags):
    """EGNNLayer.forward called directly (reference signature): outputs in the caller's edge order and
    the gradients wrt h, coord and edge_messages against autograd on the fp64 oracle layer."""
    from oracle import egnn_oracle as orc
    from pointvs_amd.egnn_satorras import EGNNLayer
    torch.manual_seed(11)
    hid = 32
    layer = EGNNLayer(hid, hid, hid, edges_in_d=3, **flags).cuda()
    rng = np.random.default_rng(5)
    pairs = rng.integers(0, 90, size=(2, 1500))
    g = random_graph(90, pairs[:, pairs[0] != pairs[1]], seed=21)   # no self loops: with normalize=True
    # a zero-length edge has d(diff/(|diff|+1e-8))/dx = 1e8, which cancels only in exact arithmetic
    h0 = torch.from_numpy(rng.normal(size=(90, hid)).astype(np.float32))
    m0 = torch.from_numpy(rng.normal(size=(g.edge_index.shape[1], hid)).astype(np.float32))
    wh = torch.from_numpy(rng.normal(size=(90, hid)).astype(np.float32))
    wx = torch.from_numpy(rng.normal(size=(90, 3)).astype(np.float32))
    wm = torch.from_numpy(rng.normal(size=m0.shape).astype(np.float32))

    h = h0.cuda().requires_grad_(True)
    x = g.pos.cuda().requires_grad_(True)
    mm = m0.cuda().requires_grad_(True)
    h1, x1, ea, m1 = layer(h, g.edge_index.cuda(), x, g.edge_attr.cuda(), mm)
    assert ea is not None and m1.shape == m0.shape
    ((h1 * wh.cuda()).sum() + (x1 * wx.cuda()).sum() + (m1 * wm.cuda()).sum()).backward()

    sd = {'L.' + k: v.detach().cpu().double().requires_grad_(True) for k, v in layer.state_dict().items()}
    kw = dict(orc.BUILD_NET_DEFAULTS, residual=True, normalize=False, tanh=False, graphnorm=False)  # EGNNLayer ctor defaults
    kw.update(flags)
    kw['edge_attention_here'] = kw['edge_attention']
    kw['node_attention_here'] = kw['node_attention']
    hr = h0.double().requires_grad_(True)
    xr = g.pos.double().requires_grad_(True)
    mr = m0.double().requires_grad_(True)
    h2, x2, m2, _, _ = orc.egnn_layer(sd, 'L.', kw, hr, g.edge_index, xr, g.edge_attr, mr)
    ((h2 * wh.double()).sum() + (x2 * wx.double()).sum() + (m2 * wm.double()).sum()).backward()
    assert rel_err(h1.detach().cpu().numpy(), h2.detach().numpy()) < TOL
    assert rel_err(x1.detach().cpu().numpy(), x2.detach().numpy()) < TOL
    assert rel_err(m1.detach().cpu().numpy(), m2.detach().numpy()) < TOL
    assert rel_err(h.grad.cpu().numpy(), hr.grad.numpy()) < TOL
    assert rel_err(x.grad.cpu().numpy(), xr.grad.numpy()) < TOL
    if flags.get('edge_residual'):
        assert rel_err(mm.grad.cpu().numpy(), mr.grad.numpy()) < TOL
    for name, p in layer.named_parameters():
        ref = sd['L.' + name].grad
        assert ref is not None and p.grad is not None, name
        assert rel_err(p.grad.cpu().numpy(), ref.numpy()) < TOL, name


def test_checkpoint_roundtrip_and_legacy_keys(tmp_path):
    """save() writes the reference's checkpoint dict; load_weights() reads it back, including the
    reference's legacy key names (point_neural_network_base.py:501-565)."""
    import yaml
    from pointvs_amd.egnn_satorras import SartorrasEGNN
    kw = dict(BASE_KW, edge_attention=True, node_attention=True)
    torch.manual_seed(1)
    a = SartorrasEGNN(tmp_path / 'run', 2e-3, 1e-4, **kw)
    a.save()
    ckpt = tmp_path / 'run' / 'checkpoints' / 'pose_ckpt_epoch_0.pt'
    assert ckpt.exists() and (tmp_path / 'run' / 'model_kwargs.yaml').exists()
    assert yaml.safe_load((tmp_path / 'run' / 'model_kwargs.yaml').read_text())['k'] == 32
    blob = torch.load(ckpt, map_location='cpu')
    assert set(blob) == {'learning_rate', 'weight_decay', 'p_epoch', 'a_epoch', 'model_state_dict',
                         'optimiser_state_dict'}
    torch.manual_seed(2)
    b = SartorrasEGNN(tmp_path / 'other', 2e-3, 1e-4, silent=True, **kw)
    b.load_weights(ckpt)
    for (k1, v1), (k2, v2) in zip(a.state_dict().items(), b.state_dict().items()):
        assert k1 == k2 and torch.equal(v1.cpu(), v2.cpu())
    legacy = dict(blob)
    legacy['model_state_dict'] = {k.replace('node_att_mlp', 'node_attention_mlp').replace(
        'att_mlp.0', 'att_mlp.2') if 'node_att' not in k else k.replace('node_att_mlp', 'node_attention_mlp'): v
        for k, v in blob['model_state_dict'].items()}
    legacy['model_state_dict'] = {k.replace('layers.1.att_mlp', 'layers.1.edge_attention_mlp'): v
                                  for k, v in legacy['model_state_dict'].items()}
    torch.save(legacy, tmp_path / 'legacy.pt')
    c = SartorrasEGNN(tmp_path / 'third', 2e-3, 1e-4, silent=True, **kw)
    c.load_weights(tmp_path / 'legacy.pt')
    for (k1, v1), (k3, v3) in zip(a.state_dict().items(), c.state_dict().items()):
        assert torch.equal(v1.cpu(), v3.cpu()), k1


def test_train_and_val_entry_points(tmp_path):
    """train_model / val keep the reference's calling convention on a tiny synthetic loader."""
    from pointvs_amd.egnn_multitask import MultitaskSatorrasEGNN
    from pointvs_amd.graph import Batch
    from pointvs_amd.synthetic import synthetic_graph
    gs = [synthetic_graph(s, n_nodes=120, n_lig=10, edge_radius=6.0) for s in range(30, 36)]
    loader = [Batch.from_data_list(gs[:3]), Batch.from_data_list(gs[3:])]
    torch.manual_seed(3)
    model = MultitaskSatorrasEGNN(tmp_path / 'm', 2e-3, 1e-4, **dict(BASE_KW, residual=True))
    before = model.layers[1].edge_mlp[0].weight.detach().clone()
    losses = model.train_model(loader, epochs=2)
    assert len(losses) == 4 and all(np.isfinite(losses))
    assert not torch.equal(before, model.layers[1].edge_mlp[0].weight.detach())
    assert model.val(loader, predictions_file=tmp_path / 'pred.txt') is True
    text = (tmp_path / 'pose_pred.txt').read_text()      # <task>_<name>, point_neural_network_base.py:223-224
    assert text.count('\n') == 6
    with torch.no_grad():                                  # the file holds sigmoid(model(batch)), in order
        want = torch.cat([torch.sigmoid(model(b.to('cuda')).reshape(-1)) for b in loader]).cpu().tolist()
    for line, g, p in zip(text.splitlines(), gs, want):
        assert line == f'{int(g.y):.3f} | {p:.3f} {g.rec_fname} {g.lig_fname}'
    assert (tmp_path / 'm' / 'checkpoints' / 'pose_ckpt_epoch_2.pt').exists()


def test_point_vs_entry_runs_the_readme_sequence_on_synthetic_graphs(tmp_path):
    """`point_vs.py multitask <dir> --model_task both -ea 1 -ep 1 --layers 3` (README.md:56-65) on
    synthetic graphs: pose training -> pose validation -> affinity training -> affinity validation,
    with the reference's records in save_path (point_vs.py:85-86, 258-275)."""
    import importlib.util
    import yaml
    root = Path(__file__).resolve().parent.parent
    spec = importlib.util.spec_from_file_location('pvs_entry', root / 'point_vs.py')
    entry = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(entry)
    model = entry.main(['multitask', str(tmp_path / 'run'), '--model_task', 'both', '-ea', '1', '-ep', '1',
                        '--layers', '3', '--synthetic_graphs', '12', '--synthetic_atoms', '150', '-b', '4',
                        '--edge_radius', '6', '--end_flag'])
    run = tmp_path / 'run'
    assert yaml.safe_load((run / 'cmd_args.yaml').read_text())['layers'] == 3
    assert yaml.safe_load((run / 'model_kwargs.yaml').read_text())['model_task'] == 'classification'
    assert (run / 'checkpoints' / 'pose_ckpt_epoch_1.pt').exists()
    assert (run / 'checkpoints' / 'affinity_ckpt_epoch_1.pt').exists()
    assert (run / 'pose_predictions.txt').read_text().count('\n') == 12
    assert (run / 'affinity_predictions.txt').read_text().count('\n') == 12
    assert (run / '_FINISHED').exists()
    assert model.model_task == 'regression' and model.p_epoch == 1 and model.a_epoch == 1


@pytest.mark.parametrize('seed', range(8))
def test_kernel_families_agree_on_random_configurations(seed):
    """Fuzz: random layer flags, hidden size, graph shape (isolated nodes, E not a multiple of the
    tile, several graphs) - the default MFMA kernels (split fp16 / bf16 products; node-level launches folded), the
    exact-fp32-MFMA family (PVS_EGNN_BF16X3=0), the default kernels with every node-level launch apart
    (PVS_EGNN_SPLIT_SMALL=1, PVS_FUSED_HEAD=0) and the generic kernels give the same outputs and gradients."""
    rng = np.random.default_rng(1000 + seed)
    flags = dict(
        k=int(rng.choice([32, 64])), num_layers=int(rng.integers(1, 4)),
        residual=bool(rng.integers(2)), edge_residual=bool(rng.integers(2)),
        edge_attention=bool(rng.integers(2)), node_attention=bool(rng.integers(2)),
        normalize=bool(rng.integers(2)), tanh=bool(rng.integers(2)), graphnorm=bool(rng.integers(2)),
        update_coords=bool(rng.integers(4) > 0), permutation_invariance=bool(rng.integers(4) == 0),
        attention_activation_fn=str(rng.choice(['sigmoid', 'tanh', 'relu', 'silu'])))
    variant = int(rng.integers(3))
    if flags['edge_attention'] and seed % 3 == 0:
        flags['softmax_attention'] = True
    if variant == 1:
        flags['gated_residual'] = True
    elif variant == 2:
        flags['rezero'] = True
    model, _ = make_model(seed=seed, **flags)
    n = int(rng.integers(40, 2500))
    e = int(rng.integers(1, 40)) * n + int(rng.integers(0, 31))
    g = random_graph(n, e, seed=seed, n_graphs=int(rng.integers(1, 5)))
    runs = {}
    for name, env in (('mfma', {}), ('fp32', {'PVS_EGNN_BF16X3': '0'}), ('generic', {'PVS_EGNN_KERNELS': 'generic'}),
                      ('split', {'PVS_EGNN_SPLIT_SMALL': '1', 'PVS_FUSED_HEAD': '0'})):
        for k_ in ('PVS_EGNN_BF16X3', 'PVS_EGNN_KERNELS', 'PVS_EGNN_SPLIT_SMALL', 'PVS_FUSED_HEAD'):
            os.environ.pop(k_, None)
        os.environ.update(env)
        try:
            runs[name] = gpu_run(model, g)
        finally:
            for k_ in env:
                os.environ.pop(k_, None)
    y_ref, g_ref = runs['generic']
    for name in ('mfma', 'fp32', 'split'):      # ('split': the node-level launches of a layer one by one, as in round 2)
        y, grads = runs[name]
        assert rel_err(y, y_ref) < TOL, (name, flags)
        for pname in g_ref:
            assert (g_ref[pname] is None) == (grads[pname] is None), (name, pname)
            if g_ref[pname] is not None:
                assert rel_err(grads[pname], g_ref[pname]) < 3 * TOL, (name, pname, flags)


@pytest.mark.parametrize('changes', [dict(), dict(k=64), dict(residual=True), dict(k=64, residual=True, node_attention=True,
                                                                           edge_attention=True)])
def test_folded_node_level_launches_equal_the_separate_ones(changes, monkeypatch):
    """Round 3 folded the node-level launches of a layer (P|Q as one product carrying the forward's clears, the node
    MLP as one kernel each way, no output-stage launches, the gate's weight gradients inside the weight-gradient pass).
    The products keep their MFMA order, so plain layers give the SAME BITS either way; with a node gate the row dot
    products are summed in another order (1e-6). The head + loss as fused ops change summation orders too."""
    model, _ = make_model(seed=5, num_layers=3, **changes)
    g = random_graph(1500, 40000, seed=9, n_graphs=3)
    y_a, g_a = gpu_run(model, g)
    monkeypatch.setenv('PVS_EGNN_SPLIT_SMALL', '1')
    y_b, g_b = gpu_run(model, g)
    gated = changes.get('node_attention', False)
    for name, a, b in [('logits', y_a, y_b)] + [(n, g_a[n], g_b[n]) for n in g_a]:
        assert (a is None) == (b is None), name
        if a is None:
            continue
        if gated:
            assert rel_err(a, b) < 1e-6, name
        else:
            assert np.array_equal(np.asarray(a), np.asarray(b)), name
    # and the head + loss as separate ops (other summation orders)
    monkeypatch.setenv('PVS_FUSED_HEAD', '0')
    y_c, g_c = gpu_run(model, g)
    assert rel_err(y_a, y_c) < 1e-6
    for n in g_a:
        if g_a[n] is not None:
            assert rel_err(g_a[n], g_c[n]) < 1e-5, n


def test_fused_clip_adam_matches_torch_adam():
    """pvs_adam_clip_step (one launch) vs clip_grad_value_ + torch.optim.Adam over several steps,
    including a parameter that never receives a gradient and one that starts late."""
    from pointvs_amd.optim import FusedClipAdam
    torch.manual_seed(0)
    shapes = [(32, 68), (32,), (1, 32), (64, 64), (3,)]
    a = [torch.nn.Parameter(torch.randn(s, device='cuda')) for s in shapes]
    b = [torch.nn.Parameter(p.detach().clone()) for p in a]
    oa = FusedClipAdam(a, lr=2e-3, weight_decay=1e-4)
    ob = torch.optim.Adam(b, lr=2e-3, weight_decay=1e-4)
    for step in range(5):
        for k, (pa, pb) in enumerate(zip(a, b)):
            if k == 4 or (k == 2 and step < 2):      # never / late
                pa.grad = pb.grad = None
                continue
            g = torch.randn(pa.shape, generator=torch.Generator().manual_seed(10 * step + k)).cuda() * 2
            pa.grad, pb.grad = g.clone(), g.clone()
        oa.step(clip_value=1.0)
        torch.nn.utils.clip_grad_value_(b, 1.0)
        ob.step()
    for pa, pb in zip(a, b):
        assert rel_err(pa.detach().cpu().numpy(), pb.detach().cpu().numpy()) < 1e-6
    for pa, pb in zip(a, b):
        if pa.grad is not None:
            assert torch.equal(pa.grad, pb.grad)     # clipped in place, like clip_grad_value_
    sa, sb = oa.state_dict()['state'], ob.state_dict()['state']
    assert set(sa) == set(sb)
    for k in sa:
        assert float(sa[k]['step']) == float(sb[k]['step'])
        assert rel_err(sa[k]['exp_avg_sq'].cpu().numpy(), sb[k]['exp_avg_sq'].cpu().numpy()) < 1e-6


@needs_caching_allocator
def test_capturable_fused_clip_adam_matches_the_host_counted_form_eagerly_and_replayed():
    """FusedClipAdam(capturable=True): step counters on the device as torch's capturable Adam keeps them (the same
    state_dict), the bias corrections formed in the kernel (pvs_adam_clip_step_dev). Six steps - three eager, then the
    step captured in a hipGraph and replayed three times on new gradients written into the same tensors - must leave
    the parameters, moments and counters of the host-counted fused form (bit for bit: the same arithmetic on the same
    correction factors) and of clip_grad_value_ + torch.optim.Adam(capturable=True)."""
    from pointvs_amd.optim import FusedClipAdam
    torch.manual_seed(0)
    shapes = [(32, 68), (32,), (1, 32), (64, 64)]
    a = [torch.nn.Parameter(torch.randn(s, device='cuda')) for s in shapes]
    b = [torch.nn.Parameter(p.detach().clone()) for p in a]
    c = [torch.nn.Parameter(p.detach().clone()) for p in a]
    oa = FusedClipAdam(a, lr=2e-3, weight_decay=1e-4, capturable=True)
    ob = FusedClipAdam(b, lr=2e-3, weight_decay=1e-4)
    oc = torch.optim.Adam(c, lr=2e-3, weight_decay=1e-4, capturable=True)
    grads = [[torch.randn(s, generator=torch.Generator().manual_seed(10 * t + k)).cuda() * 2 for k, s in enumerate(shapes)]
             for t in range(6)]
    for p in a + b + c:
        p.grad = torch.zeros_like(p)
    stream = torch.cuda.Stream()
    hip_graph = None
    with torch.cuda.stream(stream):
        for t in range(6):
            for k in range(len(shapes)):
                for ps in (a, b, c):
                    ps[k].grad.copy_(grads[t][k])
            if t < 3:
                oa.step(clip_value=1.0)
            elif hip_graph is None:
                stream.synchronize()
                hip_graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(hip_graph, stream=stream):
                    oa.step(clip_value=1.0)
                hip_graph.replay()               # (capturing does not execute)
            else:
                hip_graph.replay()
            ob.step(clip_value=1.0)
            torch.nn.utils.clip_grad_value_(c, 1.0)
            oc.step()
    torch.cuda.synchronize()
    assert oa._fast['fusable'] and oa._fast['groups'][0][0]['on_device']
    sa, sb, sc = oa.state_dict()['state'], ob.state_dict()['state'], oc.state_dict()['state']
    for k, (pa, pb, pc) in enumerate(zip(a, b, c)):
        assert torch.equal(pa.detach(), pb.detach()), k
        assert rel_err(pa.detach().cpu().numpy(), pc.detach().cpu().numpy()) < 1e-6
        assert torch.equal(sa[k]['exp_avg'], sb[k]['exp_avg']) and torch.equal(sa[k]['exp_avg_sq'], sb[k]['exp_avg_sq'])
        assert sa[k]['step'].is_cuda and sa[k]['step'].dtype == torch.float32 and float(sa[k]['step']) == 6.0
        assert float(sc[k]['step']) == 6.0 and not sb[k]['step'].is_cuda
    # the state moves between the flavours: torch's capturable Adam continues from the fused optimiser's checkpoint
    od = torch.optim.Adam([torch.nn.Parameter(p.detach().clone()) for p in a], lr=2e-3, weight_decay=1e-4, capturable=True)
    od.load_state_dict(oa.state_dict())
    assert float(od.state_dict()['state'][0]['step']) == 6.0


@pytest.mark.parametrize('config', ['cfg2', 'cfg3'])
def test_kernel_families_agree_at_baseline_batch_size(config):
    """BASELINE configs 2 and 3 at their full per-GPU batch of 32 graphs: MFMA kernels vs generic kernels,
    outputs and every gradient."""
    from pointvs_amd.synthetic import CONFIGS, synthetic_batch
    cfg = CONFIGS[config]
    model, _ = make_model(seed=5, **{k: v for k, v in cfg['model'].items() if k in BASE_KW})
    g = synthetic_batch(cfg['cfg_id'], 32, **cfg['graph'])
    os.environ.pop('PVS_EGNN_KERNELS', None)
    y_a, g_a = gpu_run(model, g)
    os.environ['PVS_EGNN_KERNELS'] = 'generic'
    try:
        y_b, g_b = gpu_run(model, g)
    finally:
        os.environ.pop('PVS_EGNN_KERNELS', None)
    assert rel_err(y_a, y_b) < TOL
    for pname in g_a:
        assert (g_a[pname] is None) == (g_b[pname] is None)
        if g_a[pname] is not None:
            assert rel_err(g_a[pname], g_b[pname]) < 2 * TOL, pname


def test_softmax_attention_rows_spanning_tiles_and_chunks():
    """Softmax edge attention on the MFMA path at the BASELINE graph shape (rows of ~160 edges span
    several 32-edge tiles: the online rescaling) and on a star graph (a 20,000-edge row spans many
    chunk-sized pieces of one wave's work): equal to the generic kernels' per-row softmax, and the
    attention weights of every row sum to one (the property the reference's test_attention checks)."""
    from pointvs_amd.graph import prepared_for
    from pointvs_amd.synthetic import CONFIGS, synthetic_batch
    cfg = CONFIGS['cfg2']
    kw = dict(edge_attention=True, softmax_attention=True, node_attention=True, residual=True, num_layers=2)
    model, _ = make_model(seed=9, **kw)
    n = 20001
    hub, leaves = np.zeros(n - 1, dtype=np.int64), np.arange(1, n)
    star = random_graph(n, np.concatenate([np.stack([hub, leaves]), np.stack([leaves, hub])], axis=1), seed=6)
    for g in (synthetic_batch(cfg['cfg_id'], 2, **cfg['graph']), star):
        os.environ.pop('PVS_EGNN_KERNELS', None)
        y_a, g_a = gpu_run(model, g)
        layer = model.layers[1]
        att = np.asarray(layer.att_val).reshape(-1)
        rows = np.asarray(g.edge_index[0])
        sums = np.bincount(rows, weights=att.astype(np.float64), minlength=int(g.x.shape[0]))
        has_edges = np.bincount(rows, minlength=int(g.x.shape[0])) > 0
        assert np.abs(sums[has_edges] - 1.0).max() < 1e-4
        os.environ['PVS_EGNN_KERNELS'] = 'generic'
        try:
            y_b, g_b = gpu_run(model, g)
        finally:
            os.environ.pop('PVS_EGNN_KERNELS', None)
        assert rel_err(y_a, y_b) < TOL
        for pname in g_a:
            if g_a[pname] is not None:
                assert rel_err(g_a[pname], g_b[pname]) < 1e-4, pname


def test_last_layer_coordinates_are_available_on_demand():
    """The model forward skips the last layer's coordinate branch (nothing reads x_L); the layer's
    `intermediate_coords` side attribute (egnn_satorras.py:175) must still give the reference value."""
    from pointvs_amd.graph import prepared_for
    model, _ = make_model(seed=7, num_layers=3, tanh=True, residual=True)
    g = random_graph(300, 6000, seed=3, n_graphs=2).to('cuda')
    with torch.no_grad():
        y_skip = model(g).reshape(-1).clone()
        lazy = model.layers[-1].intermediate_coords
        feats, edges, coords, eattr, _ = model.unpack_graph(g)
        pg = prepared_for(edges, eattr, feats.size(0))
        trace = {}
        model.embed_prepared(pg, feats, coords, trace=trace)          # full evaluation
        os.environ['PVS_EGNN_KEEP_DEAD_COORDS'] = '1'
        try:
            y_full = model(g).reshape(-1).clone()
        finally:
            os.environ.pop('PVS_EGNN_KEEP_DEAD_COORDS')
    assert torch.equal(y_skip, y_full)
    assert np.array_equal(lazy, trace['x3'].cpu().numpy())
    assert not np.array_equal(lazy, trace['x2'].cpu().numpy())


def test_backward_on_a_forward_only_graph_fails_loudly():
    """A graph prepared without the by-column lists (forward-only) must make the backward raise, not
    read NULL pointers."""
    from pointvs_amd.egnn_satorras import EGNNLayer
    from pointvs_amd.graph import prepare_graph
    torch.manual_seed(0)
    layer = EGNNLayer(32, 32, 32, edges_in_d=3).cuda()
    g = random_graph(60, 700, seed=1).to('cuda')
    pg = prepare_graph(g.edge_index, g.edge_attr, 60, need_backward=False)
    h = torch.randn(60, 32, device='cuda', requires_grad=True)
    h_out, _, _ = layer.forward_prepared(pg, h, g.pos)
    with pytest.raises(RuntimeError, match='by-column'):
        h_out.sum().backward()


@pytest.mark.parametrize('changes', [dict(), dict(edge_attention=True, node_attention=True, residual=True),
                                     dict(k=64, edge_attention=True), dict(edge_residual=True, tanh=True),
                                     dict(k=64, edge_residual=True, edge_attention=True, tanh=True),
                                     dict(edge_residual=True, edge_attention=True),
                                     dict(k=128), dict(k=100, edge_residual=True, edge_attention=True, node_attention=True)])
def test_bitwise_reproducible_at_baseline_size(changes):
    """No atomics and no unordered LDS hand-offs anywhere: four runs of the same cfg2-shaped batch give
    identical bits in the outputs and in every gradient (every kernel family: the H = 32 f16x2 backward with and
    without edge residual / attention, the H = 64 one-wave-per-16-edge-tile backward with and without edge residual,
    the four-wave team backward of the wide layers - eleven workgroup barriers and shared LDS images per tile). A race
    in a kernel shows up here as run-to-run differences long before it breaks a tolerance."""
    from pointvs_amd.synthetic import CONFIGS, synthetic_batch
    cfg = CONFIGS['cfg2']
    model, _ = make_model(seed=11, **dict({k: v for k, v in cfg['model'].items() if k in BASE_KW}, **changes))
    g = synthetic_batch(cfg['cfg_id'], 2 if changes.get('k', 32) > 64 else 8, **cfg['graph'])
    # (every product of the step, the head's included, is one of this library's kernels - no rocBLAS, no
    # atomics: profiles/r02_bench_cfg2_kernel_stats.csv lists every kernel of a step)
    runs = [gpu_run(model, g) for _ in range(4)]
    for y, grads in runs[1:]:
        assert y.tobytes() == runs[0][0].tobytes()
        for name, gr in grads.items():
            if gr is not None:
                assert gr.tobytes() == runs[0][1][name].tobytes(), name


@pytest.mark.parametrize('flags', [dict(), dict(edge_attention=True, node_attention=True, residual=True, tanh=True,
                                                normalize=True),
                                   dict(edge_attention=True, softmax_attention=True, graphnorm=True,
                                        permutation_invariance=True, gated_residual=True, residual=True)])
def test_public_submethods_compose_to_the_fused_layer(flags):
    """EGNNLayer.coord2radial / edge_model / coord_model / node_model (egnn_satorras.py:123-187) chained the
    way the reference's forward chains them (:189-206) give the fused layer's outputs and input gradients."""
    from pointvs_amd.egnn_satorras import EGNNLayer
    torch.manual_seed(4)
    layer = EGNNLayer(32, 32, 32, edges_in_d=3, **flags).cuda()
    g = random_graph(150, 2500, seed=8).to('cuda')
    h0 = torch.randn(150, 32, device='cuda')
    outs = []
    for fused in (True, False):
        h = h0.clone().requires_grad_(True)
        x = g.pos.clone().requires_grad_(True)
        if fused:
            h_out, x_out, _, m = layer(h, g.edge_index, x, g.edge_attr)
        else:
            row, col = g.edge_index
            radial, diff = layer.coord2radial(g.edge_index, x)
            m = layer.edge_model(h[row], h[col], radial, g.edge_attr)
            x_out = layer.coord_model(x, g.edge_index, diff, m)
            h_out, _ = layer.node_model(h, g.edge_index, m)
        (h_out.square().sum() + x_out.square().sum() + m.sum()).backward()
        outs.append([t.detach().cpu().numpy() for t in (h_out, x_out, m, h.grad, x.grad)])
    for a, b, name in zip(outs[0], outs[1], ('h', 'x', 'm', 'g_h', 'g_x')):
        assert rel_err(a, b) < 2e-5, name


def test_prefetched_graph_is_picked_up_by_the_next_forward():
    """graph.prefetch_graph (the next batch's CSR/CSC build on a side stream, a data loader's look-ahead) hands
    the same arrays to the next prepared_for call with those tensors - for plain and for generate_edges-tagged
    batches - and the model's numbers do not change."""
    from pointvs_amd import graph as pgraph
    from pointvs_amd.graph import Batch
    from pointvs_amd.synthetic import synthetic_graph
    items = [synthetic_graph(900 + k, n_nodes=n, n_lig=10, edge_radius=6.0) for k, n in enumerate((200, 350))]
    batch = Batch.from_data_list(items).to('cuda')
    n = int(batch.x.shape[0])
    model, _ = make_model(seed=4, num_layers=2)
    y_plain, g_plain = gpu_run(model, batch)
    was = pgraph.CACHE_ENABLED
    pgraph.CACHE_ENABLED = False
    try:
        for layout in (None, pgraph.runs_layout(batch)):
            ref = pgraph.prepare_graph(batch.edge_index, batch.edge_attr, n, need_backward=True, layout=layout)
            pgraph.prefetch_graph(batch.edge_index, batch.edge_attr, n, layout=layout)
            got = pgraph.prepared_for(batch.edge_index, batch.edge_attr, n, layout=layout)
            torch.cuda.synchronize()
            for name in ('rowptr', 'row', 'col', 'etype', 'perm', 'colptr', 'cedge', 'inv_deg'):
                assert torch.equal(ref.t[name], got.t[name]), name
        pgraph.prefetch_graph(batch.edge_index, batch.edge_attr, n, layout=pgraph.runs_layout(batch))
        model.zero_grad()
        y = model(batch).reshape(-1)
        assert y.detach().cpu().numpy().tobytes() == y_plain.tobytes()
    finally:
        pgraph.CACHE_ENABLED = was


def test_prepare_by_merging_sorted_runs_equals_the_sort():
    """pvs_graph_prepare_runs (batches tagged edge_layout == 'generate_edges': two row-sorted runs per graph,
    merged by counting) gives array for array what the radix-sort path gives - on ragged batches, with and
    without the by-column lists - and a list that breaks the promised layout raises instead of mis-sorting."""
    from pointvs_amd.graph import Batch, prepare_graph, runs_layout
    from pointvs_amd.synthetic import synthetic_graph
    sizes = [(300, 20, 6.0), (57, 5, 5.0), (800, 30, 7.0), (33, 32, 9.0), (120, 1, 4.0)]
    items = [synthetic_graph(700 + k, n_nodes=n, n_lig=nl, edge_radius=r) for k, (n, nl, r) in enumerate(sizes)]
    batch = Batch.from_data_list(items).to('cuda')
    assert batch.edge_layout == 'generate_edges'
    layout = runs_layout(batch)
    n = int(batch.x.shape[0])
    for need_backward in (True, False):
        a = prepare_graph(batch.edge_index, batch.edge_attr, n, need_backward=need_backward, layout=layout)
        b = prepare_graph(batch.edge_index, batch.edge_attr, n, need_backward=need_backward)
        a.check_status(); b.check_status()
        for name in ('rowptr', 'row', 'col', 'etype', 'perm', 'inv_deg') + (('colptr', 'cedge') if need_backward else ()):
            assert torch.equal(a.t[name], b.t[name]), name
    # the model-level path picks the merge up from the tag and gives the same numbers as the sort
    model, _ = make_model(seed=2, num_layers=2)
    y_runs, g_runs = gpu_run(model, batch)
    os.environ['PVS_PREPARE_RUNS'] = '0'
    try:
        plain = Batch.from_data_list(items)
        y_sort, g_sort = gpu_run(model, plain)
    finally:
        os.environ.pop('PVS_PREPARE_RUNS')
    assert y_runs.tobytes() == y_sort.tobytes()
    for name in g_runs:
        if g_runs[name] is not None:
            assert g_runs[name].tobytes() == g_sort[name].tobytes(), name
    # a broken promise: edges of one graph shuffled
    bad = Batch.from_data_list(items).to('cuda')
    e0 = bad.graph_edge_counts[0]
    perm = torch.randperm(e0, generator=torch.Generator().manual_seed(0)).cuda()
    bad.edge_index[:, :e0] = bad.edge_index[:, :e0][:, perm]
    bad.edge_attr[:e0] = bad.edge_attr[:e0][perm]
    pg = prepare_graph(bad.edge_index, bad.edge_attr, n, layout=runs_layout(bad))
    with pytest.raises(ValueError, match='generate_edges'):
        pg.check_status()


@pytest.mark.parametrize('sizes', [
    [(2000, 30, 10.0)],                                             # one BASELINE-size graph: 128 chunks of it
    [(2000, 30, 10.0), (1500, 25, 10.0), (2000, 30, 8.0), (700, 10, 10.0)],
    [(64, 8, 20.0)] * 70,                                           # many small complete graphs: chunks < one row
    [(4096, 30, 4.0), (100, 5, 5.0)],                               # the largest graph the counting path takes
    [(4500, 30, 4.0), (100, 5, 5.0)],                               # one node more than that: the radix sort
    [(3000, 30, 14.0), (300, 10, 6.0)],                             # 1.4 M edges in one graph: tiles beyond the 16-bit LDS list
])
@pytest.mark.parametrize('placement', ['tiles', 'scatter'])
def test_by_column_lists_by_counting_equal_the_sort(sizes, placement, monkeypatch):
    """Round 3: with a host bound on the graphs' sizes pvs_graph_prepare_runs builds colptr / cedge by a counting
    transpose per graph (graph_prepare.hip, k_csc_pass) instead of a radix sort of all edges by column. Same arrays
    as the sort (stable: ascending sorted position inside a column, duplicate edges included), whatever the chunking.
    Round 6: the placement pass through LDS-sorted tiles (k_csc_place_tiles; PVS_CSC_TILES=2 forces it where the
    heuristic would not take it, 0 selects the direct scatter) - both array for array the sort's output."""
    from pointvs_amd.graph import Batch, prepare_graph, runs_layout
    from pointvs_amd.synthetic import synthetic_graph
    monkeypatch.setenv('PVS_CSC_TILES', '2' if placement == 'tiles' else '0')
    items = [synthetic_graph(900 + k, n_nodes=n, n_lig=nl, edge_radius=r) for k, (n, nl, r) in enumerate(sizes)]
    batch = Batch.from_data_list(items).to('cuda')
    layout = runs_layout(batch)
    assert layout[0].max_graph_nodes == max(s[0] for s in sizes)
    n = int(batch.x.shape[0])
    a = prepare_graph(batch.edge_index, batch.edge_attr, n, need_backward=True, layout=layout)
    b = prepare_graph(batch.edge_index, batch.edge_attr, n, need_backward=True)
    a.check_status(); b.check_status()
    for name in ('rowptr', 'row', 'col', 'etype', 'perm', 'colptr', 'cedge'):
        assert torch.equal(a.t[name], b.t[name]), name
    # the definition, independent of either implementation
    col = a.t['col'].cpu().numpy().astype(np.int64)
    order = np.argsort(col, kind='stable')
    assert np.array_equal(a.t['cedge'].cpu().numpy(), order)
    assert np.array_equal(a.t['colptr'].cpu().numpy(), np.searchsorted(col[order], np.arange(n + 1)))


@pytest.mark.parametrize('family', ['default', 'h32_att', 'h32_edgeres_att', 'h64_att', 'generic_h16', 'wide128',
                                    'wide96_edgeres_att'])
def test_results_do_not_depend_on_stale_memory(family):
    """A result that changes with the bytes the allocator happens to hand out is a read of memory this step
    has not written - invisible while every step repeats the previous one (the block comes back holding the
    same values), wrong in training. Between repeats every cached block is overwritten, once with NaN and
    once with finite garbage (tools/soak.py, which runs 300 repeats per state and kernel family); per-layer
    node features and coordinates, logits, loss and every gradient must keep ONE bit pattern."""
    from tools import soak
    rec = soak.soak_family(family, soak.FAMILIES[family], repeats=3, n_graphs=2, log=lambda *_: None)
    assert rec['tensors_hashed'] > 10
    assert not rec['tensors_with_more_than_one_hash'], rec


@pytest.mark.parametrize('hid,flags', [(32, dict()), (32, dict(edge_attention=True, tanh=True)),
                                       (32, dict(edge_residual=True, normalize=True)),
                                       (64, dict()), (64, dict(edge_attention=True, node_attention=True))])
def test_upstream_gradients_spanning_many_binades(hid, flags):
    """The fp16-split products carry one power-of-two scale per operand and 32-edge tile (edge_mfma_common.h,
    "f16x2"). Here the graphs of one batch receive upstream gradients of 1, 1e-18, 1e+12 and 1e-24 (a saturated BCE
    loss does that: sigmoid'(37) ~ 1e-16): every graph's input gradients must be as accurate RELATIVE TO THAT
    GRAPH'S OWN magnitude as fp32 allows - the scales follow the tiles down and up, and no tile mixes two graphs
    (PvsGraph.graph_eptr) - and the weight gradients, sums over all graphs dominated by the largest, relative to
    theirs. Oracle: autograd on the fp64 layer."""
    from oracle import egnn_oracle as orc
    from pointvs_amd.egnn_satorras import EGNNLayer
    from pointvs_amd.graph import Batch
    from pointvs_amd.synthetic import synthetic_graph
    torch.manual_seed(3)
    layer = EGNNLayer(hid, hid, hid, edges_in_d=3, **flags).cuda()
    items = [synthetic_graph(700 + k, n_nodes=260, n_lig=12, edge_radius=6.0) for k in range(4)]
    g = Batch.from_data_list(items)
    n = g.x.shape[0]
    scales = torch.tensor([1.0, 1e-18, 1e12, 1e-24], dtype=torch.float64)[g.batch]
    rng = np.random.default_rng(2)
    h0 = torch.from_numpy(rng.normal(size=(n, hid)).astype(np.float32))
    wh = torch.from_numpy(rng.normal(size=(n, hid))) * scales[:, None]
    wx = torch.from_numpy(rng.normal(size=(n, 3))) * scales[:, None]

    from pointvs_amd.graph import prepared_for
    h = h0.cuda().requires_grad_(True)
    x = g.pos.cuda().requires_grad_(True)
    pg = prepared_for(g.edge_index.cuda(), g.edge_attr.cuda(), n)
    pg.set_graph_ptr(g.ptr.cuda())          # as the models do: tiles of the fp16-split backward end at graph ends
    h1, x1, _ = layer.forward_prepared(pg, h, x)
    ((h1 * wh.float().cuda()).sum() + (x1 * wx.float().cuda()).sum()).backward()

    sd = {'L.' + k: v.detach().cpu().double().requires_grad_(True) for k, v in layer.state_dict().items()}
    kw = dict(orc.BUILD_NET_DEFAULTS, residual=True, normalize=False, tanh=False, graphnorm=False)
    kw.update(flags)
    kw['edge_attention_here'] = kw['edge_attention']
    kw['node_attention_here'] = kw['node_attention']
    hr = h0.double().requires_grad_(True)
    xr = g.pos.double().requires_grad_(True)
    h2, x2, _, _, _ = orc.egnn_layer(sd, 'L.', kw, hr, g.edge_index, xr, g.edge_attr, None)
    ((h2 * wh.float().double()).sum() + (x2 * wx.float().double()).sum()).backward()

    def graph_rel(a, b):
        a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
        return float(np.abs(a - b).max() / np.abs(b).max())
    for gid in range(4):
        rows = (g.batch == gid).numpy()
        assert graph_rel(h.grad.cpu().numpy()[rows], hr.grad.numpy()[rows]) < 2e-5, ('g_h', gid)
        assert graph_rel(x.grad.cpu().numpy()[rows], xr.grad.numpy()[rows]) < 2e-5, ('g_x', gid)
    for name, p in layer.named_parameters():
        ref = sd['L.' + name].grad.numpy()
        assert np.isfinite(p.grad.cpu().numpy()).all(), name
        assert graph_rel(p.grad.cpu().numpy(), ref) < 2e-5, name


def test_edge_dropout_drops_both_directions_of_a_pair_together():
    """dropout_adj(force_undirected=True) as SartorrasEGNN.get_embeddings applies it (egnn_satorras.py:320-323):
    survivors are drawn among the row <= col copies only, their reverses are appended with the same attributes,
    the kept fraction is 1 - p, p = 0 / eval are the identity, and the draw is a function of (seed, step).
    (Semantics restated from torch_geometric 2.0.4; the random stream is the library's own: no parity vectors.)"""
    from pointvs_amd import functional as PF
    from pointvs_amd.graph import Batch
    from pointvs_amd.synthetic import synthetic_graph
    g = Batch.from_data_list([synthetic_graph(60 + k, n_nodes=400, n_lig=20, edge_radius=6.0) for k in range(3)]).to('cuda')
    ei, ea = g.edge_index, g.edge_attr
    same_i, same_a = PF.dropout_adj(ei, ea, 0.0, training=True)
    assert same_i is ei and same_a is ea
    empty_i, empty_a = PF.dropout_adj(ei[:, :0], ea[:0], 0.5, training=True)          # (used to raise)
    assert empty_i.shape == (2, 0) and empty_a.shape == (0, 3)
    same_i, _ = PF.dropout_adj(ei, ea, 0.4, training=False)
    assert same_i is ei
    p = 0.3
    oi, oa = PF.dropout_adj(ei, ea, p, training=True, seed=5, step=1)
    oi2, oa2 = PF.dropout_adj(ei, ea, p, training=True, seed=5, step=1)
    oi3, _ = PF.dropout_adj(ei, ea, p, training=True, seed=5, step=2)
    assert torch.equal(oi, oi2) and torch.equal(oa, oa2)
    assert oi.shape != oi3.shape or not torch.equal(oi, oi3)
    k = oi.shape[1] // 2
    assert oi.shape[1] == 2 * k and oa.shape == (2 * k, 3)
    # second half = the reverses of the first half, attributes repeated
    assert torch.equal(oi[0, :k], oi[1, k:]) and torch.equal(oi[1, :k], oi[0, k:]) and torch.equal(oa[:k], oa[k:])
    assert bool((oi[0, :k] <= oi[1, :k]).all())
    # survivors are input edges with their own attributes, in input order (a subsequence of the row <= col copies)
    cand = (ei[0] <= ei[1]).nonzero().reshape(-1)
    n = int(g.x.shape[0])
    key_in = (ei[0, cand] * n + ei[1, cand]) * 4 + ea[cand].argmax(1)
    key_out = (oi[0, :k] * n + oi[1, :k]) * 4 + oa[:k].argmax(1)
    it = iter(key_in.tolist())
    assert all(any(v == w for w in it) for v in key_out.tolist()), 'survivors are not a subsequence of the input'
    frac = k / cand.numel()
    assert abs(frac - (1 - p)) < 4 * np.sqrt(p * (1 - p) / cand.numel()) + 1e-3, frac


def test_model_with_edge_dropout_trains_and_is_the_plain_model_in_eval():
    """dropout > 0 no longer raises: in training mode every forward draws a new edge subset (different losses for
    different steps, same for the same (seed, step)), gradients are finite; in eval mode the model is the plain one."""
    from pointvs_amd.graph import Batch
    from pointvs_amd.synthetic import synthetic_graph
    g = Batch.from_data_list([synthetic_graph(80 + k, n_nodes=300, n_lig=16, edge_radius=6.0) for k in range(2)])
    plain, _ = make_model(seed=4, num_layers=2, residual=True)
    drop, _ = make_model(seed=4, num_layers=2, residual=True, dropout=0.25)
    y_plain, _ = gpu_run(plain, g)
    y_eval, _ = gpu_run(drop, g)                   # make_model returns .eval()
    assert np.array_equal(y_plain, y_eval)
    drop.train()
    torch.manual_seed(9)
    drop._dropout_calls = 0
    y1, g1 = gpu_run(drop, g)
    y2, _ = gpu_run(drop, g)
    drop._dropout_calls = 0
    y1b, _ = gpu_run(drop, g)
    assert np.array_equal(y1, y1b) and not np.array_equal(y1, y2) and not np.array_equal(y1, y_plain)
    assert all(np.isfinite(v).all() for v in g1.values() if v is not None)


# ---- dynamic range INSIDE one graph (round 4) --------------------------------------------
def row_rel(a, b):
    """max over rows of  max_c |a - b| / max_c |b|  (each row relative to ITS OWN magnitude), and the 99th percentile."""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    a, b = a.reshape(a.shape[0], -1), b.reshape(b.shape[0], -1)
    scale = np.abs(b).max(1)
    keep = scale > 0
    r = np.abs(a - b).max(1)[keep] / scale[keep]
    return float(r.max()), float(np.quantile(r, 0.99))


def dynamic_range_errors(hid, flags, log2_range, seed=5, n_nodes=500):
    """One layer on ONE graph whose node rows span 2^(+-log2_range): (a) the input features h, row i times
    2^(log2_range u_i); (b) separately, the upstream gradient g_h' rows. Returns, per kernel family ('f16x2' = the
    default fp16-split products, 'fp32' = PVS_EGNN_BF16X3=0, exact fp32 MFMAs), the row-relative errors against the
    fp64 oracle of: h_out and the per-edge edge_feat / att_val of run (a); g_h of run (b)."""
    from oracle import egnn_oracle as orc
    from pointvs_amd.egnn_satorras import EGNNLayer
    from pointvs_amd.graph import Batch
    from pointvs_amd.synthetic import synthetic_graph
    torch.manual_seed(seed)
    layer = EGNNLayer(hid, hid, hid, edges_in_d=3, **flags).cuda()
    g = Batch.from_data_list([synthetic_graph(900 + seed, n_nodes=n_nodes, n_lig=16, edge_radius=6.0)])
    n = g.x.shape[0]
    rng = np.random.default_rng(seed)
    row_scale = torch.from_numpy(np.exp2(log2_range * rng.uniform(-1, 1, size=(n, 1))))
    h_plain = torch.from_numpy(rng.normal(size=(n, hid)).astype(np.float32))
    h_wide = (h_plain.double() * row_scale).float()                   # (a): rows of h span the range
    w_plain = torch.from_numpy(rng.normal(size=(n, hid)).astype(np.float32))
    w_wide = (w_plain.double() * row_scale).float()                   # (b): rows of the upstream gradient do

    sd = {'L.' + k: v.detach().cpu().double().requires_grad_(True) for k, v in layer.state_dict().items()}
    kw = dict(orc.BUILD_NET_DEFAULTS, residual=True, normalize=False, tanh=False, graphnorm=False)
    kw.update(flags)
    kw['edge_attention_here'] = kw['edge_attention']
    kw['node_attention_here'] = kw['node_attention']
    names = [name for name, _ in layer.named_parameters()]

    def oracle(h_in, w_up):
        for v in sd.values():
            v.grad = None
        hr = h_in.double().requires_grad_(True)
        h2, _, m2, att2, _ = orc.egnn_layer(sd, 'L.', kw, hr, g.edge_index, g.pos.double(), g.edge_attr, None)
        (h2 * w_up.double()).sum().backward()
        pg = {name: (None if sd['L.' + name].grad is None else sd['L.' + name].grad.numpy().copy()) for name in names}
        return (h2.detach().numpy(), m2.detach().numpy(), None if att2 is None else att2.detach().numpy(),
                hr.grad.numpy(), pg)

    ref_a, ref_b = oracle(h_wide, w_plain), oracle(h_plain, w_wide)
    out = {}
    for family, env in (('f16x2', None), ('fp32', '0')):
        if env is None:
            os.environ.pop('PVS_EGNN_BF16X3', None)
        else:
            os.environ['PVS_EGNN_BF16X3'] = env
        try:
            def run(h_in, w_up):
                layer.zero_grad(set_to_none=True)
                h = h_in.cuda().requires_grad_(True)
                h1, _, _, m1 = layer(h, g.edge_index.cuda(), g.pos.cuda(), g.edge_attr.cuda())
                (h1 * w_up.cuda()).sum().backward()
                att = layer.att_val
                pg = {name: (None if p.grad is None else p.grad.detach().cpu().numpy()) for name, p in layer.named_parameters()}
                return h1.detach().cpu().numpy(), m1.detach().cpu().numpy(), att, h.grad.cpu().numpy(), pg
            got_a, got_b = run(h_wide, w_plain), run(h_plain, w_wide)
        finally:
            os.environ.pop('PVS_EGNN_BF16X3', None)
        rec = dict(h_out=row_rel(got_a[0], ref_a[0]), edge_feat=row_rel(got_a[1], ref_a[1]),
                   g_h_wide_input=row_rel(got_a[3], ref_a[3]), g_h=row_rel(got_b[3], ref_b[3]))
        if ref_a[2] is not None:    # a gate value lies in [0, 1]: absolute error (a gate of 1e-30 has no relative one)
            d = np.abs(np.asarray(got_a[2], dtype=np.float64).reshape(-1) - ref_a[2].reshape(-1))
            rec['att_val'] = (float(d.max()), float(np.quantile(d, 0.99)))
        # every parameter gradient of both runs, each tensor relative to ITS OWN largest entry (round 5: the weight
        # gradients of the H = 32 backward accumulate under lazily moving operand scales, edge_mfma_common.h)
        pgrads = {}
        for tag, got, ref in (('wide_input', got_a[4], ref_a[4]), ('wide_upstream', got_b[4], ref_b[4])):
            for name in names:
                if ref[name] is None or not np.abs(ref[name]).max() > 0:      # (unused by this loss: None or zeros)
                    assert got[name] is None or not np.any(got[name]), (tag, name)
                    continue
                assert np.isfinite(got[name]).all(), (tag, name)
                pgrads[f'{tag}:{name}'] = float(np.abs(got[name].astype(np.float64) - ref[name]).max() / np.abs(ref[name]).max())
        rec['param_grads'] = pgrads
        out[family] = rec
    return out


_ATT = dict(edge_attention=True, node_attention=True)


@pytest.mark.parametrize('log2_range', [6, 20])
@pytest.mark.parametrize('hid,flags', [(32, {}), (32, _ATT), (64, {}), (64, _ATT)])
def test_dynamic_range_inside_one_graph(hid, flags, log2_range):
    """The fp16-split products ("f16x2": two fp16 parts per operand, 22 bits, power-of-two operand scales) against the
    exact-fp32-MFMA family on ONE graph whose node rows span 2^(+-6) and 2^(+-20) - magnitudes mixed inside every
    32-edge tile, which the per-graph test above cannot do. Every row (node or edge) is compared RELATIVE TO ITS OWN
    magnitude with the fp64 oracle; the split products must stay within 4x the fp32 family's error (+ 2e-6):
      * forward (per-EDGE operand scales since round 4): h_out, the returned per-edge messages and gate values;
      * backward with the upstream gradient rows spanning the range: g_h;
      * EVERY parameter gradient of both runs (round 5), each tensor relative to its own largest entry.
    Two stated exceptions, both on the WORST row only (the 99th percentile over rows obeys the 4x bound):
      * h_out and the gate values with edge attention: a gate sigmoid(w_a . m + b_a) whose logit is a small difference of
        large terms amplifies the error of m by sum|w_a m| / |logit| in BOTH families; there the split products'
        per-operand rounding (two roundings to 11 bits against one to 24) shows: up to 16x (h_out) / 32x (one gate
        value, absolute error 7e-4 against 3e-5) on the worst row;
      * g_h when the INPUT rows span 2^(+-20): the backward keeps ONE scale per 32-edge tile (its weight gradients
        sum over the tile's edges; kept while the tile maximum stays within two binades of its ceiling), so an element
        2^-k below its tile's largest keeps 22 - max(0, k - 14) bits (absolute error 2^-36 of the tile maximum): up to 64x on the worst row - where the fp32 family is itself
        1e-4 ... 1e-2 off. DESIGN.md section 4 states this bound; bench.py's config.arithmetic names it."""
    rec = dynamic_range_errors(hid, flags, log2_range)
    a, b = rec['f16x2'], rec['fp32']
    att = bool(flags)
    for name, err in a['param_grads'].items():
        # fp32-family error of the same tensor, with a floor of a few fp32 roundings of a sum of this size. With edge
        # attention and INPUT rows spanning 2^(+-20) every gradient inherits the gate-value exception above (one gate
        # off by 7e-4 sits on a row 2^20 above the rest: measured 21.5x on all tensors alike, where the fp32 family is
        # itself 3.7e-4 off its own largest entry): the gate's 32x applies (profiles/r05_dynamic_range.txt)
        worst = 32 if (att and log2_range == 20 and name.startswith('wide_input:')) else 4
        assert err <= worst * b['param_grads'][name] + 2e-6, (name, err, b['param_grads'][name])
    for tensor in a:
        if tensor == 'param_grads':
            continue
        mx, p99 = a[tensor]
        mx32, p9932 = b[tensor]
        assert p99 <= 4 * p9932 + 2e-6, (tensor, 'p99', p99, p9932)
        worst = 64 if tensor == 'g_h_wide_input' else 32 if tensor == 'att_val' else 16 if (tensor == 'h_out' and att) else 4
        assert mx <= worst * mx32 + (1e-5 if tensor == 'g_h_wide_input' else 2e-6), (tensor, 'max', mx, mx32)


@pytest.mark.parametrize('breakage', ['edge_ptr_short', 'edge_ptr_past_end', 'edge_ptr_not_monotone', 'node_ptr_shifted',
                                      'node_ptr_negative'])
def test_counting_transpose_stays_in_bounds_on_broken_layout_tables(breakage):
    """the by-column counting transpose of pvs_graph_prepare_runs used the caller's edge_ptr /
    node_ptr unclamped, so a DIRECT C-ABI caller with inconsistent tables (Python's runs_layout validates them on the
    host) could make it read col / write cedge out of bounds before the host sees status bit 4. Here the tables are
    corrupted behind runs_layout's back, with max_graph_nodes > 0 (the counting path) and with the allocator's caching
    off, so that every output is its own allocation and an access outside it faults: the call must complete, set
    status bit 4 (check_status raises) and leave colptr monotone inside [0, E] with every list entry inside [0, E)."""
    from pointvs_amd.graph import Batch, prepare_graph, runs_layout
    from pointvs_amd.synthetic import synthetic_graph
    items = [synthetic_graph(300 + k, n_nodes=n, n_lig=8, edge_radius=6.0) for k, n in enumerate((200, 90, 310))]
    batch = Batch.from_data_list(items).to('cuda')
    node_ptr, edge_ptr = runs_layout(batch)
    n, e = int(batch.x.shape[0]), int(batch.edge_index.shape[1])
    node_ptr, edge_ptr = node_ptr.clone(), edge_ptr.clone()
    if breakage == 'edge_ptr_short':
        edge_ptr[-1] = e - 1000
    elif breakage == 'edge_ptr_past_end':
        edge_ptr[-1] = e + 100000
        edge_ptr[2] = e + 5000
    elif breakage == 'edge_ptr_not_monotone':
        edge_ptr[1], edge_ptr[2] = edge_ptr[2].item(), edge_ptr[1].item()
    elif breakage == 'node_ptr_shifted':
        node_ptr[0] = 50
        node_ptr[2] = n + 400
    else:
        node_ptr[1] = -7
    node_ptr.max_graph_nodes = 310
    pg = prepare_graph(batch.edge_index, batch.edge_attr, n, need_backward=True, layout=(node_ptr, edge_ptr))
    torch.cuda.synchronize()                      # (an out-of-bounds access would surface here as a GPU fault)
    with pytest.raises(ValueError):
        pg.check_status()
    colptr, cedge = pg.t['colptr'].cpu().numpy(), pg.t['cedge'].cpu().numpy()
    assert colptr[0] >= 0 and colptr[-1] <= e and np.all(np.diff(colptr) >= 0)
    used = cedge[:colptr[-1]]
    assert used.size == 0 or (used.min() >= 0 and used.max() < e)
    for name in ('row', 'col'):
        v = pg.t[name].cpu().numpy()
        assert v.min() >= 0 and v.max() < n, name
    rowptr = pg.t['rowptr'].cpu().numpy()
    assert rowptr[0] >= 0 and rowptr[-1] <= e and np.all(np.diff(rowptr) >= 0)


def test_profile_hook_reports_every_launch():
    """pvs_profile_read_each (round 5; bench.py's roofline.avg_launch_ms_full_work): with the edge-backward group enabled, a
    3-layer training step records exactly three launches, their per-launch times sum to pvs_profile_read's total, and
    the first one - the last layer's backward, which has no coordinate branch - is the shortest."""
    import ctypes as C
    from pointvs_amd import _lib
    from pointvs_amd.graph import Batch
    from pointvs_amd.synthetic import synthetic_graph
    lib = _lib.lib()
    model, _ = make_model(seed=1, num_layers=3)
    g = Batch.from_data_list([synthetic_graph(40 + k, n_nodes=1200, n_lig=20, edge_radius=8.0) for k in range(4)])
    gpu_run(model, g)                                   # warm-up (allocations, first-use work)
    lib.pvs_profile_reset()
    lib.pvs_profile_enable(1 << 2)                      # group 1 = edge_bwd
    try:
        gpu_run(model, g)
        torch.cuda.synchronize()
    finally:
        lib.pvs_profile_enable(0)
    tot, cnt = C.c_double(0.0), C.c_int64(0)
    assert lib.pvs_profile_read(b'edge_bwd', C.byref(tot), C.byref(cnt)) == 0
    buf, n = (C.c_double * 8)(), C.c_int64(0)
    assert lib.pvs_profile_read_each(b'edge_bwd', buf, 8, C.byref(n)) == 0
    each = [buf[k] for k in range(n.value)]
    assert cnt.value == 3 and n.value == 3, (cnt.value, n.value)
    assert abs(sum(each) - tot.value) < 1e-6 and all(t > 0 for t in each)
    assert each[0] < min(each[1:]), each
    none, zero = (C.c_double * 1)(), C.c_int64(-1)
    assert lib.pvs_profile_read_each(b'edge_fwd', none, 1, C.byref(zero)) == 0 and zero.value == 0      # group not enabled
    assert lib.pvs_profile_read_each(b'no_such_group', none, 1, C.byref(zero)) == -1
    lib.pvs_profile_reset()
