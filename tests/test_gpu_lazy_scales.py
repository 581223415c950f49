"""Directed tests of the lazily moving operand scales of the H = 32 edge backward (round 5; VERDICT r04 weak item 1,
ADVICE r04 item 1). The kernel (csrc/edge_bwd_f16.hip) accumulates its weight gradients IN the matrix core's
accumulators at the scale of the operand images; when an image's power-of-two scale moves, `pvs_rescale_acc`
(csrc/edge_mfma_common.h) multiplies the accumulators by the exact ratio, SKIPS a tile that lies more than 2^60 below
what the accumulator has held, or RESTARTS the accumulator when the tile lies more than 2^60 above it.

The graphs here are disjoint complete graphs on 33 nodes: every row of the CSR has exactly 32 edges, a wave's chunk
starts at a row, so every 32-edge tile IS one row and the upstream gradient of a node sets the magnitude of exactly
one tile. Consecutive rows then step by chosen powers of two, so that ONE launch drives the rescale (2^5, 2^30), skip
and restart (2^70) branches inside every wave's chunk. Oracle: autograd of the fp64 layer; bound: the strict per-tensor
criterion of tests/_golden.py (1e-5 of the tensor's own largest entry + 4 x the fp32 reference's own distance).

Reference semantics: /root/reference/point_vs/models/geometric/egnn_satorras.py:123-132,168-187 (edge_model,
coord_model) under autograd."""
import numpy as np
import pytest
import torch

from tests._golden import CaseLog, assert_strict

pytestmark = pytest.mark.gpu

K = 33          # nodes per complete graph: 32 edges per row
H = 32

PATTERNS = {
    # consecutive differences: +5, +30, -70, +5, +70, (zero tile), -30, -5, -5: rescale both ways, skip, restart
    'mixed': [0, 5, 35, -35, -30, 40, None, 10, 5],
    # three steps of 2^-50 each: every step is inside the 2^60 window, the sum is not (an accumulator that follows each
    # step multiplies what it holds by 2^150)
    'staircase': [60, 10, -40, -90, -40, 10],
    # gentle steps around the lazy window of two binades: the scale moves on some tiles and stays on others
    'window': [0, 1, 2, 3, 4, 2, 0, -3, -1],
}


def complete_blocks(n_blocks, seed):
    from pointvs_amd.graph import Batch
    rng = np.random.default_rng(seed)
    n = n_blocks * K
    src, dst = np.nonzero(~np.eye(K, dtype=bool))
    ei = np.concatenate([np.stack([src, dst]) + b * K for b in range(n_blocks)], axis=1)
    et = rng.integers(0, 3, ei.shape[1])
    pos = rng.normal(size=(n, 3)).astype(np.float32) * 1.5
    return Batch(x=torch.zeros(n, 1), pos=torch.from_numpy(pos), edge_index=torch.from_numpy(ei.astype(np.int64)),
                 edge_attr=torch.nn.functional.one_hot(torch.from_numpy(et), 3),
                 batch=torch.zeros(n, dtype=torch.long), num_graphs=1)


def run_layer(layer, flags, g, h0, wh, wx):
    """(GPU input gradients + parameter gradients), (oracle fp64), (oracle fp32) of loss = sum(h' wh) + sum(x' wx)."""
    from oracle import egnn_oracle as orc
    from pointvs_amd.graph import prepared_for
    n = h0.shape[0]
    layer.zero_grad(set_to_none=True)
    h = h0.cuda().requires_grad_(True)
    x = g.pos.cuda().requires_grad_(True)
    pg = prepared_for(g.edge_index.cuda(), g.edge_attr.cuda(), n)
    h1, x1, _ = layer.forward_prepared(pg, h, x)
    ((h1 * wh.float().cuda()).sum() + (x1 * wx.float().cuda()).sum()).backward()
    got = dict(g_h=h.grad.cpu().numpy(), g_x=x.grad.cpu().numpy())
    for name, p in layer.named_parameters():
        got[name] = None if p.grad is None else p.grad.cpu().numpy()

    kw = dict(orc.BUILD_NET_DEFAULTS, residual=True, normalize=False, tanh=False, graphnorm=False)
    kw.update(flags)
    kw['edge_attention_here'] = kw['edge_attention']
    kw['node_attention_here'] = kw['node_attention']
    refs = []
    for dtype in (torch.float64, torch.float32):
        sd = {'L.' + k: v.detach().cpu().to(dtype).requires_grad_(True) for k, v in layer.state_dict().items()}
        hr = h0.to(dtype).requires_grad_(True)
        xr = g.pos.to(dtype).requires_grad_(True)
        h2, x2, _, _, _ = orc.egnn_layer(sd, 'L.', kw, hr, g.edge_index, xr, g.edge_attr, None)
        ((h2 * wh.float().to(dtype)).sum() + (x2 * wx.float().to(dtype)).sum()).backward()
        ref = dict(g_h=hr.grad.numpy(), g_x=xr.grad.numpy())
        for name, _ in layer.named_parameters():
            ref[name] = None if sd['L.' + name].grad is None else sd['L.' + name].grad.numpy()
        refs.append(ref)
    return got, refs[0], refs[1]


def check(case, got, ref64, ref32, names, row_tol=2e-5):
    log = CaseLog(case)
    for name in names:
        if ref64[name] is None:
            assert got[name] is None or not np.any(got[name]), name
            continue
        assert_strict(got[name], ref64[name], ref32[name], f'{case} grad {name}', log=log)
    # input gradients: every row relative to its own magnitude (rows differ by up to 2^150 here)
    for key in ('g_h', 'g_x'):
        a, b = got[key].astype(np.float64), ref64[key]
        assert np.isfinite(a).all(), key
        scale = np.abs(b).max(1)
        keep = scale > 0
        r = np.abs(a - b).max(1)[keep] / scale[keep]
        assert r.max() < row_tol, (case, key, float(r.max()))
    log.finish()


@pytest.mark.parametrize('flags', [dict(), dict(normalize=True, tanh=True)], ids=['plain', 'normalize_tanh'])
@pytest.mark.parametrize('pattern', sorted(PATTERNS))
def test_tile_magnitudes_stepping_inside_one_chunk(pattern, flags):
    """Upstream gradient rows (= tiles) step by the pattern's powers of two inside every wave's chunk: 24 complete
    graphs = 792 rows = 25,344 edges = 56 chunks of ~14 rows, so each chunk walks through more than one period.
    Every parameter gradient is held to the strict bound; in particular edge_mlp.2.* and coord_mlp.0.*, the sums that
    live in the rescaled accumulators."""
    from pointvs_amd.egnn_satorras import EGNNLayer
    torch.manual_seed(11)
    layer = EGNNLayer(H, H, H, edges_in_d=3, **flags).cuda()
    g = complete_blocks(24, seed=7)
    n = g.pos.shape[0]
    rng = np.random.default_rng(3)
    p = PATTERNS[pattern]
    row_scale = np.array([0.0 if p[r % len(p)] is None else 2.0 ** p[r % len(p)] for r in range(n)])
    h0 = torch.from_numpy(rng.normal(size=(n, H)).astype(np.float32))
    wh = torch.from_numpy(rng.normal(size=(n, H)) * row_scale[:, None])
    wx = torch.from_numpy(rng.normal(size=(n, 3)) * row_scale[:, None])
    got, ref64, ref32 = run_layer(layer, flags, g, h0, wh, wx)
    names = [name for name, _ in layer.named_parameters()]
    for must in ('edge_mlp.2.weight', 'edge_mlp.2.bias', 'coord_mlp.0.weight', 'coord_mlp.0.bias'):
        assert must in names and ref64[must] is not None and got[must] is not None, must
    check(f'lazy_{pattern}_{"nt" if flags else "plain"}', got, ref64, ref32, names)


def test_bias_sums_survive_an_all_zero_activation_tile():
    """one decision used to skip a tile's weight-gradient AND bias-column products. With all-zero
    ACTIVATIONS in a tile (messages m = 0: here one complete graph whose nodes have h = 0, under edge-MLP biases and
    rho / edge-class columns set to 0) the activation image's scale lies 2^110 away from its neighbours' and the tile's
    weight-gradient product is rightly skipped - but its gradient rows are ordinary and their column sums belong in
    edge_mlp.2.bias / coord_mlp.0.bias."""
    from pointvs_amd.egnn_satorras import EGNNLayer
    torch.manual_seed(12)
    layer = EGNNLayer(H, H, H, edges_in_d=3).cuda()
    with torch.no_grad():
        layer.edge_mlp[0].bias.zero_()
        layer.edge_mlp[0].weight[:, 2 * H:].zero_()
        layer.edge_mlp[2].bias.zero_()
    g = complete_blocks(24, seed=8)
    n = g.pos.shape[0]
    rng = np.random.default_rng(4)
    h0 = rng.normal(size=(n, H)).astype(np.float32)
    for blk in (0, 5, 6, 17, 23):                  # first / last of the range, two neighbours, one alone
        h0[blk * K:(blk + 1) * K] = 0.0
    h0 = torch.from_numpy(h0)
    wh = torch.from_numpy(rng.normal(size=(n, H)))
    wx = torch.from_numpy(rng.normal(size=(n, 3)))
    got, ref64, ref32 = run_layer(layer, {}, g, h0, wh, wx)
    # the zeroed tiles carry a fifth of the bias sums
    names = [name for name, _ in layer.named_parameters()]
    check('lazy_zero_activation_tiles', got, ref64, ref32, names)
