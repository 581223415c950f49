"""GPU radius-graph builder (pvs_radius_graph_*) against the reference's
generate_edges: its own test vectors, outputs of the reference itself (tests/golden/edges_*.npz)
and, at BASELINE sizes, the COO path (synthetic generator + pvs_graph_prepare). Integer work:
everything is compared for exact equality."""
import json

import numpy as np
import pytest
import torch

from tests._golden import GOLDEN_DIR, needs_caching_allocator

pytestmark = pytest.mark.gpu

EDGE_CASES = ['edges_small', 'edges_small_prune', 'edges_default_radii', 'edges_r10', 'edges_bonds',
              'edges_no_inter']


@pytest.mark.parametrize('name', EDGE_CASES)
def test_generate_edges_matches_reference_outputs(name):
    from pointvs_amd.radius_graph import generate_edges
    z = np.load(GOLDEN_DIR / f'{name}.npz')
    keep, ei, attrs = generate_edges(torch.from_numpy(z['xyz']).cuda(), torch.from_numpy(z['bp']).cuda(),
                                     float(z['inter']), float(z['intra']), prune=bool(z['prune']))
    assert np.array_equal(keep.cpu().numpy(), z['keep'])
    assert np.array_equal(ei[0].cpu().numpy(), z['rows']) and np.array_equal(ei[1].cpu().numpy(), z['cols'])
    assert np.array_equal(attrs.cpu().numpy(), z['attrs'])


def test_generate_edges_matches_reference_test_vectors():
    from pointvs_amd.radius_graph import generate_edges
    d = json.loads((GOLDEN_DIR / 'generate_edges_reference_tests.json').read_text())
    xyz = torch.tensor([d['struct']['x'], d['struct']['y'], d['struct']['z']], dtype=torch.float32).t().contiguous()
    bp = torch.tensor(d['struct']['bp'])
    for key, prune in (('no_prune', False), ('prune', True)):
        keep, ei, attrs = generate_edges(xyz.cuda(), bp.cuda(), d['inter_radius'], d['intra_radius'], prune=prune)
        assert ei[0].tolist() == d[key]['rows'] and ei[1].tolist() == d[key]['cols']
        assert attrs.tolist() == d[key]['attrs']
    assert keep.tolist() == list(range(8))


def _batch(n_graphs, n_nodes, radius, seed0=50):
    from pointvs_amd.graph import Batch
    from pointvs_amd.synthetic import synthetic_graph
    return Batch.from_data_list([synthetic_graph(seed0 + g, n_nodes=n_nodes, n_lig=12, edge_radius=radius)
                                 for g in range(n_graphs)])


@pytest.mark.parametrize('n_graphs,n_nodes,radius', [(3, 300, 6.0), (5, 257, 4.0), (32, 2000, 10.0)])
def test_radius_graph_equals_prepare_graph_of_the_coo(n_graphs, n_nodes, radius):
    """Array for array what pvs_graph_prepare builds from the reference-order COO + one-hot of the
    same atoms (the synthetic generator follows generate_edges; ragged sizes, then BASELINE cfg2)."""
    from pointvs_amd.graph import prepare_graph
    from pointvs_amd.radius_graph import edges_in_reference_order, radius_graph
    b = _batch(n_graphs, n_nodes, radius).to('cuda')
    pg_ref = prepare_graph(b.edge_index, b.edge_attr, b.x.shape[0])
    pg_ref.check_status()
    pg = radius_graph(b.pos, b.x[:, -1], b.ptr, inter_radius=radius)
    assert pg.n_edges == pg_ref.n_edges == b.edge_index.shape[1]
    for k in ('rowptr', 'row', 'col', 'etype', 'colptr', 'cedge', 'inv_deg'):
        assert torch.equal(pg.t[k][:len(pg_ref.t[k])], pg_ref.t[k]), k
    # perm: the reference lists edges graph by graph (PyG collation); the builder's order inside one
    # graph is the reference's, so compare per graph through the edge lists
    ei, attrs = edges_in_reference_order(radius_graph(b.pos[:n_nodes].contiguous(), b.x[:n_nodes, -1], None,
                                                      inter_radius=radius))
    e0 = b.graph_edge_counts[0]
    assert torch.equal(ei, b.edge_index[:, :e0]) and torch.equal(attrs, b.edge_attr[:e0].argmax(1))


def test_model_step_is_identical_on_a_built_graph():
    """Forward + backward of the model on a radius_graph() result == on prepare_graph() of the COO."""
    from pointvs_amd.egnn_satorras import SartorrasEGNN
    from pointvs_amd.graph import prepare_graph
    from pointvs_amd.radius_graph import radius_graph
    import tempfile
    b = _batch(4, 400, 6.0).to('cuda')
    kw = dict(dim_input=12, k=32, dim_output=1, num_layers=2, residual=True, edge_residual=False,
              edge_attention=True, normalize=False, tanh=True, dropout=0.0, graphnorm=False, update_coords=True,
              permutation_invariance=False, node_attention=False, gated_residual=False, rezero=False,
              softmax_attention=False, model_task='classification')
    torch.manual_seed(0)
    model = SartorrasEGNN(tempfile.mkdtemp(), 2e-3, 1e-4, silent=True, **kw)
    outs = []
    for pg in (prepare_graph(b.edge_index, b.edge_attr, b.x.shape[0]),
               radius_graph(b.pos, b.x[:, -1], b.ptr, inter_radius=6.0)):
        model.zero_grad(set_to_none=True)
        feats, _, _ = model.embed_prepared(pg, b.x, b.pos)
        feats.square().sum().backward()
        outs.append((feats.detach().clone(), [p.grad.clone() for p in model.parameters() if p.grad is not None]))
    assert torch.equal(outs[0][0], outs[1][0])
    for ga, gb in zip(outs[0][1], outs[1][1]):
        assert torch.equal(ga, gb)


def test_pose_batcher_matches_a_collated_batch_of_the_same_poses():
    """Screening batches (BASELINE config 5): forward on PoseBatcher's GPU-built batch == forward on
    the batch collated from per-pose graphs made with the oracle edge rule."""
    import tempfile
    from oracle.generate_edges_oracle import generate_edges as oracle_edges
    from pointvs_amd.egnn_satorras import SartorrasEGNN
    from pointvs_amd.graph import Batch, Data
    from pointvs_amd.radius_graph import PoseBatcher
    from pointvs_amd.synthetic import random_poses, screening_set
    lig, rec, feats = screening_set(seed=5001, n_nodes=300, n_lig=12)
    poses = random_poses(lig, 3, seed=3, max_shift=4.0)
    kw = dict(dim_input=12, k=32, dim_output=1, num_layers=2, residual=False, edge_residual=False,
              edge_attention=False, normalize=False, tanh=False, dropout=0.0, graphnorm=False, update_coords=True,
              permutation_invariance=False, node_attention=False, gated_residual=False, rezero=False,
              softmax_attention=False, model_task='classification')
    torch.manual_seed(0)
    model = SartorrasEGNN(tempfile.mkdtemp(), 2e-3, 1e-4, silent=True, **kw).eval()
    batcher = PoseBatcher(rec.cuda(), feats, 12, 3, edge_radius=6.0)
    with torch.no_grad():
        fast = model(batcher.load(poses.cuda())).reshape(-1).cpu()
    items = []
    bp = feats[:, -1].numpy()
    for p in poses:
        pos = torch.cat([p, rec], 0)
        _, (rows, cols), attrs = oracle_edges(pos.numpy(), bp, 6.0, 6.0, prune=False)
        items.append(Data(x=feats, pos=pos, edge_index=torch.from_numpy(np.vstack([rows, cols])).long(),
                          edge_attr=torch.nn.functional.one_hot(torch.from_numpy(attrs).long(), 3),
                          y=torch.tensor(0), lig_fname='l', rec_fname='r'))
    with torch.no_grad():
        slow = model(Batch.from_data_list(items).to('cuda')).reshape(-1).cpu()
    assert torch.equal(fast, slow)


@pytest.mark.parametrize('flags', [dict(), dict(edge_attention=True, node_attention=True, tanh=True, residual=True),
                                   dict(k=64, normalize=True, graphnorm=True)])
def test_receptor_screen_matches_the_plain_forward(flags):
    """First-layer receptor-receptor sums reused across poses (pointvs_amd/screening.py) == plain
    forward of the same poses, to fp32 summation order (rel 1e-5 per tensor)."""
    import tempfile
    from pointvs_amd.egnn_satorras import SartorrasEGNN
    from pointvs_amd.radius_graph import PoseBatcher
    from pointvs_amd.screening import ReceptorScreen
    from pointvs_amd.synthetic import random_poses, screening_set
    lig, rec, feats = screening_set(seed=5002, n_nodes=500, n_lig=14)
    poses = random_poses(lig, 8, seed=4, max_shift=5.0).cuda()
    kw = dict(dim_input=12, k=32, dim_output=1, num_layers=3, residual=False, edge_residual=False,
              edge_attention=False, normalize=False, tanh=False, dropout=0.0, graphnorm=False, update_coords=True,
              permutation_invariance=False, node_attention=False, gated_residual=False, rezero=False,
              softmax_attention=False, model_task='classification')
    kw.update(flags)
    torch.manual_seed(1)
    model = SartorrasEGNN(tempfile.mkdtemp(), 2e-3, 1e-4, silent=True, **kw).eval()
    screen = ReceptorScreen(model, rec.cuda(), feats, 14, 4, edge_radius=7.0)
    assert screen.reuse
    plain = PoseBatcher(rec.cuda(), feats, 14, 4, edge_radius=7.0)
    for k in range(2):
        fast = screen(poses[4 * k:4 * k + 4]).reshape(-1)
        with torch.no_grad():
            slow = model(plain.load(poses[4 * k:4 * k + 4])).reshape(-1)
        err = float((fast - slow).abs().max() / slow.abs().max().clamp_min(1e-30))
        assert err < 1e-5, err


def test_radius_graph_with_bond_radius_equals_prepare_graph_of_oracle_edges():
    """estimate_bonds shape (intra 2 A, inter 6 A, data_loaders.py:359-360) on a ragged batch: the
    built graph == pvs_graph_prepare of the oracle's per-graph edge lists, array for array."""
    from oracle.generate_edges_oracle import generate_edges as oracle_edges
    from pointvs_amd.graph import Batch, Data, prepare_graph
    from pointvs_amd.radius_graph import radius_graph
    from pointvs_amd.synthetic import synthetic_graph
    items = []
    for k, n in enumerate((150, 333, 64, 257)):
        g = synthetic_graph(900 + k, n_nodes=n, n_lig=9, edge_radius=1.0, density=0.08)
        _, (rows, cols), attrs = oracle_edges(g.pos.numpy(), g.x[:, -1].numpy(), 6.0, 2.0, prune=False)
        items.append(Data(x=g.x, pos=g.pos, edge_index=torch.from_numpy(np.vstack([rows, cols])).long(),
                          edge_attr=torch.nn.functional.one_hot(torch.from_numpy(attrs).long(), 3),
                          y=torch.tensor(0), lig_fname='l', rec_fname='r'))
    b = Batch.from_data_list(items).to('cuda')
    ref = prepare_graph(b.edge_index, b.edge_attr, b.x.shape[0])
    pg = radius_graph(b.pos, b.x[:, -1], b.ptr, inter_radius=6.0, intra_radius=2.0)
    assert pg.n_edges == ref.n_edges
    for k in ('rowptr', 'row', 'col', 'etype', 'colptr', 'cedge', 'inv_deg'):
        assert torch.equal(pg.t[k][:len(ref.t[k])], ref.t[k]), k


def test_pose_batch_builder_equals_the_generic_builder():
    """pvs_screen_graph_build (receptor-receptor pairs from a template, only ligand contacts tested)
    == pvs_radius_graph_* on the same pose batch, array for array; its ligand-touching CSR == the
    ligand-only build."""
    import tempfile
    from pointvs_amd.egnn_satorras import SartorrasEGNN
    from pointvs_amd.radius_graph import PoseBatcher, radius_graph
    from pointvs_amd.screening import ReceptorScreen
    from pointvs_amd.synthetic import random_poses, screening_set
    lig, rec, feats = screening_set(seed=5003, n_nodes=700, n_lig=17)
    poses = random_poses(lig, 5, seed=9, max_shift=6.0).cuda()
    kw = dict(dim_input=12, k=32, dim_output=1, num_layers=2, residual=False, edge_residual=False,
              edge_attention=False, normalize=False, tanh=False, dropout=0.0, graphnorm=False, update_coords=True,
              permutation_invariance=False, node_attention=False, gated_residual=False, rezero=False,
              softmax_attention=False, model_task='classification')
    torch.manual_seed(0)
    model = SartorrasEGNN(tempfile.mkdtemp(), 2e-3, 1e-4, silent=True, **kw).eval()
    for r_inter, r_intra in ((7.0, None), (6.0, 2.5)):
        screen = ReceptorScreen(model, rec.cuda(), feats, 17, 5, edge_radius=r_inter, intra_radius=r_intra)
        assert screen.fast_graph
        pg, gl = screen._build_fast(poses)
        screen.check()
        ref_b = PoseBatcher(rec.cuda(), feats, 17, 5, r_inter, r_intra).load(poses)
        ref = ref_b.prepared
        n = ref.n_nodes
        e = int(screen._fast['rowptr'][n].item())
        assert e == ref.n_edges
        assert torch.equal(screen._fast['rowptr'], ref.t['rowptr'])
        for k in ('row', 'col', 'etype'):
            assert torch.equal(screen._fast[k][:e], ref.t[k][:e]), k
        assert torch.equal(screen._fast['inv_deg'], ref.t['inv_deg'])
        lig_ref = radius_graph(ref_b.pos, ref_b.x[:, -1], ref_b.ptr, r_inter, r_intra, max_graph_nodes=717,
                               need_backward=False, ligand_pairs_only=True)
        el = int(screen._fast['rowptr_l'][n].item())
        assert el == lig_ref.n_edges
        assert torch.equal(screen._fast['rowptr_l'], lig_ref.t['rowptr'])
        for k in ('row', 'col', 'etype'):
            assert torch.equal(screen._fast[k + '_l'][:el], lig_ref.t[k][:el]), k


@needs_caching_allocator
def test_captured_screening_step_replays_on_new_poses():
    """hipGraph capture of the whole screening step (graph build + layer stack + head): replays on
    other pose batches give the eager results."""
    import tempfile
    from pointvs_amd.egnn_satorras import SartorrasEGNN
    from pointvs_amd.screening import ReceptorScreen
    from pointvs_amd.synthetic import random_poses, screening_set
    lig, rec, feats = screening_set(seed=5004, n_nodes=600, n_lig=20)
    poses = random_poses(lig, 12, seed=2, max_shift=5.0).cuda()
    kw = dict(dim_input=12, k=32, dim_output=1, num_layers=3, residual=True, edge_residual=False,
              edge_attention=True, normalize=False, tanh=True, dropout=0.0, graphnorm=False, update_coords=True,
              permutation_invariance=False, node_attention=False, gated_residual=False, rezero=False,
              softmax_attention=False, model_task='classification')
    torch.manual_seed(2)
    model = SartorrasEGNN(tempfile.mkdtemp(), 2e-3, 1e-4, silent=True, **kw).eval()
    eager = ReceptorScreen(model, rec.cuda(), feats, 20, 4, edge_radius=7.0)
    want = [eager(poses[4 * k:4 * k + 4]).reshape(-1).clone() for k in range(3)]
    eager.check()
    graph = ReceptorScreen(model, rec.cuda(), feats, 20, 4, edge_radius=7.0).capture(poses[:4])
    for k in (2, 0, 1):
        got = graph.replay(poses[4 * k:4 * k + 4]).reshape(-1).clone()
        assert torch.equal(got, want[k])
    graph.check()


@needs_caching_allocator
def test_screening_sweep_streams_size_buckets_and_writes_predictions(tmp_path):
    """ScreeningSweep: ligands of two sizes (two captured buckets, the second ligand of a size reuses the
    first one's captured step with its own features), pose counts that are not multiples of the batch,
    scores equal to the plain forward of the same poses, predictions file in the reference line format."""
    import tempfile
    from pointvs_amd.egnn_satorras import SartorrasEGNN
    from pointvs_amd.radius_graph import PoseBatcher
    from pointvs_amd.screening import ScreeningSweep
    from pointvs_amd.synthetic import random_poses, screening_set
    kw = dict(dim_input=12, k=32, dim_output=1, num_layers=3, residual=False, edge_residual=False,
              edge_attention=False, normalize=False, tanh=False, dropout=0.0, graphnorm=False, update_coords=True,
              permutation_invariance=False, node_attention=False, gated_residual=False, rezero=False,
              softmax_attention=False, model_task='classification')
    torch.manual_seed(2)
    model = SartorrasEGNN(tempfile.mkdtemp(), 2e-3, 1e-4, silent=True, **kw).eval()
    lig_a, rec, feats_a = screening_set(seed=5003, n_nodes=400, n_lig=12)
    rec_feats = feats_a[12:]
    gen = torch.Generator().manual_seed(0)
    ligs = [('ligA', feats_a[:12], lig_a, 7), ('ligB', feats_a[:9].roll(1, 0), lig_a[:9] * 0.9, 5),
            ('ligC', feats_a[:12].roll(3, 0), lig_a.flip(0), 4)]
    work = [(name, f, random_poses(pos, n, seed=20 + k, max_shift=4.0).cuda())
            for k, (name, f, pos, n) in enumerate(ligs)]
    sweep = ScreeningSweep(model, rec.cuda(), rec_feats, edge_radius=6.0, batch_size=4)
    got = sweep.run(work, predictions_file=tmp_path / 'screen.txt')
    assert sorted(sweep.buckets) == [9, 12] and sweep.batches_run == 2 + 2 + 1
    lines = (tmp_path / 'screen.txt').read_text().splitlines()
    assert len(lines) == 7 + 5 + 4
    at = 0
    for name, f, poses in work:
        n_lig = f.shape[0]
        plain = PoseBatcher(rec.cuda(), torch.cat([f, rec_feats], 0), n_lig, 1, edge_radius=6.0)
        for k in range(poses.shape[0]):
            with torch.no_grad():
                want = torch.sigmoid(model(plain.load(poses[k:k + 1])).reshape(-1))[0]
            assert abs(float(got[name][k, 0]) - float(want)) < 1e-5 * max(1.0, abs(float(want))), (name, k)
            assert lines[at] == f'{float(got[name][k, 0]):.3f} | receptor {name}_pose{k}'
            at += 1
