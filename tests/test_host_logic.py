"""CPU-side checks: C-ABI export surface, class surface / state_dict contract, synthetic inputs,
loud failure without a GPU, and the data-parallel gradient exchange on gloo (world_size 2)."""
import ctypes
import os
import re
import socket
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent


def test_library_exports_every_declared_symbol():
    from pointvs_amd import _lib
    header = (ROOT / 'include' / 'pvs_egnn.h').read_text()
    declared = set(re.findall(r'\b(pvs_[a-z0-9_]+)\s*\(', header))
    assert declared, 'no declarations parsed'
    assert _lib.LIB_PATH.exists(), 'run __graft_entry__.build() first'
    handle = ctypes.CDLL(str(_lib.LIB_PATH))
    for name in sorted(declared):
        assert hasattr(handle, name), f'{name} declared in pvs_egnn.h but not exported'
    assert declared == set(_lib.EXPORTED_SYMBOLS), declared ^ set(_lib.EXPORTED_SYMBOLS)
    assert _lib.lib().pvs_version() >= 100


def test_state_dict_contract_and_seeded_init_match_reference():
    from pointvs_amd.egnn_multitask import MultitaskSatorrasEGNN
    from pointvs_amd.egnn_satorras import SartorrasEGNN
    from tests._golden import CASES, GoldenCase
    for name in CASES:
        c = GoldenCase(name)
        torch.manual_seed(c.meta['seed'])
        cls = SartorrasEGNN if c.meta['class'] == 'SartorrasEGNN' else MultitaskSatorrasEGNN
        model = cls(Path('/tmp/pvs_t'), c.meta['lr'], c.meta['wd'], None, None, silent=True,
                    **c.meta['kwargs'])
        sd = model.state_dict()
        assert list(sd.keys()) == list(c.sd.keys()), name
        for k, v in sd.items():
            assert np.array_equal(v.cpu().numpy(), c.sd[k]), (name, k)
        assert [n for n, _ in model.named_parameters()] == c.meta['param_order']


def test_param_counts_match_survey():
    from pointvs_amd.egnn_satorras import SartorrasEGNN
    from pointvs_amd.synthetic import CONFIGS
    m2 = SartorrasEGNN(Path('/tmp/pvs_t'), 2e-3, 1e-4, silent=True, **CONFIGS['cfg2']['model'])
    m3 = SartorrasEGNN(Path('/tmp/pvs_t'), 2e-3, 1e-4, silent=True, **CONFIGS['cfg3']['model'])
    assert m2.param_count == 22913      # SURVEY.md §8a row a1 [probe]
    assert m3.param_count == 354201


@pytest.mark.skipif(torch.cuda.is_available(), reason='checks the no-GPU behaviour')
def test_no_cpu_fallback():
    """The product path must fail loudly on CPU tensors instead of computing somewhere else."""
    from pointvs_amd.egnn_satorras import EGNNLayer
    layer = EGNNLayer(16, 16, 16, edges_in_d=3)
    h = torch.randn(5, 16)
    ei = torch.tensor([[0, 1, 2], [1, 2, 3]])
    ea = torch.nn.functional.one_hot(torch.tensor([0, 1, 2]), 3)
    with pytest.raises(RuntimeError, match='HIP device only'):
        layer(h, ei, torch.randn(5, 3), ea)


def test_product_never_imports_the_oracle():
    for path in (ROOT / 'pointvs_amd').rglob('*.py'):
        text = path.read_text()
        assert 'import oracle' not in text and 'from oracle' not in text, path


def test_synthetic_graph_follows_reference_edge_rule():
    from pointvs_amd.synthetic import synthetic_graph
    g = synthetic_graph(2000, n_nodes=300, n_lig=20, edge_radius=6.0)
    ei, et = g.edge_index.numpy(), g.edge_attr.argmax(1).numpy()
    assert g.edge_attr.dtype == torch.int64 and bool((g.edge_attr.sum(1) == 1).all())
    bp = g.x[:, 11].numpy().astype(int)
    d = np.linalg.norm(g.pos.numpy()[ei[0]] - g.pos.numpy()[ei[1]], axis=1)
    assert (d < 6.0 + 1e-4).all() and (d > 0).all()
    n_inter = int((et == 1).sum())
    # block 1 = inter-molecular pairs, row-major; block 2 = all pairs, row-major
    assert (bp[ei[0, :n_inter]] != bp[ei[1, :n_inter]]).all()
    for lo, hi in ((0, n_inter), (n_inter, ei.shape[1])):
        keys = ei[0, lo:hi].astype(np.int64) * 300 + ei[1, lo:hi]
        assert (np.diff(keys) > 0).all()
    blk2 = et[n_inter:]
    both_rec = (bp[ei[0, n_inter:]] == 1) & (bp[ei[1, n_inter:]] == 1)
    assert (blk2[both_rec] == 2).all() and (blk2[~both_rec] == 0).all()
    # every inter pair appears twice, graph is symmetric
    pairs = set(zip(ei[0].tolist(), ei[1].tolist()))
    assert all((b, a) in pairs for a, b in pairs)
    assert ei.shape[1] - len(pairs) == n_inter
    g2 = synthetic_graph(2000, n_nodes=300, n_lig=20, edge_radius=6.0)
    assert torch.equal(g.edge_index, g2.edge_index) and torch.equal(g.pos, g2.pos)


def test_batch_collation():
    from pointvs_amd.graph import Batch
    from pointvs_amd.synthetic import synthetic_graph
    gs = [synthetic_graph(s, n_nodes=50, n_lig=5, edge_radius=5.0) for s in (1, 2, 3)]
    b = Batch.from_data_list(gs)
    assert b.num_graphs == 3 and b.ptr.tolist() == [0, 50, 100, 150]
    assert b.batch.tolist() == [0] * 50 + [1] * 50 + [2] * 50
    off = gs[0].edge_index.shape[1]
    assert torch.equal(b.edge_index[:, off:off + gs[1].edge_index.shape[1]], gs[1].edge_index + 50)
    assert b.y.tolist() == [1, 0, 1]


def test_shard_range_partitions_all_graphs():
    from pointvs_amd.distributed import shard_range
    for n, w in ((256, 8), (33, 4), (5, 8)):
        spans = [shard_range(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))


def _ddp_worker(rank, world, port, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        sys.path.insert(0, str(ROOT))
        from pointvs_amd.distributed import GradAllReducer
        torch.manual_seed(0)
        params = [torch.nn.Parameter(torch.randn(4, 3)), torch.nn.Parameter(torch.randn(5)),
                  torch.nn.Parameter(torch.randn(2, 2))]
        gen = torch.Generator().manual_seed(100 + rank)
        params[0].grad = torch.randn(4, 3, generator=gen)
        params[1].grad = None                         # "last layer coord_mlp": stays None
        params[2].grad = torch.randn(2, 2, generator=gen)
        red = GradAllReducer(params)
        red()
        red()   # second call reuses the bucket; averaging already-equal values changes nothing
        out[rank] = (params[0].grad.clone(), params[1].grad, params[2].grad.clone())
    finally:
        dist.destroy_process_group()


def test_grad_allreduce_gloo_world2():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_ddp_worker, args=(2, port, out), nprocs=2, join=True)
    expect0 = sum(torch.randn(4, 3, generator=torch.Generator().manual_seed(100 + r))
                  for r in range(2)) / 2
    for rank in range(2):
        g0, g1, g2 = out[rank]
        assert g1 is None
        assert torch.allclose(g0, expect0, atol=1e-6)
    assert torch.equal(out[0][2], out[1][2])


def _overlap_worker(rank, world, port, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        sys.path.insert(0, str(ROOT))
        from pointvs_amd.distributed import GradAllReducer, OverlappedGradAllReducer
        res = {}
        for name, cls in (('flat', GradAllReducer), ('overlap', OverlappedGradAllReducer)):
            torch.manual_seed(0)
            net = torch.nn.Sequential(torch.nn.Linear(6, 8), torch.nn.SiLU(), torch.nn.Linear(8, 8),
                                      torch.nn.SiLU(), torch.nn.Linear(8, 1))
            unused = torch.nn.Parameter(torch.zeros(3))        # never receives a gradient
            params = list(net.parameters()) + [unused]
            red = cls(params)
            steps = []
            for step in range(3):
                x = torch.randn(5, 6, generator=torch.Generator().manual_seed(10 * step + rank))
                for p in params:
                    p.grad = None
                net(x).square().sum().backward()
                red()
                steps.append([None if p.grad is None else p.grad.clone() for p in params])
            res[name] = steps
        out[rank] = res
    finally:
        dist.destroy_process_group()


def test_overlapped_allreduce_matches_flat_gloo_world2():
    """Bucketed exchange started from the backward hooks == the flat exchange, on every step."""
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_overlap_worker, args=(2, port, out), nprocs=2, join=True)
    for rank in range(2):
        for a, b in zip(out[rank]['flat'], out[rank]['overlap']):
            for ga, gb in zip(a, b):
                assert (ga is None) == (gb is None)
                if ga is not None:
                    assert torch.allclose(ga, gb, atol=1e-6)
    for ga, gb in zip(out[0]['overlap'][-1], out[1]['overlap'][-1]):
        assert ga is None or torch.equal(ga, gb)


def _accum_worker(rank, world, port, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        sys.path.insert(0, str(ROOT))
        from pointvs_amd.distributed import GradAllReducer, OverlappedGradAllReducer
        res = {}
        n_local = 3 if rank == 0 else 2          # uneven shards: 5 "graphs" over 2 ranks
        for name in ('flat', 'overlap_nosync', 'overlap_double'):
            torch.manual_seed(0)
            net = torch.nn.Sequential(torch.nn.Linear(6, 8), torch.nn.SiLU(), torch.nn.Linear(8, 1))
            params = list(net.parameters())
            red = GradAllReducer(params) if name == 'flat' else OverlappedGradAllReducer(params)
            steps = []
            for step in range(3):
                for p in params:
                    p.grad = None
                xs = [torch.randn(n_local, 6, generator=torch.Generator().manual_seed(100 * step + 10 * rank + k))
                      for k in range(2)]           # two micro-batches per optimiser step
                if name == 'overlap_nosync':
                    with red.no_sync():
                        net(xs[0]).square().mean().backward()
                    net(xs[1]).square().mean().backward()
                else:                              # 'overlap_double': second backward without no_sync()
                    net(xs[0]).square().mean().backward()
                    net(xs[1]).square().mean().backward()
                red(weight=n_local)
                steps.append([p.grad.clone() for p in params])
            res[name] = steps
        # what single-process training on the global batch would give: mean over all 5 rows of each
        # micro-batch, summed over the two micro-batches
        torch.manual_seed(0)
        net = torch.nn.Sequential(torch.nn.Linear(6, 8), torch.nn.SiLU(), torch.nn.Linear(8, 1))
        single = []
        for step in range(3):
            net.zero_grad()
            for k in range(2):
                x = torch.cat([torch.randn(n, 6, generator=torch.Generator().manual_seed(100 * step + 10 * r + k))
                               for r, n in ((0, 3), (1, 2))])
                net(x).square().mean().backward()
            single.append([p.grad.clone() for p in net.parameters()])
        res['single'] = single
        out[rank] = res
    finally:
        dist.destroy_process_group()


def test_gradient_accumulation_and_uneven_shards_gloo_world2():
    """(1) a second backward before reducer() must not drop gradients, with and without
    no_sync(); (2) ranks holding 3 and 2 graphs give the gradient of the GLOBAL mean loss."""
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_accum_worker, args=(2, port, out), nprocs=2, join=True)
    for rank in range(2):
        for name in ('flat', 'overlap_nosync', 'overlap_double'):
            for got, want in zip(out[rank][name], out[rank]['single']):
                for a, b in zip(got, want):
                    assert torch.allclose(a, b, atol=1e-6), name


def _mismatch_worker(rank, world, port, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        sys.path.insert(0, str(ROOT))
        from pointvs_amd.distributed import OverlappedGradAllReducer
        torch.manual_seed(0)
        net = torch.nn.Sequential(torch.nn.Linear(4, 4), torch.nn.SiLU(), torch.nn.Linear(4, 1))
        extra = torch.nn.Parameter(torch.zeros(2))
        params = list(net.parameters()) + [extra]
        red = OverlappedGradAllReducer(params)
        raised = []
        for step in range(3):
            for p in params:
                p.grad = None
            y = net(torch.randn(3, 4)).sum()
            if step == 2 and rank == 1:          # only rank 1's gradient set changes
                y = y + extra.sum()
            y.backward()
            try:
                red()
                raised.append(False)
            except RuntimeError:
                raised.append(True)
        out[rank] = raised
    finally:
        dist.destroy_process_group()


def test_changed_gradient_set_raises_on_every_rank_gloo_world2():
    """the 'set of parameters changed' error must reach all ranks (no hang)."""
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_mismatch_worker, args=(2, port, out), nprocs=2, join=True)
    assert out[0] == [False, False, True] and out[1] == [False, False, True]


def _straggler_worker(rank, world, port, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        sys.path.insert(0, str(ROOT))
        import time
        from pointvs_amd.distributed import GradAllReducer, OverlappedGradAllReducer
        res = {}
        for name in ('flat', 'overlap'):
            torch.manual_seed(0)
            net = torch.nn.Sequential(torch.nn.Linear(6, 8), torch.nn.SiLU(), torch.nn.Linear(8, 8), torch.nn.SiLU(),
                                      torch.nn.Linear(8, 8), torch.nn.SiLU(), torch.nn.Linear(8, 1))
            params = list(net.parameters())
            launched = []
            if name == 'overlap':
                if rank == 1:
                    # rank 1's backward is slow where it matters: the hooks of the LAST layers (the first gradients
                    # to land, the first bucket) run late, so rank 0 has launched bucket 0 - and reached reducer()
                    # with the rest - long before rank 1 launches anything. Registered BEFORE the reducer's own
                    # hooks: hooks of a parameter run in registration order.
                    for p in params[-2:] + params[2:4]:
                        p.register_post_accumulate_grad_hook(lambda _p: time.sleep(0.3))
                red = OverlappedGradAllReducer(params, n_buckets=3)
            else:
                red = GradAllReducer(params)
            steps = []
            for step in range(4):
                x = torch.randn(5, 6, generator=torch.Generator().manual_seed(10 * step + rank))
                for p in params:
                    p.grad = None
                if name == 'overlap' and red._buckets is not None:
                    import pointvs_amd.distributed as D
                    orig = D._Bucket.launch

                    def spy(self, weight, flag, group, _orig=orig, _red=red):
                        launched.append(_red._buckets.index(self))
                        return _orig(self, weight, flag, group)
                    D._Bucket.launch = spy
                    try:
                        net(x).square().sum().backward()
                        red()
                    finally:
                        D._Bucket.launch = orig
                else:
                    net(x).square().sum().backward()
                    red()
                steps.append([p.grad.clone() for p in params])
            res[name] = steps
            res[name + '_order'] = list(launched)
        out[rank] = res
    finally:
        dist.destroy_process_group()


def test_bucket_order_is_the_same_on_a_delayed_rank_gloo_world2():
    """collectives are matched by ISSUE ORDER, so the buckets of the overlapped exchange must be
    launched in one order on every rank however the ranks' backward passes are timed. Rank 1's gradient hooks are
    delayed by 0.3 s each: both ranks must launch buckets 0, 1, 2 in that order on every step, finish within 60 s (no
    hang) and hold the same reduced gradients - the flat exchange's."""
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    out = mgr.dict()
    ctx = mp.start_processes(_straggler_worker, args=(2, port, out), nprocs=2, join=False, start_method='spawn')
    import time
    deadline = time.time() + 60
    while not ctx.join(timeout=1.0):
        if time.time() > deadline:
            for p in ctx.processes:
                p.kill()
            raise AssertionError('the delayed-rank exchange did not finish within 60 s (ranks stuck in a collective)')
    for rank in range(2):
        assert out[rank]['overlap_order'] == [0, 1, 2] * 3, out[rank]['overlap_order']     # (step 0: the flat exchange)
        for a, b in zip(out[rank]['flat'], out[rank]['overlap']):
            for ga, gb in zip(a, b):
                assert torch.allclose(ga, gb, atol=1e-6)
    for a, b in zip(out[0]['overlap'], out[1]['overlap']):
        for ga, gb in zip(a, b):
            assert torch.equal(ga, gb)


def test_fused_clip_adam_falls_back_to_torch_on_cpu_tensors():
    """No GPU here: FusedClipAdam must behave exactly like clip_grad_value_ + torch.optim.Adam."""
    from pointvs_amd.optim import FusedClipAdam
    torch.manual_seed(0)
    a = [torch.nn.Parameter(torch.randn(7, 3)), torch.nn.Parameter(torch.randn(5))]
    b = [torch.nn.Parameter(p.detach().clone()) for p in a]
    oa = FusedClipAdam(a, lr=2e-3, weight_decay=1e-4)
    ob = torch.optim.Adam(b, lr=2e-3, weight_decay=1e-4)
    for step in range(3):
        for pa, pb in zip(a, b):
            g = torch.randn(pa.shape, generator=torch.Generator().manual_seed(step)) * 3
            pa.grad, pb.grad = g.clone(), g.clone()
        oa.step(clip_value=1.0)
        torch.nn.utils.clip_grad_value_(b, 1.0)
        ob.step()
    for pa, pb in zip(a, b):
        assert torch.equal(pa, pb)
    assert set(oa.state_dict()['state'][0]) == set(ob.state_dict()['state'][0])


def test_rank_sampler_shards_one_seeded_draw():
    """data_loaders.py:181-186 under data parallelism: the union of the ranks' shares is the draw a
    single process's WeightedRandomSampler makes with the same generator; equal length on all ranks."""
    from pointvs_amd.data_loaders import GraphLoader, RankWeightedSampler, class_balance_weights
    from pointvs_amd.synthetic import synthetic_graph
    labels = [0] * 9 + [1] * 4
    w = class_balance_weights(labels)
    assert torch.allclose(w[:9], torch.full((9,), 1 / 9, dtype=torch.double)) and float(w[-1]) == 0.25
    assert class_balance_weights([1, 1, 1]) is None
    ranks = [RankWeightedSampler(w, rank=r, world=3, seed=11) for r in range(3)]
    for epoch in (0, 1):
        for s in ranks:
            s.set_epoch(epoch)
        ref = list(torch.utils.data.WeightedRandomSampler(
            w, len(w), generator=torch.Generator().manual_seed(11 + epoch)))
        shares = [list(s) for s in ranks]
        assert len({len(x) for x in shares}) == 1 and len(shares[0]) == len(ranks[0]) == 5
        padded = ref + ref[:2]
        assert all(shares[r] == padded[r::3] for r in range(3))
    data = [synthetic_graph(s, n_nodes=40, n_lig=4, edge_radius=5.0) for s in range(13)]
    batches = list(GraphLoader(data, batch_size=2, sampler=ranks[0]))
    assert [b.num_graphs for b in batches] == [2, 2, 1] and len(GraphLoader(data, 2, ranks[0])) == 3


class _TwoHeadToy:
    """Built lazily (needs the package on sys.path inside the spawned ranks): a two-headed model on the
    REAL harness (PointNeuralNetworkBase.train_model / val / save / set_task) with CPU tensors."""

    @staticmethod
    def make(save_path, silent):
        from torch import nn
        from pointvs_amd.point_neural_network_base import PointNeuralNetworkBase

        class Toy(PointNeuralNetworkBase):
            def build_net(self, **kw):
                self.head_pose = nn.Linear(4, 1)
                self.head_affinity = nn.Linear(4, 1)
                return nn.Sequential(nn.Linear(3, 4), nn.SiLU())

            def unpack_input_data_and_predict(self, item):
                x, y, names = item
                body = self.layers(x)
                head = self.head_pose if self.model_task == 'classification' else self.head_affinity
                return head(body).reshape(-1), y, names, ['rec'] * len(names)

        torch.manual_seed(0)
        return Toy(save_path, 1e-2, 1e-4, silent=silent, model_task='classification')


def _harness_worker(rank, world, port, tmp, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        sys.path.insert(0, str(ROOT))
        from pointvs_amd.distributed import OverlappedGradAllReducer, shard_range
        model = _TwoHeadToy.make(Path(tmp) / 'run', silent=rank != 0)
        model.grad_sync = OverlappedGradAllReducer(list(model.parameters()))
        model.log_interval = 2
        gen = torch.Generator().manual_seed(7)
        xs, ys = torch.randn(22, 3, generator=gen), (torch.arange(22) % 2).float()
        names = [f'lig{i:02d}' for i in range(22)]

        def loader(lo, hi, step=3):
            return [(xs[k:min(k + step, hi)], ys[k:min(k + step, hi)], names[k:min(k + step, hi)])
                    for k in range(lo, hi, step)]
        lo, hi = shard_range(22, rank, world)
        train = loader(rank * 6, rank * 6 + 6)          # three steps of two... equal step count on both ranks
        model.set_task('classification')
        model.train_model(train, epochs=2, epoch_end_validation_set=loader(lo, hi))
        model.val(loader(lo, hi))
        model.set_task('regression')                      # the other head: the exchange must re-plan
        model.train_model(train, epochs=1, epoch_end_validation_set=loader(lo, hi))
        model.val(loader(lo, hi))
        model.grad_sync.check()
        out[rank] = {n: p.detach().clone() for n, p in model.named_parameters()}
    finally:
        dist.destroy_process_group()


def test_two_rank_harness_writes_whole_files_and_switches_heads_gloo_world2(tmp_path):
    """ADVICE r2 (high + medium): under data parallelism every rank validates its own share - the
    predictions file must come out complete and in data-set order (per-rank parts joined by rank 0),
    checkpoints are written by rank 0 only, and `--model_task both` (pose head, then affinity head)
    must not trip the gradient exchange when the set of parameters with gradients changes."""
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_harness_worker, args=(2, port, str(tmp_path), out), nprocs=2, join=True)
    for n in out[0]:                                         # replicas stayed in lockstep through both phases
        assert torch.equal(out[0][n], out[1][n]), n
    run = tmp_path / 'run'
    files = sorted(p.name for p in run.iterdir() if p.suffix == '.txt')
    # (epoch-end validation runs after every epoch BUT the last, point_neural_network_base.py:481 `epoch < epochs`)
    assert files == ['affinity_predictions.txt', 'pose_predictions.txt', 'pose_predictions_epoch_1.txt'], files
    assert not list(run.glob('*.rank*')) and not list(run.glob('*.joining'))
    for f in files:
        lines = (run / f).read_text().splitlines()
        assert [ln.split()[-1] for ln in lines] == [f'lig{i:02d}' for i in range(22)], f
    ckpts = sorted(p.name for p in (run / 'checkpoints').iterdir())
    assert ckpts == ['affinity_ckpt_epoch_1.pt', 'pose_ckpt_epoch_1.pt', 'pose_ckpt_epoch_2.pt']
    for c in ckpts:
        torch.load(run / 'checkpoints' / c, map_location='cpu', weights_only=False)     # whole, readable


def test_rank_sampler_without_weights_iterates_in_index_order():
    """weights=None is the reference's sampler=None, shuffle=False (data_loaders.py:176-178,
    512-520): index order every epoch, strided over ranks, padded by wrapping."""
    from pointvs_amd.data_loaders import RankWeightedSampler
    ranks = [RankWeightedSampler(None, 7, r, 2, seed=3) for r in range(2)]
    for epoch in (0, 1):
        for s in ranks:
            s.set_epoch(epoch)
        assert list(ranks[0]) == [0, 2, 4, 6] and list(ranks[1]) == [1, 3, 5, 0]
    assert list(RankWeightedSampler(None, 5)) == [0, 1, 2, 3, 4]
    shuffled = RankWeightedSampler(None, 50, seed=3, shuffle=True)
    assert sorted(shuffled) == list(range(50)) and list(shuffled) != list(range(50))


def test_runs_layout_refuses_tables_that_do_not_match_the_edge_list():
    """an edge list edited after collation no longer matches its per-graph counts;
    the merge-of-runs preparation trusts those tables, so such a batch must take the sort path."""
    from pointvs_amd.graph import Batch, runs_layout
    from pointvs_amd.synthetic import synthetic_graph
    items = [synthetic_graph(s, n_nodes=40, n_lig=4, edge_radius=5.0) for s in range(3)]
    batch = Batch.from_data_list(items)
    node_ptr, edge_ptr = runs_layout(batch)
    assert node_ptr.tolist() == [0, 40, 80, 120] and int(edge_ptr[-1]) == batch.edge_index.size(1)
    keep = torch.ones(batch.edge_index.size(1), dtype=torch.bool)
    keep[5] = False
    batch.edge_index = batch.edge_index[:, keep]            # filtered after collation: counts are stale
    batch.edge_attr = batch.edge_attr[keep]
    assert runs_layout(batch) is None
    batch2 = Batch.from_data_list(items)
    batch2.x = batch2.x[:-1]                                 # node table no longer covers the nodes
    assert runs_layout(batch2) is None


def test_layer_parameter_cache_is_validated_slot_by_slot_and_never_copied():
    """the per-layer cache of the parameters' ctypes struct must (1) not travel with copy.deepcopy / pickle
    (a struct of pointers cannot be pickled; EMA / SWA copies a model that has run), (2) notice a parameter replaced
    INSIDE a leaf module, a replaced leaf module and a replaced container - none of which passes through the layer's
    own __setattr__. The probes are plain dict look-ups, so all of this is host logic (the struct itself needs a GPU:
    tests/test_gpu_host_contracts.py runs the same through the kernels)."""
    import copy
    import pickle
    from pointvs_amd.egnn_satorras import EGNNLayer
    layer = EGNNLayer(32, 32, 32, edges_in_d=3, edge_attention=True, node_attention=True, graphnorm=True,
                      gated_residual=True, edge_residual=True)
    params = layer._params()
    probes = layer._slot_probes(params)
    assert probes is not None and len(probes) == sum(p is not None for p in params) == 20

    def still_valid(pr):
        for mods, cont, seq, idx, leaf, name, p in pr:
            if mods is None:
                if leaf.get(name) is not p:
                    return False
            elif mods.get(cont) is not seq or seq._modules.get(idx) is not leaf or leaf._parameters.get(name) is not p:
                return False
        return True
    assert still_valid(probes)
    old = layer.node_mlp[0].weight
    layer.node_mlp[0].weight = torch.nn.Parameter(old.detach().clone())          # nested parameter
    assert not still_valid(probes)
    probes = layer._slot_probes(layer._params())
    layer.edge_mlp[2] = torch.nn.Linear(32, 32)                                   # leaf module
    assert not still_valid(probes)
    probes = layer._slot_probes(layer._params())
    layer.coord_mlp = torch.nn.Sequential(*list(layer.coord_mlp))                 # container (same leaves)
    assert not still_valid(probes)
    probes = layer._slot_probes(layer._params())
    layer.edge_gate_parameter = torch.nn.Parameter(torch.ones(1))                 # the layer's own parameter
    assert not still_valid(probes)
    # a weight that is not a registered parameter (parametrizations compute it on access): nothing to cache
    torch.nn.utils.parametrizations.weight_norm(layer.node_mlp[3])
    assert layer._slot_probes(layer._params()) is None

    layer = EGNNLayer(32, 32, 32, edges_in_d=3)
    layer.__dict__['_pcache'] = ('stands in for the ctypes struct', ctypes.c_void_p(1))
    layer.__dict__['_att_src'] = lambda: torch.zeros(1)
    twin = copy.deepcopy(layer)
    assert '_pcache' not in twin.__dict__ and twin.att_val is None and '_pcache' in layer.__dict__
    again = pickle.loads(pickle.dumps(layer))
    assert '_pcache' not in again.__dict__ and [k for k, _ in again.named_parameters()] == [k for k, _ in layer.named_parameters()]


def test_checkpoints_written_under_capture_hold_the_optimisers_own_form(tmp_path):
    """while train_model(capture=True) runs, the replayer has every optimiser group flipped to capturable
    with device step counters; a checkpoint written at an epoch end in between must carry the groups' OWN flags (and host
    counters: tests/test_gpu_training_trajectory.py checks those on the device), like the reference's files."""
    from pointvs_amd.egnn_satorras import SartorrasEGNN
    from pointvs_amd.synthetic import CONFIGS
    model = SartorrasEGNN(tmp_path, 2e-3, 1e-4, silent=True, **CONFIGS['cfg2']['model'])
    for p in model.parameters():
        p.grad = torch.zeros_like(p)
    model.optimiser.step()
    plain = model._optimiser_state_for_checkpoint()
    assert not any(g.get('capturable') for g in plain['param_groups'])
    model._capturable_was = [g.get('capturable', False) for g in model.optimiser.param_groups]       # what _StepReplayer does
    for g in model.optimiser.param_groups:
        g['capturable'] = True
    under = model._optimiser_state_for_checkpoint()
    assert not any(g.get('capturable') for g in under['param_groups'])
    assert all(g['capturable'] for g in model.optimiser.param_groups)             # the live optimiser is untouched
    assert under['state'].keys() == plain['state'].keys()
    model.save(tmp_path / 'ck.pt')
    ck = torch.load(tmp_path / 'ck.pt', weights_only=False)
    assert not any(g.get('capturable') for g in ck['optimiser_state_dict']['param_groups'])


def test_stack_plan_lays_the_gradients_of_all_layers_out_in_one_buffer():
    """functional.StackPlan / _GradLayout (the one-call layer stack, pvs_egnn_stack_bwd): every live parameter gradient gets a
    16-byte aligned range of ONE buffer in the order of `plan.params`; which gradients exist follows the per-layer path -
    a layer's coord_mlp only when its coordinates fed something (every layer but the last, and the last one only when a
    gradient arrives for its coordinates), the edge gate never (no edge residual in a stack). Pure host logic."""
    import copy
    from pointvs_amd import _lib
    from pointvs_amd import functional as PF
    from pointvs_amd.egnn_satorras import EGNNLayer
    from pointvs_amd.optim import FusedClipAdam
    torch.manual_seed(0)
    layers = [EGNNLayer(32, 32, 32, edges_in_d=3, edge_attention=True, node_attention=(k == 1), residual=True, rezero=True)
              for k in range(3)]
    descs = [l._desc() for l in layers]
    params = [l._params() for l in layers]
    plan = PF.StackPlan(descs, params, [_lib.PvsLayerParams() for _ in layers])
    assert plan.n_layers == 3 and plan.hidden == 32 and plan.any_eatt and plan.any_natt
    assert len(plan.params) == sum(p is not None for ps in params for p in ps) == len(plan.index)
    fields = _lib.PARAM_FIELDS
    for live_last in (False, True):
        lay = plan.grad_layout(live_last)
        assert lay is plan.grad_layout(live_last)
        taken, cursor = [], 0
        sizes = list(lay.sizes)
        for (layer, slot), p, (k, shape) in zip(plan.index, plan.params, lay.take):
            name = fields[slot]
            dead = (name.startswith('coord_') and layer == 2 and not live_last) or name == 'edge_gate'
            assert (k < 0) == dead, (layer, name, k)
            if k >= 0:
                assert sizes[k] == p.numel() and (shape is None) == (p.dim() == 1) and (shape is None or shape == tuple(p.shape))
                taken.append(k)
        assert taken == sorted(taken)
        # every gradient starts at a multiple of 4 floats; the pads are the entries nobody takes
        for k, size in enumerate(sizes):
            if k in taken:
                assert cursor % 4 == 0, (k, cursor)
            cursor += size
        assert cursor == lay.total
        structs = lay.structs(4096)
        assert structs is lay.structs(4096) and len(structs) == 3
        assert structs[0].edge_w1 == 4096 and structs[2].edge_gate is None
        assert (structs[2].coord_w1 is None) == (not live_last)
    # an optimiser copy (EMA / snapshot of a whole model) starts with its own empty upload ring and no work list
    opt = FusedClipAdam([torch.nn.Parameter(torch.zeros(3))], lr=1e-3)
    twin = copy.deepcopy(opt)
    assert twin._ring == [] and twin._fast is None and twin._recent == {}


def test_training_loops_run_with_the_long_lived_heap_frozen(monkeypatch):
    """point_neural_network_base.long_lived_heap_frozen (train_model, ScreeningSweep.run, bench.py's timed region): what
    exists when the loop starts is moved out of the collector's sight (no 60-170 ms full collection over torch's ~170,000
    import-time objects inside the loop), and handed back afterwards; PVS_GC_FREEZE=0 leaves the collector alone."""
    import gc
    from pointvs_amd.point_neural_network_base import long_lived_heap_frozen
    before = gc.get_freeze_count()
    with long_lived_heap_frozen():
        inside = gc.get_freeze_count()
        assert inside > before + 10000          # (torch's modules alone are far more)
        junk = [[k] for k in range(1000)]       # young objects are still collected: the collector stays enabled
        assert gc.isenabled()
        del junk
    assert gc.get_freeze_count() == 0 or gc.get_freeze_count() <= before
    monkeypatch.setenv('PVS_GC_FREEZE', '0')
    with long_lived_heap_frozen():
        assert gc.get_freeze_count() <= before


def test_frozen_heap_nests():
    """Validation at the end of an epoch runs inside a training run: the inner loop must not thaw the outer one's heap."""
    import gc
    from pointvs_amd.point_neural_network_base import long_lived_heap_frozen
    with long_lived_heap_frozen():
        outer = gc.get_freeze_count()
        with long_lived_heap_frozen():
            assert gc.get_freeze_count() == outer
        assert gc.get_freeze_count() == outer > 0
    assert gc.get_freeze_count() == 0
