"""Multi-GPU path on the PRODUCT, testable on one GPU: two fresh ranks share cuda:0 over gloo and run
the real SartorrasEGNN training step with OverlappedGradAllReducer (hook-driven buckets on the real
autograd Functions of the HIP path), the per-rank seeded sampler and FusedClipAdam."""
import multiprocessing as mp
import socket

import numpy as np
import pytest
import torch

from tests import _ddp_gpu as W
from tests._golden import CaseLog, assert_strict, grad_floor, rel_err

pytestmark = pytest.mark.gpu


def test_two_ranks_sharing_one_gpu_train_in_lockstep(tmp_path):
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    ctx = mp.get_context('forkserver')       # server started in conftest before the GPU was touched
    procs = [ctx.Process(target=W.rank_main, args=(r, 2, port, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=600)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    r0, r1 = (np.load(tmp_path / f'rank{r}.npz') for r in range(2))
    names = sorted(k[len('final/'):] for k in r0.files if k.startswith('final/'))

    # the sampler: ranks hold disjoint strided shares of ONE seeded draw, a different one per epoch
    for epoch in range(2):
        union = np.empty(W.N_GRAPHS, dtype=np.int64)
        union[0::2], union[1::2] = r0[f'order{epoch}'], r1[f'order{epoch}']
        assert sorted(union.tolist()) == list(range(W.N_GRAPHS))
    assert not np.array_equal(r0['order0'], r0['order1'])

    none_expected = None
    for step in range(W.STEPS):
        none_now = sorted(n for n in names if bool(r0[f's{step}/none/{n}']))
        for n in names:
            # the None set is preserved by the exchange, identically on both ranks
            assert bool(r0[f's{step}/none/{n}']) == bool(r1[f's{step}/none/{n}']) == \
                bool(r0[f's{step}/none_after/{n}']) == bool(r1[f's{step}/none_after/{n}']), n
            if n in none_now:
                continue
            mean = 0.5 * (r0[f's{step}/local/{n}'].astype(np.float64) + r1[f's{step}/local/{n}'])
            for r in (r0, r1):
                assert rel_err(r[f's{step}/reduced/{n}'], mean) < 1e-6, (step, n)
            assert np.array_equal(r0[f's{step}/reduced/{n}'], r1[f's{step}/reduced/{n}']), (step, n)
        none_expected = none_now if none_expected is None else none_expected
        assert none_now == none_expected
    assert len(none_expected) == 3          # the last layer's coord_mlp
    for n in names:                          # identical weights on both ranks after the optimiser steps
        assert np.array_equal(r0[f'final/{n}'], r1[f'final/{n}']), n

    # step 0 against ONE process on the union batch: mean of the rank gradients == global-batch gradient
    from pointvs_amd.egnn_satorras import SartorrasEGNN
    from pointvs_amd.graph import Batch
    data = W.dataset()
    torch.manual_seed(0)
    model = SartorrasEGNN(tmp_path / 'single', 2e-3, 1e-4, silent=True, **W.MODEL_KW).train()
    idx = np.concatenate([r0['order0'][:W.PER_RANK], r1['order0'][:W.PER_RANK]])
    batch = Batch.from_data_list([data[i] for i in idx]).to('cuda')
    y_pred, y_true, _, _ = model.unpack_input_data_and_predict(batch)
    model.get_loss(y_true.cuda(), y_pred).backward()
    for n, p in model.named_parameters():
        if p.grad is not None:
            assert rel_err(r0[f's0/reduced/{n}'], p.grad.cpu().numpy()) < 1e-5, n


def _run_ranks(target, n, tmp_path, timeout=900):
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    ctx = mp.get_context('forkserver')       # server started in conftest before the GPU was touched
    procs = [ctx.Process(target=target, args=(r, n, port, str(tmp_path))) for r in range(n)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=timeout)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]


def test_config4_two_ranks_on_baseline_graphs_match_the_fp64_oracle(tmp_path):
    """BASELINE config 4 (config 2 data parallel) at config-2 GRAPH size, on the product path: two ranks share
    cuda:0 over gloo, the global batch of 8 graphs (2000 atoms, r = 10 A) is split 4 + 4. The exchanged
    gradient - step 0 through the flat exchange, step 1 through the hook-driven overlapped buckets - must be
    the gradient of the mean loss over the GLOBAL batch = the mean of the eight per-graph fp64 ORACLE
    gradients evaluated at that step's weights (bound: SURVEY 8c, as tests/test_gpu_baseline_parity.py)."""
    from oracle import egnn_oracle as orc
    from pointvs_amd.graph import Batch
    from pointvs_amd.synthetic import CONFIGS
    _run_ranks(W.rank_main_cfg4, 2, tmp_path)
    r0, r1 = (np.load(tmp_path / f'rank{r}.npz') for r in range(2))
    names = sorted(k[len('final/'):] for k in r0.files if k.startswith('final/'))
    cfg = CONFIGS['cfg2']
    ocfg = dict(cfg['model'], _class='SartorrasEGNN')
    graphs = [Batch.from_data_list([g]) for g in W.cfg4_graphs()]
    assert not bool(r0['s0/overlapped']) and bool(r0['s1/overlapped']) and bool(r1['s1/overlapped'])
    for step in range(W.CFG4_STEPS):
        sd = {n: r0[f's{step}/weights/{n}'] for n in names}
        for n in names:                       # replicas in lockstep
            assert np.array_equal(r0[f's{step}/weights/{n}'], r1[f's{step}/weights/{n}']), (step, n)
        mean64, mean32 = {}, {}
        log = CaseLog(f'cfg4_two_ranks_step{step}')
        for g in graphs:
            y_true = g.y.float().reshape(-1)
            for dtype, acc in ((torch.float64, mean64), (torch.float32, mean32)):
                _, _, grads = orc.forward_backward(sd, ocfg, g.x, g.pos, g.edge_index, g.edge_attr, g.batch,
                                                   y_true, dtype=dtype)
                for k, v in grads.items():
                    if v is not None:
                        acc[k] = acc.get(k, 0.0) + v.numpy().astype(np.float64) / len(graphs)
        for n in names:
            got0, got1 = r0[f's{step}/reduced/{n}'], r1[f's{step}/reduced/{n}']
            if n not in mean64:               # the last layer's coord_mlp: None before and after the exchange
                assert got0.size == 0 and got1.size == 0, n
                continue
            assert np.array_equal(got0, got1), (step, n)
            bound = max(1e-5, 2.0 * rel_err(mean32[n], mean64[n]))
            assert rel_err(got0, mean64[n]) <= bound, (step, n, rel_err(got0, mean64[n]), bound)
            assert_strict(got0, mean64[n], mean32[n], f'{log.case} grad {n}', floor=grad_floor(mean64), log=log)
        log.finish()
    for n in names:
        assert np.array_equal(r0[f'final/{n}'], r1[f'final/{n}']), n


def _run_one(target, out_path, timeout=600):
    ctx = mp.get_context('forkserver')
    p = ctx.Process(target=target, args=(str(out_path),))
    p.start()
    p.join(timeout=timeout)
    assert p.exitcode == 0, p.exitcode


def test_rccl_all_reduce_on_a_process_group_of_one(tmp_path):
    """The 'nccl' backend (RCCL) has never run on a multi-GPU node in this project (no such node was leased):
    at least load it and run the gradient exchange's collective on device memory with world_size 1."""
    import json
    _run_one(W.rccl_single_rank_reducer, tmp_path / 'reducer.json')
    rec = json.loads((tmp_path / 'reducer.json').read_text())
    assert rec == dict(backend='nccl', same=True, exchanged=True), rec


def test_bench_multi_rank_path_runs_on_rccl_with_one_rank(tmp_path):
    """bench.py's multi-rank code (init_process_group('nccl', device_id=...), hook-driven bucketed all-reduce,
    barriers around the timed region, max-over-ranks of the time) executed end to end on RCCL."""
    import json
    _run_one(W.rccl_single_rank_bench, tmp_path / 'bench.txt')
    line = [ln for ln in (tmp_path / 'bench.txt').read_text().splitlines() if ln.startswith('{')][-1]
    rec = json.loads(line)
    assert rec['n_gpus'] == 1 and rec['value'] > 0 and rec['config']['launch'] == 'eager'
    assert rec['config']['arithmetic']          # says how the fp32 products are formed; wording not pinned
