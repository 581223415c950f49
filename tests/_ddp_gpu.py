"""Rank body of tests/test_gpu_multiprocess.py: the REAL SartorrasEGNN training step (HIP kernels,
FusedClipAdam) under OverlappedGradAllReducer, two ranks sharing cuda:0 over gloo. Started from a
fork server that never touched the GPU (tests/conftest.py), so no process that has initialised HIP
is ever forked or replaced."""
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent

MODEL_KW = dict(dim_input=12, k=32, dim_output=1, num_layers=3, residual=False, edge_residual=False,
                edge_attention=False, normalize=False, tanh=False, dropout=0.0, graphnorm=False,
                update_coords=True, permutation_invariance=False, node_attention=False,
                gated_residual=False, rezero=False, softmax_attention=False, model_task='classification')
GRAPH_KW = dict(n_nodes=300, n_lig=20, edge_radius=6.0)
N_GRAPHS, PER_RANK, STEPS, SEED = 16, 4, 3, 5


def dataset():
    from pointvs_amd.synthetic import synthetic_graph
    return [synthetic_graph(9000 + k, **GRAPH_KW) for k in range(N_GRAPHS)]


def rank_main(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      HSA_ENABLE_IPC_MODE_LEGACY='0')
    if str(ROOT) not in sys.path:
        sys.path.insert(0, str(ROOT))
    import numpy as np
    import torch
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from pointvs_amd.data_loaders import GraphLoader, RankWeightedSampler
        from pointvs_amd.distributed import OverlappedGradAllReducer
        from pointvs_amd.egnn_satorras import SartorrasEGNN
        torch.cuda.set_device(0)
        torch.manual_seed(0)
        model = SartorrasEGNN(Path(out_dir) / f'm{rank}', 2e-3, 1e-4, silent=True, **MODEL_KW).train()
        params = list(model.parameters())
        names = [n for n, _ in model.named_parameters()]
        reducer = OverlappedGradAllReducer(params)
        model.grad_sync = None
        sampler = RankWeightedSampler(None, N_GRAPHS, rank, world, seed=SEED, shuffle=True)
        loader = GraphLoader(dataset(), batch_size=PER_RANK, sampler=sampler, device='cuda')
        rec = {}
        step = 0
        for epoch in range(2):
            sampler.set_epoch(epoch)
            rec[f'order{epoch}'] = np.array(list(sampler))
            for batch in loader:
                if step >= STEPS:
                    break
                y_pred, y_true, _, _ = model.unpack_input_data_and_predict(batch)
                loss = model.get_loss(y_true.cuda(), y_pred)
                model.optimiser.zero_grad()
                loss.backward()
                for n, p in zip(names, params):
                    rec[f's{step}/local/{n}'] = (np.zeros(0) if p.grad is None else p.grad.detach().cpu().numpy().copy())
                    rec[f's{step}/none/{n}'] = np.array(p.grad is None)
                reducer(weight=batch.num_graphs)
                for n, p in zip(names, params):
                    rec[f's{step}/reduced/{n}'] = (np.zeros(0) if p.grad is None else p.grad.detach().cpu().numpy().copy())
                    rec[f's{step}/none_after/{n}'] = np.array(p.grad is None)
                model.optimiser.step(clip_value=1.0)
                step += 1
        reducer.check()
        for n, p in zip(names, params):
            rec[f'final/{n}'] = p.detach().cpu().numpy().copy()
        torch.cuda.synchronize()
        np.savez(Path(out_dir) / f'rank{rank}.npz', **rec)
    finally:
        dist.destroy_process_group()


# ---- BASELINE config 4 on one GPU: cfg2-size graphs, fixed global batch of 8 split 4 + 4 -------------------
CFG4_GLOBAL, CFG4_STEPS = 8, 2


def cfg4_graphs():
    from pointvs_amd.synthetic import CONFIGS, synthetic_graph
    cfg = CONFIGS['cfg2']
    return [synthetic_graph(1000 * cfg['cfg_id'] + 40 + k, **cfg['graph']) for k in range(CFG4_GLOBAL)]


def rank_main_cfg4(rank, world, port, out_dir):
    """Config 2's model on whole config-2 graphs (2000 atoms, r = 10 A), data parallel: the global batch of 8
    is dealt 4 + 4, gradients go through OverlappedGradAllReducer (step 0: the flat exchange that learns the
    layout; step 1: hook-driven buckets overlapped with the backward), the update is FusedClipAdam's."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      HSA_ENABLE_IPC_MODE_LEGACY='0')
    if str(ROOT) not in sys.path:
        sys.path.insert(0, str(ROOT))
    import numpy as np
    import torch
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from pointvs_amd.distributed import OverlappedGradAllReducer, shard_range
        from pointvs_amd.egnn_satorras import SartorrasEGNN
        from pointvs_amd.graph import Batch
        from pointvs_amd.optim import FusedClipAdam
        from pointvs_amd.synthetic import CONFIGS
        torch.cuda.set_device(0)
        torch.manual_seed(0)
        model = SartorrasEGNN(Path(out_dir) / f'm{rank}', 2e-3, 1e-4, silent=True, **CONFIGS['cfg2']['model']).train()
        assert isinstance(model.optimiser, FusedClipAdam)
        params = list(model.parameters())
        names = [n for n, _ in model.named_parameters()]
        reducer = OverlappedGradAllReducer(params)
        lo, hi = shard_range(CFG4_GLOBAL, rank, world)
        batch = Batch.from_data_list(cfg4_graphs()[lo:hi]).to('cuda')
        rec = {}
        for step in range(CFG4_STEPS):
            for n, p in zip(names, params):
                rec[f's{step}/weights/{n}'] = p.detach().cpu().numpy().copy()
            y_pred, y_true, _, _ = model.unpack_input_data_and_predict(batch)
            loss = model.get_loss(y_true.cuda(), y_pred)
            model.optimiser.zero_grad()
            loss.backward()
            reducer(weight=batch.num_graphs)
            rec[f's{step}/overlapped'] = np.array(reducer._buckets is not None and step > 0)
            for n, p in zip(names, params):
                rec[f's{step}/reduced/{n}'] = (np.zeros(0) if p.grad is None else p.grad.detach().cpu().numpy().copy())
            model.optimiser.step(clip_value=1.0)
        reducer.check()
        for n, p in zip(names, params):
            rec[f'final/{n}'] = p.detach().cpu().numpy().copy()
        torch.cuda.synchronize()
        np.savez(Path(out_dir) / f'rank{rank}.npz', **rec)
    finally:
        dist.destroy_process_group()


def rccl_single_rank_bench(out_path):
    """`bench.py --gpus 1 --force-dist` in a process of its own: a process group of ONE rank on the 'nccl'
    backend (= RCCL) runs every collective of the multi-rank bench path - the hook-driven bucketed all-reduce,
    the barriers, the max-over-ranks reduction of the time."""
    import contextlib
    import io
    os.environ.update(HSA_ENABLE_IPC_MODE_LEGACY='0', MASTER_ADDR='127.0.0.1')
    os.environ.pop('PVS_BENCH_BACKEND', None)
    if str(ROOT) not in sys.path:
        sys.path.insert(0, str(ROOT))
    import bench
    sys.argv = ['bench.py', '--gpus', '1', '--steps', '3', '--warmup', '2', '--batch', '2', '--force-dist',
                '--no-cpu-baseline']
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        bench.main()
    Path(out_path).write_text(buf.getvalue())


def rccl_single_rank_reducer(out_path):
    """GradAllReducer over a 1-rank 'nccl' group: loads RCCL and runs its all-reduce on device buffers."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29541', RANK='0', WORLD_SIZE='1',
                      HSA_ENABLE_IPC_MODE_LEGACY='0')
    if str(ROOT) not in sys.path:
        sys.path.insert(0, str(ROOT))
    import json
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    try:
        from pointvs_amd.distributed import GradAllReducer
        gen = torch.Generator().manual_seed(3)
        params = [torch.nn.Parameter(torch.randn(5, 3, generator=gen).cuda()),
                  torch.nn.Parameter(torch.randn(7, generator=gen).cuda()),
                  torch.nn.Parameter(torch.randn(2, 2, generator=gen).cuda())]
        grads = [torch.randn(5, 3, generator=gen).cuda(), None, torch.randn(2, 2, generator=gen).cuda()]
        for p, g in zip(params, grads):
            p.grad = None if g is None else g.clone()
        red = GradAllReducer(params, exchange_when_alone=True)
        red(weight=4)
        red(weight=4)
        red.check()
        same = all((p.grad is None) if g is None else torch.allclose(p.grad, g, rtol=1e-6, atol=0)
                   for p, g in zip(params, grads))
        Path(out_path).write_text(json.dumps(dict(backend=dist.get_backend(), same=bool(same),
                                                  exchanged=red._bucket is not None)))
    finally:
        dist.destroy_process_group()
