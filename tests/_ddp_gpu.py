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
