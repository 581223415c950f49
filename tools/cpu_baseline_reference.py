#!/usr/bin/env python3
"""SURVEY 8d "CPU baseline" leg (b): the REAL reference and the CPU oracle timed on the SAME BASELINE-shaped graph.

BUILD CONTAINER ONLY - it imports /root/reference (under the import-only stand-ins of tests/golden/_refstubs for the
third parties this image lacks) and never travels to the GPU box; bench.py's `cpu_baseline` leg times the oracle there
(kind "port"). This tool shows that the oracle is a representative stand-in for the reference's own eager-PyTorch CPU
path: one cfg2 graph (3 layers, 32 channels, 2000 atoms, r = 10 A) and one cfg3 graph (12 layers, 64 channels, edge +
node attention, r = 6 A) from pointvs_amd/synthetic.py, the same initial weights on both sides, and per side 2 warm-up +
5 timed training steps of the reference's own loop body (point_neural_network_base.py:176-199 / 417-429: forward, BCE,
zero_grad, backward, clip_grad_value_(1.0), Adam lr 2e-3 wd 1e-4), median, thread count stated.

    python tools/cpu_baseline_reference.py [--threads 8] > profiles/r06_cpu_reference_vs_oracle.txt
"""
import argparse
import os
import sys
import tempfile
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / 'tests' / 'golden'))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--threads', type=int, default=os.cpu_count() or 1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    args = ap.parse_args()
    cwd = os.getcwd()
    import make_golden as mg                 # reference on sys.path under the stubs (it chdirs into /root/reference)
    os.chdir(cwd)
    import numpy as np
    import torch
    from point_vs.models.geometric.egnn_satorras import SartorrasEGNN as RefEGNN
    from torch_geometric.data import Batch as RefBatch, Data as RefData          # (stubs)
    from oracle import egnn_oracle as orc
    from pointvs_amd.graph import Batch
    from pointvs_amd.synthetic import CONFIGS, synthetic_graph
    torch.set_num_threads(args.threads)
    print(f'# reference (/root/reference under tests/golden/_refstubs) vs CPU oracle (oracle/egnn_oracle.py), one graph, '
          f'fwd + BCE + bwd + clip + Adam; {args.warmup} warm-up + {args.steps} timed steps, median; '
          f'{args.threads} torch threads on a {os.cpu_count()}-CPU container, torch {torch.__version__}')
    for name in ('cfg2', 'cfg3'):
        cfg = CONFIGS[name]
        g = synthetic_graph(1000 * cfg['cfg_id'], **cfg['graph'])
        with tempfile.TemporaryDirectory() as tmp:
            torch.manual_seed(0)
            ref = RefEGNN(Path(tmp), 2e-3, 1e-4, None, None, silent=True, **cfg['model'])
            sd0 = {k: v.detach().clone().numpy() for k, v in ref.state_dict().items()}
            batch = RefBatch.from_data_list([RefData(
                x=g.x, edge_index=g.edge_index, edge_attr=g.edge_attr, pos=g.pos, y=g.y.reshape(1),
                lig_fname=g.lig_fname, rec_fname=g.rec_fname, dE=None, rmsd=None)])
            ref.train()
            ref.eta = '0'

            def ref_step():
                y_pred, y_true, _, _ = ref.unpack_input_data_and_predict(mg.clone_graph(batch))
                return float(ref.backprop(y_true, y_pred))

            gb = Batch.from_data_list([g])
            ocfg = dict(cfg['model'], _class='SartorrasEGNN')
            state = {'sd': sd0}

            def oracle_step():
                y, loss, grads = orc.forward_backward(state['sd'], ocfg, gb.x, gb.pos, gb.edge_index, gb.edge_attr,
                                                      gb.batch, gb.y.float())
                new = orc.adam_step(state['sd'], grads, 2e-3, 1e-4)
                state['sd'] = {k: (new[k].numpy() if k in new else v) for k, v in state['sd'].items()}
                return float(loss)

            rows = []
            for label, fn in (('reference', ref_step), ('oracle', oracle_step)):
                losses, times = [], []
                for it in range(args.warmup + args.steps):
                    t0 = time.perf_counter()
                    losses.append(fn())
                    times.append(time.perf_counter() - t0)
                med = float(np.median(times[args.warmup:]))
                rows.append((label, med, losses))
            n, e = int(g.x.shape[0]), int(g.edge_index.shape[1])
            print(f'\n{name}: N = {n}, E = {e}, {cfg["model"]["num_layers"]} layers, {cfg["model"]["k"]} channels')
            for label, med, losses in rows:
                print(f'  {label:9s} {med * 1e3:8.1f} ms/step = {1.0 / med:6.3f} graphs/s   losses '
                      + ' '.join(f'{v:.6f}' for v in losses))
            (_, mr, lr_), (_, mo, lo) = rows
            # (the oracle's optimiser leg is a FIRST Adam step every time - oracle.adam_step carries no moments, the same
            # arithmetic volume - so only the first two losses are the same numbers: same weights, then one identical step)
            drift = max(abs(a - b) / max(abs(a), 1e-12) for a, b in zip(lr_[:2], lo[:2]))
            print(f'  oracle / reference time = {mo / mr:.3f}; first two losses (same weights, then one identical Adam '
                  f'step) agree to {drift:.1e} relative; later ones differ by design (the oracle\'s optimiser leg '
                  f'restarts its moments every step: timing only)')


if __name__ == '__main__':
    main()
