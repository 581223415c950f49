"""(GPU) Which torch operators launch device work inside one training step (the library's own kernels aside):
torch.profiler over three cfg2 steps, operators with device time, and the Python line that called each.
Usage: python tools/step_ops.py [--config cfg2]"""
import argparse
import os
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
os.environ.setdefault('PVS_EGNN_KEEP_DEAD_COORDS', '1')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--config', default='cfg2')
    ap.add_argument('--batch', type=int, default=32)
    args = ap.parse_args()
    from pointvs_amd.egnn_satorras import SartorrasEGNN
    from pointvs_amd.synthetic import CONFIGS, synthetic_batch
    cfg = CONFIGS[args.config]
    torch.manual_seed(0)
    model = SartorrasEGNN(Path('/tmp/pvs_ops'), 2e-3, 1e-4, silent=True, **cfg['model']).train()
    batch = synthetic_batch(cfg['cfg_id'], args.batch, **cfg['graph']).to('cuda')
    y = batch.y.float()

    def step():
        pred = model(batch).reshape(-1)
        loss = model.get_loss(y, pred)
        model.optimiser.zero_grad()
        loss.backward()
        model.optimiser.step(clip_value=1.0)

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        for _ in range(3):
            step()
        torch.cuda.synchronize()
    print(prof.key_averages(group_by_stack_n=4).table(sort_by='cuda_time_total', row_limit=40, max_name_column_width=60,
                                                      max_src_column_width=90))


if __name__ == '__main__':
    main()
