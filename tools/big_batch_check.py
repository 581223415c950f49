"""Batch independence at large batches: logits of graph k in a batch of B graphs == in a batch of 32.
Usage: python tools/big_batch_check.py [B ...]"""
import os
import sys
import time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from pointvs_amd.egnn_satorras import SartorrasEGNN
from pointvs_amd.synthetic import CONFIGS, synthetic_batch

cfg = CONFIGS['cfg2']
dev = torch.device('cuda', 0)
torch.manual_seed(0)
model = SartorrasEGNN(Path('/tmp/pvs_big'), 2e-3, 1e-4, silent=True, **cfg['model']).train()
ref = None
for B in [int(a) for a in sys.argv[1:]] or [32, 64, 128, 256]:
    batch = synthetic_batch(cfg['cfg_id'], B, **cfg['graph']).to(dev)
    with torch.no_grad():
        y = model(batch).reshape(-1).float().cpu()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = model(batch).reshape(-1)
    loss = model.get_loss(batch.y.float(), out)
    loss.backward()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    g = torch.cat([p.grad.reshape(-1) for p in model.parameters() if p.grad is not None])
    model.zero_grad()
    if ref is None:
        ref = y
    n = min(len(ref), len(y))
    print(f'B={B} E={batch.edge_index.shape[1]} logits[:3]={y[:3].tolist()} max|diff to B=32|={float((y[:n] - ref[:n]).abs().max()):.3e} '
          f'loss={float(loss):.5f} |grad|={float(g.norm()):.5f} nan={bool(torch.isnan(g).any())} fwd+bwd {dt * 1e3:.1f} ms')
    del batch
