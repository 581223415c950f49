"""Run-to-run bitwise reproducibility of outputs and gradients per golden case (GPU box)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch
from tests._golden import CASES, GoldenCase
from tests.test_gpu_parity import build_model, make_batch

for name in (sys.argv[1:] or CASES):
    c = GoldenCase(name)
    model = build_model(c)
    grads = []
    for _ in range(4):
        model.zero_grad()
        y, _, _, _ = model.unpack_input_data_and_predict(make_batch(c))
        model.get_loss(c.y_true.cuda(), y).backward()
        grads.append({n: p.grad.detach().cpu().numpy().copy() for n, p in model.named_parameters() if p.grad is not None})
    bad = sorted({n for g in grads[1:] for n in g if g[n].tobytes() != grads[0][n].tobytes()})
    print(f'{name:36s} {"OK" if not bad else "DIFF"}')
    for n in bad:
        d = max(float(np.abs(g[n].astype(np.float64) - grads[0][n]).max()) for g in grads[1:])
        print(f'      {n:40s} max|diff| {d:.3e}  max|g| {float(np.abs(grads[0][n]).max()):.3e}')
