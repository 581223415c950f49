"""train_model(capture=True) against the eager loop on a small in-memory dataset at the reference's default shape
(NOT a BASELINE configuration): 16 resident batches of 32 graphs x 500 atoms, r = 4 A, 6 layers, 32 channels; graphs/s
over epochs 3.. (every batch is replayed from its third visit on).   usage (GPU box): python tools/train_capture_bench.py"""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from pointvs_amd.egnn_satorras import SartorrasEGNN  # noqa: E402
from pointvs_amd.synthetic import CONFIGS, synthetic_batch  # noqa: E402

cfg = CONFIGS['real4A']
n_batches, epochs = 16, 12
loader = [synthetic_batch(cfg['cfg_id'], 32, first_graph=32 * k, **cfg['graph']).to('cuda') for k in range(n_batches)]
def run(capture, epochs):
    torch.manual_seed(0)
    model = SartorrasEGNN(Path('/tmp/pvs_capture_bench'), 2e-3, 1e-4, silent=True, **cfg['model'])
    model.only_save_best_models = True            # no checkpoint writes inside the timed epochs
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    losses = model.train_model(loader, epochs=epochs, capture=capture)
    torch.cuda.synchronize()
    return time.perf_counter() - t0, losses, getattr(model, 'last_capture_stats', None)


run(False, 2); run(True, 3)          # warm the process (first-use work of the library and of torch)
for capture in (False, True):
    t_short, _, _ = run(capture, 4)
    t_long, losses, stats = run(capture, 4 + epochs)
    steady = (t_long - t_short) / (epochs * n_batches)            # epochs 5 .. : every step a replay under capture
    print(f"capture={capture!s:5s}  {len(losses)} steps in {t_long:.3f} s; steady state {steady * 1e3:6.3f} ms/step = "
          f"{32 / steady:9.1f} graphs/s  (last loss {losses[-1]:.6f})  {stats or ''}")
