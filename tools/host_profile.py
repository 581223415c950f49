"""Where the HOST time of an eager training step goes at the launch-bound shape (bench.py --config real4A: the reference's
CLI defaults). cProfile over 200 steps, sorted by own time and by cumulative time.   usage (GPU box): python tools/host_profile.py"""
import cProfile
import io
import os
import pstats
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
os.environ['PVS_EGNN_KEEP_DEAD_COORDS'] = '1'
import torch  # noqa: E402

from pointvs_amd import graph as pgraph  # noqa: E402
from pointvs_amd.egnn_satorras import SartorrasEGNN  # noqa: E402
from pointvs_amd.synthetic import CONFIGS, synthetic_batch  # noqa: E402

cfg = CONFIGS[sys.argv[1] if len(sys.argv) > 1 else 'real4A']
pgraph.CACHE_ENABLED = False
batch = synthetic_batch(cfg['cfg_id'], 32, **cfg['graph']).to('cuda')
y_true = batch.y.float()
torch.manual_seed(0)
model = SartorrasEGNN(Path('/tmp/pvs_hostprof'), 2e-3, 1e-4, silent=True, **cfg['model']).train()


def step():
    y = model(batch).reshape(-1)
    loss = model.get_loss(y_true, y)
    model.optimiser.zero_grad()
    loss.backward()
    model.optimiser.step(clip_value=1.0)


for _ in range(20):
    step()
torch.cuda.synchronize()
import time  # noqa: E402
t0 = time.perf_counter()
for _ in range(200):
    step()
torch.cuda.synchronize()
print(f'{(time.perf_counter() - t0) / 200 * 1e3:.3f} ms per step, unprofiled')
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    step()
torch.cuda.synchronize()
pr.disable()
for key in ('tottime', 'cumulative'):
    out = io.StringIO()
    pstats.Stats(pr, stream=out).sort_stats(key).print_stats(28)
    print(out.getvalue()[:6000])
