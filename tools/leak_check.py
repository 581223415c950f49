"""Does a long eager training run grow? 4,000 steps at the reference's default shape (bench.py --config real4A: the one-call layer
stack, FusedClipAdam's remembered pointer tables, graph preparation every step), with device memory (allocated / reserved),
the process's resident set and the number of Python container objects sampled every 500 steps: all must be flat after the
first sample.   usage (GPU box): python tools/leak_check.py [steps]"""
import gc
import os
import resource
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
os.environ['PVS_EGNN_KEEP_DEAD_COORDS'] = '1'
import torch  # noqa: E402

from pointvs_amd import graph as pgraph  # noqa: E402
from pointvs_amd.egnn_satorras import SartorrasEGNN  # noqa: E402
from pointvs_amd.synthetic import CONFIGS, synthetic_batch  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
cfg = CONFIGS['real4A']
pgraph.CACHE_ENABLED = False
batches = [synthetic_batch(cfg['cfg_id'], 32, first_graph=32 * k, **cfg['graph']).to('cuda') for k in range(4)]
torch.manual_seed(0)
model = SartorrasEGNN(Path('/tmp/pvs_leak'), 2e-3, 1e-4, silent=True, **cfg['model']).train()


def rss_mb():
    with open('/proc/self/statm') as f:
        return int(f.read().split()[1]) * resource.getpagesize() / 2 ** 20


rows = []
for k in range(steps):
    b = batches[k % len(batches)]
    y = model(b).reshape(-1)
    loss = model.get_loss(b.y.float(), y)
    model.optimiser.zero_grad()
    loss.backward()
    model.optimiser.step(clip_value=1.0)
    if (k + 1) % 500 == 0:
        torch.cuda.synchronize()
        gc.collect()
        rows.append((k + 1, torch.cuda.memory_allocated() / 2 ** 20, torch.cuda.memory_reserved() / 2 ** 20, rss_mb(),
                     len(gc.get_objects()), float(loss.detach())))
        print('step %5d  device allocated %8.1f MiB  reserved %8.1f MiB  host RSS %8.1f MiB  tracked objects %7d  loss %.5f'
              % rows[-1], flush=True)
first, last = rows[1], rows[-1]          # (the first sample still holds warm-up growth: pools, pinned rings, caches)
grew = {'device allocated': last[1] - first[1], 'device reserved': last[2] - first[2], 'host RSS': last[3] - first[3],
        'tracked objects': last[4] - first[4]}
print('growth from step %d to step %d:' % (first[0], last[0]), {k: round(v, 1) for k, v in grew.items()})
ok = grew['device allocated'] < 1 and grew['device reserved'] < 1 and grew['host RSS'] < 16 and grew['tracked objects'] < 2000
print('FLAT' if ok else 'GROWING')
sys.exit(0 if ok else 1)
