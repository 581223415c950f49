"""Which parts of a step survive hipGraph capture + instantiate + replay? Each probe runs in a child
process (a failing capture can take the process down)."""
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
PROBES = sys.argv[2:] or ['prepare', 'forward', 'fwd_bwd', 'optimizer']
SIZE = sys.argv[1] if len(sys.argv) > 1 else 'small'

CHILD = r'''
import os, sys, faulthandler
faulthandler.enable()
sys.path.insert(0, %r)
probe = sys.argv[1]
if probe.endswith('_generic'):
    os.environ['PVS_EGNN_KERNELS'] = 'generic'
if probe.endswith('_fp32'):
    os.environ['PVS_EGNN_BF16X3'] = '0'
import torch
from pathlib import Path
from pointvs_amd import graph as pgraph
from pointvs_amd.egnn_satorras import SartorrasEGNN
from pointvs_amd.synthetic import CONFIGS, synthetic_batch
cfg = CONFIGS['cfg2']
batch = (synthetic_batch(2, 32, **cfg['graph']) if sys.argv[2] == 'full'
         else synthetic_batch(2, 2, **dict(cfg['graph'], n_nodes=600))).to('cuda')
torch.manual_seed(0)
model = SartorrasEGNN(Path('/tmp/pvs_probe'), 2e-3, 1e-4, silent=True, **cfg['model']).train()
params = list(model.parameters())
model.optimiser = torch.optim.Adam(params, lr=2e-3, weight_decay=1e-4, capturable=True)
pgraph.CACHE_ENABLED = False
y_true = batch.y.float()

def body():
    if probe == 'prepare':
        return pgraph.prepare_graph(batch.edge_index, batch.edge_attr, batch.x.shape[0]).t['rowptr']
    if probe.startswith('forward'):
        with torch.no_grad():
            return model(batch)
    y = model(batch).reshape(-1)
    loss = model.get_loss(y_true, y)
    model.optimiser.zero_grad()
    loss.backward()
    if probe == 'optimizer':
        torch.nn.utils.clip_grad_value_(params, 1.0)
        model.optimiser.step()
    return loss

side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(2):
        body()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = body()
print('captured', flush=True)
g.replay(); g.replay()
torch.cuda.synchronize()
print('replayed', float(out.float().sum()), flush=True)
''' % str(ROOT)

for probe in PROBES:
    r = subprocess.run([sys.executable, '-c', CHILD, probe, SIZE], capture_output=True, text=True)
    tail = [l for l in (r.stdout + r.stderr).splitlines() if 'Warning' not in l and 'run_backward' not in l]
    print(f'{probe:18s} rc={r.returncode}  ' + ' | '.join(tail[-3:])[:300])
