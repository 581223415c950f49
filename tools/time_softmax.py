"""fwd+bwd time of the reference tests' MODEL_KWARGS (softmax edge attention, node attention,
GraphNorm, 6 layers, k=32; test/setup_and_params.py:72-87) on cfg2-shaped graphs.
Usage: [PVS_EGNN_KERNELS=generic] python tools/time_softmax.py [graphs]"""
import sys
import time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from pointvs_amd.egnn_satorras import SartorrasEGNN
from pointvs_amd.synthetic import CONFIGS, synthetic_batch

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
cfg = CONFIGS['cfg2']
dev = torch.device('cuda', 0)
torch.manual_seed(0)
kw = dict(k=32, num_layers=6, dropout=0, dim_input=12, dim_output=1, graphnorm=True, update_coords=True,
          node_attention=True, residual=True, edge_attention=True, softmax_attention=True)
model = SartorrasEGNN(Path('/tmp/pvs_sm'), 2e-3, 1e-4, silent=True, **kw).train()
batch = synthetic_batch(cfg['cfg_id'], B, **cfg['graph']).to(dev)
y = batch.y.float()


def step():
    loss = model.get_loss(y, model(batch).reshape(-1))
    model.zero_grad()
    loss.backward()
    return loss


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    loss = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 5
print(f'{B} graphs, E={batch.edge_index.shape[1]}: {dt * 1e3:.2f} ms per fwd+bwd = {B / dt:.0f} graphs/s, loss {float(loss):.6f}')
