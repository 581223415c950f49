"""Debug aid for the dynamic-range test: the worst edges of edge_feat at 2^+-20 (H = 32, default flags)."""
import os
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from oracle import egnn_oracle as orc  # noqa: E402
from pointvs_amd.egnn_satorras import EGNNLayer  # noqa: E402
from pointvs_amd.graph import Batch  # noqa: E402
from pointvs_amd.synthetic import synthetic_graph  # noqa: E402

hid, R, seed = 32, 20, 5
torch.manual_seed(seed)
layer = EGNNLayer(hid, hid, hid, edges_in_d=3).cuda()
g = Batch.from_data_list([synthetic_graph(900 + seed, n_nodes=500, n_lig=16, edge_radius=6.0)])
n = g.x.shape[0]
rng = np.random.default_rng(seed)
u = rng.uniform(-1, 1, size=(n, 1))
row_scale = torch.from_numpy(np.exp2(R * u))
h_plain = torch.from_numpy(rng.normal(size=(n, hid)).astype(np.float32))
h_wide = (h_plain.double() * row_scale).float()
sd = {'L.' + k: v.detach().cpu().double() for k, v in layer.state_dict().items()}
kw = dict(orc.BUILD_NET_DEFAULTS, residual=True, normalize=False, tanh=False, graphnorm=False)
kw['edge_attention_here'] = False
kw['node_attention_here'] = False
h2, _, m2, _, _ = orc.egnn_layer(sd, 'L.', kw, h_wide.double(), g.edge_index, g.pos.double(), g.edge_attr, None)
m_ref = m2.numpy()
for fam, env in (('f16x2', None), ('fp32', '0')):
    if env:
        os.environ['PVS_EGNN_BF16X3'] = env
    with torch.no_grad():
        _, _, _, m1 = layer(h_wide.cuda(), g.edge_index.cuda(), g.pos.cuda(), g.edge_attr.cuda())
    os.environ.pop('PVS_EGNN_BF16X3', None)
    m = m1.cpu().numpy().astype(np.float64)
    scale = np.abs(m_ref).max(1)
    err = np.abs(m - m_ref).max(1) / np.maximum(scale, 1e-300)
    worst = np.argsort(-err)[:6]
    print(fam, 'max row-rel', err.max())
    for e in worst:
        i, j = int(g.edge_index[0, e]), int(g.edge_index[1, e])
        c = int(np.abs(m[e] - m_ref[e]).argmax())
        print(f'  edge {e} ({i}->{j}) log2 scale i {R * u[i, 0]:+.1f} j {R * u[j, 0]:+.1f}  row max {scale[e]:.3e}  err {err[e]:.2e}  '
              f'ch {c}: got {m[e, c]:.9e} ref {m_ref[e, c]:.9e}  |m_ref| sorted top3 {np.sort(np.abs(m_ref[e]))[-3:]}')
