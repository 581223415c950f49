"""Time the GPU radius-graph builder against pvs_graph_prepare on the cfg2 batch."""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from pointvs_amd.graph import prepare_graph   # noqa: E402
from pointvs_amd.radius_graph import radius_graph   # noqa: E402
from pointvs_amd.synthetic import CONFIGS, synthetic_batch   # noqa: E402

cfg = CONFIGS[sys.argv[1] if len(sys.argv) > 1 else 'cfg2']
b = synthetic_batch(cfg['cfg_id'], 32, **cfg['graph']).to('cuda')
r = cfg['graph']['edge_radius']
bp = b.x[:, -1].contiguous()


def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


print('E', b.edge_index.shape[1])
print('prepare_graph (COO, resident)  %.3f ms' % timeit(lambda: prepare_graph(b.edge_index, b.edge_attr, b.x.shape[0])))
print('radius_graph (pos -> CSR/CSC)  %.3f ms' % timeit(lambda: radius_graph(b.pos, bp, b.ptr, inter_radius=r, max_graph_nodes=max(b.graph_node_counts))))
