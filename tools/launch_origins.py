"""Which host operation launches each small device kernel of one eager cfg2 training step (torch.profiler with Python
stacks): the torch-side launches (fills, copies) that the C-ABI kernels' own timeline does not explain.
usage (GPU box): python tools/launch_origins.py [cfg2|cfg3|real4A]"""
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
os.environ['PVS_EGNN_KEEP_DEAD_COORDS'] = '1'
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

from pointvs_amd import graph as pgraph  # noqa: E402
from pointvs_amd.egnn_satorras import SartorrasEGNN  # noqa: E402
from pointvs_amd.synthetic import CONFIGS, synthetic_batch  # noqa: E402

cfg = CONFIGS[sys.argv[1] if len(sys.argv) > 1 else 'cfg2']
pgraph.CACHE_ENABLED = False
batch = synthetic_batch(cfg['cfg_id'], 32, **cfg['graph']).to('cuda')
y_true = batch.y.float()
torch.manual_seed(0)
model = SartorrasEGNN(Path('/tmp/pvs_origins'), 2e-3, 1e-4, silent=True, **cfg['model']).train()


def step():
    y = model(batch).reshape(-1)
    loss = model.get_loss(y_true, y)
    model.optimiser.zero_grad()
    loss.backward()
    model.optimiser.step(clip_value=1.0)
    return loss


for _ in range(5):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True,
             experimental_config=torch._C._profiler._ExperimentalConfig(verbose=True)) as prof:
    step()
    torch.cuda.synchronize()
seen = set()
for c in prof.events():
    for kk in getattr(c, 'kernels', []) or []:
        name = kk.name
        if name.startswith(('k_', 'void (anonymous', '(anonymous')) and 'at::' not in name:
            continue                      # the library's own kernels: explained by the C-ABI call that launched them
        key = (name[:60], c.name, tuple(c.stack[:1]) if c.stack else ())
        if key in seen:
            continue
        seen.add(key)
        print(f'{name[:72]:72s} {kk.duration:7.1f} us  <- {c.name}')
        for fr in (c.stack or [])[:8]:
            print('        ', fr)

print()
for ev in prof.key_averages(group_by_stack_n=12):
    if ev.key in ('aten::copy_', 'aten::fill_', 'aten::index_select', 'aten::to', 'aten::_to_copy', 'aten::clone', 'aten::contiguous') and ev.device_time_total > 0:
        print(f'{ev.key}  x{ev.count}  device {ev.device_time_total:.1f} us')
        for fr in ev.stack[:12]:
            print('        ', fr)
