"""Which part of the forward is not run-to-run reproducible? (GPU box; diagnostic)"""
import os
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / 'tests'))
sys.path.insert(0, str(ROOT))
import torch
import test_gpu_properties as t
from pointvs_amd.synthetic import CONFIGS, synthetic_batch

cfg = CONFIGS['cfg2']
model, _ = t.make_model(seed=11, **{k: v for k, v in cfg['model'].items() if k in t.BASE_KW})
g = synthetic_batch(cfg['cfg_id'], 8, **cfg['graph'])


def fwd():
    import copy
    gg = copy.copy(g)
    gg.__dict__ = dict(g.__dict__)
    gg = gg.to('cuda')
    with torch.no_grad():
        feats = model._embed_graph(gg) if hasattr(model, '_embed_graph') else None
        y = model(gg).reshape(-1)
    return y.cpu().numpy(), (None if feats is None else (feats[0] if isinstance(feats, tuple) else feats).detach().cpu().numpy())


for name, env in (('default', {}), ('no_runs', {'PVS_PREPARE_RUNS': '0'}), ('generic', {'PVS_EGNN_KERNELS': 'generic'}),
                  ('fp32', {'PVS_EGNN_BF16X3': '0'}), ('keep_dead', {'PVS_EGNN_KEEP_DEAD_COORDS': '1'})):
    os.environ.update(env)
    runs = [fwd() for _ in range(6)]
    for k in env:
        os.environ.pop(k)
    ny = sum(r[0].tobytes() != runs[0][0].tobytes() for r in runs[1:])
    nf = sum((r[1] is not None) and r[1].tobytes() != runs[0][1].tobytes() for r in runs[1:])
    print(f'{name:10s} y differs in {ny}/5 runs, node features in {nf}/5 runs', flush=True)

print('training mode (gpu_run): which runs differ from run 0 / from run 1')
for changes in (dict(), dict(k=64, edge_attention=True)):
    m2, _ = t.make_model(seed=11, **dict({k: v for k, v in cfg['model'].items() if k in t.BASE_KW}, **changes))
    runs = [t.gpu_run(m2, g) for _ in range(5)]
    print(changes, 'y vs run0:', [int(r[0].tobytes() != runs[0][0].tobytes()) for r in runs],
          'y vs run1:', [int(r[0].tobytes() != runs[1][0].tobytes()) for r in runs])
    bad = sorted({n for r in runs[2:] for n in r[1] if r[1][n] is not None and r[1][n].tobytes() != runs[1][1][n].tobytes()})
    print('   grads differing among runs 1..4:', bad[:6], len(bad))
