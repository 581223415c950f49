"""One seed of the kernel-family fuzz (tests/test_gpu_properties.py::test_kernel_families_agree_on_random_configurations) with
every family's error against the fp64 ORACLE per tensor - which family is off, and by how much - instead of the test's
pass / fail against the generic kernels.   usage (GPU box): python tools/fuzz_seed_report.py <seed>"""
import os
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from tests import test_gpu_properties as t  # noqa: E402
from tests._golden import rel_err  # noqa: E402

seed = int(sys.argv[1])
rng = np.random.default_rng(1000 + seed)
flags = dict(
    k=int(rng.choice([32, 64])), num_layers=int(rng.integers(1, 4)),
    residual=bool(rng.integers(2)), edge_residual=bool(rng.integers(2)),
    edge_attention=bool(rng.integers(2)), node_attention=bool(rng.integers(2)),
    normalize=bool(rng.integers(2)), tanh=bool(rng.integers(2)), graphnorm=bool(rng.integers(2)),
    update_coords=bool(rng.integers(4) > 0), permutation_invariance=bool(rng.integers(4) == 0),
    attention_activation_fn=str(rng.choice(['sigmoid', 'tanh', 'relu', 'silu'])))
variant = int(rng.integers(3))
if flags['edge_attention'] and seed % 3 == 0:
    flags['softmax_attention'] = True
if variant == 1:
    flags['gated_residual'] = True
elif variant == 2:
    flags['rezero'] = True
model, kw = t.make_model(seed=seed, **flags)
n = int(rng.integers(40, 2500))
e = int(rng.integers(1, 40)) * n + int(rng.integers(0, 31))
g = t.random_graph(n, e, seed=seed, n_graphs=int(rng.integers(1, 5)))
print('flags', flags, 'n', n, 'e', e)
runs = {}
for name, env in (('mfma', {}), ('fp32', {'PVS_EGNN_BF16X3': '0'}), ('generic', {'PVS_EGNN_KERNELS': 'generic'})):
    os.environ.update(env)
    runs[name] = t.gpu_run(model, g)
    for k_ in env:
        os.environ.pop(k_, None)
y64, _, g64 = t.oracle_run(model, kw, g, dtype=torch.float64)
y32, _, g32 = t.oracle_run(model, kw, g, dtype=torch.float32)
print(f"{'tensor':40s} {'max|ref|':>10s} " + ' '.join(f'{k:>10s}' for k in ('mfma', 'fp32', 'generic', 'oracle32')))
print(f"{'logits':40s} {float(np.abs(y64.numpy()).max()):10.3e} " +
      ' '.join(f'{rel_err(runs[k][0], y64.numpy()):10.2e}' for k in ('mfma', 'fp32', 'generic')) + f' {rel_err(y32.numpy(), y64.numpy()):10.2e}')
for pname, ref in g64.items():
    if ref is None:
        continue
    r = ref.numpy()
    print(f'{pname:40s} {float(np.abs(r).max()):10.3e} ' +
          ' '.join(f'{rel_err(runs[k][1][pname], r):10.2e}' for k in ('mfma', 'fp32', 'generic')) + f' {rel_err(g32[pname].numpy(), r):10.2e}')
