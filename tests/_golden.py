"""Loader for the golden fixtures in tests/golden (made by tests/golden/make_golden.py)."""
import json
from pathlib import Path

import numpy as np
import torch

GOLDEN_DIR = Path(__file__).resolve().parent / 'golden'
CASES = sorted(p.stem for p in GOLDEN_DIR.glob('c[0-9]_*.npz'))   # model cases (edges_*.npz: radius graphs)


class GoldenCase:
    def __init__(self, name):
        self.name = name
        self.z = np.load(GOLDEN_DIR / f'{name}.npz')
        self.meta = json.loads(str(self.z['cfg']))
        self.cfg = dict(self.meta['kwargs'], _class=self.meta['class'])
        src = self.z
        if self.meta.get('sd_from'):
            src = np.load(GOLDEN_DIR / f"{self.meta['sd_from']}.npz")
        self.sd = {k[3:]: src[k] for k in src.files if k.startswith('sd/')}
        self.grads = {k[5:]: self.z[k] for k in self.z.files if k.startswith('grad/')}
        self.adam = {k[5:]: self.z[k] for k in self.z.files if k.startswith('adam/')}
        self.out = {k[4:]: self.z[k] for k in self.z.files if k.startswith('out/')}
        self.x = torch.from_numpy(self.z['in/x'])
        self.pos = torch.from_numpy(self.z['in/pos'])
        self.edge_index = torch.from_numpy(self.z['in/edge_index'].astype(np.int64))
        self.edge_type = torch.from_numpy(self.z['in/edge_type'].astype(np.int64))
        self.edge_attr = torch.nn.functional.one_hot(self.edge_type, 3)  # int64, as the loader
        self.batch = torch.from_numpy(self.z['in/batch'].astype(np.int64))
        self.y_true = torch.from_numpy(self.z['in/y_true'])
        self.n_graphs = int(self.batch.max()) + 1


def rel_err(a, b):
    """max|a-b| / max(1, max|b|): the relative bound SURVEY.md §8c prescribes."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(1.0, np.abs(b).max()))
