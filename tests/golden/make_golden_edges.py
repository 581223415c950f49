"""Golden vectors for the radius-graph builder: runs the REFERENCE generate_edges
(/root/reference/point_vs/preprocessing/preprocessing.py:68) in the build container on seeded random
structures and stores inputs + outputs. Usage (container only): python tests/golden/make_golden_edges.py"""
import sys
from pathlib import Path

import numpy as np
import pandas as pd

OUT = Path(__file__).resolve().parent
sys.path.insert(0, str(OUT / '_refstubs'))   # import-only stand-ins for pymol/rdkit/... (see README there)
sys.path.insert(0, '/root/reference')
if not hasattr(np, 'product'):
    np.product = np.prod
from point_vs.preprocessing.preprocessing import generate_edges   # noqa: E402


def make_struct(seed, n, n_lig, box):
    rng = np.random.default_rng(seed)
    xyz = (rng.random((n, 3)) * box).astype(np.float32)
    # a few exactly coincident and exactly-at-radius pairs (the 1e-7 and `<` edge cases)
    if n > 8:
        xyz[5] = xyz[2]
        xyz[7] = xyz[3] + np.array([2.0, 0.0, 0.0], dtype=np.float32)
    bp = np.ones(n, dtype=np.int64)
    bp[rng.permutation(n)[:n_lig]] = 0
    return xyz, bp


CASES = [  # name, seed, n, n_lig, box, inter, intra, prune
    ('edges_small', 1, 40, 8, 6.0, 2.0, 1.5, False),
    ('edges_small_prune', 2, 60, 10, 12.0, 2.5, 1.5, True),
    ('edges_default_radii', 3, 300, 25, 18.0, 4.0, 2.0, True),
    ('edges_r10', 4, 500, 30, 24.0, 10.0, 10.0, False),
    ('edges_bonds', 5, 400, 30, 14.0, 6.0, 2.0, False),
    ('edges_no_inter', 6, 50, 0, 8.0, 3.0, 2.0, True),
]

for name, seed, n, n_lig, box, inter, intra, prune in CASES:
    xyz, bp = make_struct(seed, n, n_lig, box)
    df = pd.DataFrame({'x': xyz[:, 0], 'y': xyz[:, 1], 'z': xyz[:, 2], 'types': 6, 'bp': bp})
    df['orig'] = np.arange(n)
    out, (rows, cols), attrs = generate_edges(df.copy(), inter_radius=inter, intra_radius=intra, prune=prune)
    np.savez_compressed(OUT / f'{name}.npz', xyz=xyz, bp=bp, inter=inter, intra=intra, prune=prune,
                        keep=out['orig'].to_numpy(), rows=np.asarray(rows), cols=np.asarray(cols),
                        attrs=np.asarray(attrs))
    print(name, 'n', n, 'kept', len(out), 'edges', len(rows))
