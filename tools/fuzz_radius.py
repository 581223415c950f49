"""Fuzz of the radius-graph builder (csrc/radius_graph.hip through pointvs_amd.radius_graph.generate_edges) against the CPU
oracle's generate_edges (oracle/generate_edges_oracle.py, pinned on the reference's own arrays and outputs): random point
clouds of 2-2500 atoms, inter / intra radii 1.5-12 A (intra smaller, equal or larger), ligands of 1 to n-1 atoms, and the
cases a `<` decision can get wrong - LATTICE clouds whose pair distances hit a radius exactly (3-4-5 triangles), coincident
atoms (d <= 1e-7: no edge), clouds far from the origin (fp32 coordinates, fp64 distances). Kept atoms, edge list (the
reference's order) and edge classes must be IDENTICAL arrays, with prune off and on.
usage (GPU box): python tools/fuzz_radius.py [first_seed] [n_seeds]"""
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from oracle.generate_edges_oracle import generate_edges as oracle_edges  # noqa: E402
from pointvs_amd.radius_graph import generate_edges  # noqa: E402


def run_seed(seed):
    rng = np.random.default_rng(271828 + seed)
    n = int(rng.integers(2, 2500)) if rng.integers(4) else int(rng.integers(2, 200))
    style = int(rng.integers(4))
    if style == 0:        # lattice: integer coordinates x spacing, many distances equal to a radius
        spacing = float(rng.choice([0.5, 1.0, 1.5]))
        pos = rng.integers(-6, 7, size=(n, 3)).astype(np.float32) * np.float32(spacing)
        radii = [spacing * k for k in (1, 2, 3, 4, 5)]
        inter, intra = float(rng.choice(radii)), float(rng.choice(radii))
    else:
        size = float(rng.choice([4.0, 8.0, 15.0, 30.0]))
        pos = (rng.normal(size=(n, 3)) * size).astype(np.float32)
        if style == 2:    # far from the origin
            pos += np.float32(rng.choice([1e3, 1e4])) * rng.normal(size=(1, 3)).astype(np.float32)
        if style == 3 and n > 4:    # coincident and nearly coincident atoms
            dup = rng.integers(0, n, size=max(1, n // 10))
            pos[dup] = pos[rng.integers(0, n, size=dup.size)]
            near = rng.integers(0, n, size=max(1, n // 20))
            pos[near] = pos[rng.integers(0, n, size=near.size)] + np.float32(1e-8)
        inter = float(rng.choice([1.5, 2.0, 4.0, 6.0, 10.0, 12.0]))
        intra = float(rng.choice([1.5, 2.0, 4.0, 6.0, 10.0]))
    n_lig = int(rng.integers(1, n)) if n > 1 else 1
    bp = np.ones(n, dtype=np.int64)
    bp[rng.permutation(n)[:n_lig]] = 0
    problems = []
    for prune in (False, True):
        keep_o, (r_o, c_o), a_o = oracle_edges(pos, bp, inter, intra, prune=prune)
        keep, ei, attrs = generate_edges(torch.from_numpy(pos).cuda(), torch.from_numpy(bp).cuda(), inter, intra, prune=prune)
        keep, ei, attrs = keep.cpu().numpy(), ei.cpu().numpy(), attrs.cpu().numpy()
        if not np.array_equal(keep, np.asarray(keep_o)):
            problems.append(f'prune={prune}: kept atoms differ ({len(keep)} vs {len(keep_o)})')
        elif ei.shape[1] != len(r_o):
            problems.append(f'prune={prune}: {ei.shape[1]} edges vs {len(r_o)}')
        elif not (np.array_equal(ei[0], r_o) and np.array_equal(ei[1], c_o) and np.array_equal(attrs, a_o)):
            problems.append(f'prune={prune}: edge list or classes differ')
    return (n, n_lig, inter, intra, style), problems


if __name__ == '__main__':
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    bad, t0 = 0, time.time()
    for seed in range(first, first + count):
        shape, problems = run_seed(seed)
        if problems:
            bad += 1
            print('FAIL', seed, problems, '(n, n_lig, inter, intra, style) =', shape, flush=True)
    print(f'done: {count} seeds from {first}, failures: {bad}, {time.time() - t0:.0f} s')
