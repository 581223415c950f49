"""Fuzz of graph preparation (csrc/graph_prepare.hip): random batches in the reference loader's layout (generate_edges-ordered
graphs of random size, ligand size, radius and density, 1-40 graphs per batch, incl. single-atom ligands, complete graphs and
graphs with very few edges) through pvs_graph_prepare_runs - merge by counting, counting transpose, by-column placement by
direct scatter (PVS_CSC_TILES=0), through LDS-sorted tiles (=2) and by the launcher's own choice - against the general radix-sort
path (no layout tag), array for array (rowptr, row, col, etype, perm, inv_deg, colptr, cedge), with and without the by-column
lists, plus the definition of the by-column lists itself (stable order by column). Round 6 added the LDS-sorted tiles; the
suite's cases are six fixed batches.   usage (GPU box): python tools/fuzz_prepare.py [first_seed] [n_seeds]"""
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from pointvs_amd.graph import Batch, prepare_graph, runs_layout  # noqa: E402
from pointvs_amd.synthetic import synthetic_graph  # noqa: E402

NAMES = ('rowptr', 'row', 'col', 'etype', 'perm', 'inv_deg', 'colptr', 'cedge')


def run_seed(seed):
    rng = np.random.default_rng(31337 + seed)
    style = int(rng.integers(4))
    n_graphs = int(rng.integers(1, 41)) if style else 1
    items = []
    for k in range(n_graphs):
        if style == 3:          # many small, dense graphs (chunks shorter than a row)
            n, r = int(rng.integers(4, 90)), float(rng.choice([6.0, 12.0, 30.0]))
        elif style == 0:        # one large graph
            n, r = int(rng.integers(500, 4096)), float(rng.choice([3.0, 4.0, 6.0, 8.0, 10.0, 12.0]))
        else:
            n, r = int(rng.integers(8, 2600)), float(rng.choice([2.0, 3.0, 4.0, 6.0, 8.0, 10.0]))
        n_lig = int(rng.integers(1, min(n, 65)))
        items.append(synthetic_graph(50000 + 100 * seed + k, n_nodes=n, n_lig=n_lig, edge_radius=r,
                                     density=float(rng.choice([0.02, 0.05, 0.1]))))
    batch = Batch.from_data_list(items).to('cuda')
    n = int(batch.x.shape[0])
    e = int(batch.edge_index.shape[1])
    if e == 0:
        return n_graphs, n, e, []
    layout = runs_layout(batch)
    problems = []
    ref = {}
    for need_backward in (True, False):
        b = prepare_graph(batch.edge_index, batch.edge_attr, n, need_backward=need_backward)
        b.check_status()
        ref[need_backward] = b
    col = ref[True].t['col'][:e].cpu().numpy().astype(np.int64)
    order = np.argsort(col, kind='stable')
    if not np.array_equal(ref[True].t['cedge'][:e].cpu().numpy(), order):
        problems.append('sort path: cedge is not the stable order by column')
    for tiles in ('0', '2', None):
        if tiles is None:
            os.environ.pop('PVS_CSC_TILES', None)
        else:
            os.environ['PVS_CSC_TILES'] = tiles
        try:
            for need_backward in (True, False):
                a = prepare_graph(batch.edge_index, batch.edge_attr, n, need_backward=need_backward, layout=layout)
                a.check_status()
                for name in NAMES[:6] + (NAMES[6:] if need_backward else ()):
                    if not torch.equal(a.t[name], ref[need_backward].t[name]):
                        problems.append(f'PVS_CSC_TILES={tiles} backward={need_backward}: {name} differs from the sort path')
        finally:
            os.environ.pop('PVS_CSC_TILES', None)
    return n_graphs, n, e, problems


if __name__ == '__main__':
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    bad, t0, most = 0, time.time(), 0
    for seed in range(first, first + count):
        n_graphs, n, e, problems = run_seed(seed)
        most = max(most, e)
        if problems:
            bad += 1
            print('FAIL', seed, problems[:4], 'graphs', n_graphs, 'nodes', n, 'edges', e, flush=True)
    print(f'done: {count} seeds from {first}, failures: {bad}, largest batch {most} edges, {time.time() - t0:.0f} s')
