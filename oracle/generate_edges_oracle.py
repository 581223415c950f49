"""CPU oracle for the radius-graph builder: a numpy restatement of
`generate_edges` (/root/reference/point_vs/preprocessing/preprocessing.py:68-155).

TEST INFRASTRUCTURE ONLY: imported by tests/ (and nothing else). The product path
(`pointvs_amd/radius_graph.py` -> `pvs_radius_graph_*` in libpvs_egnn.so) never touches it.

Pinned against (tests/test_oracle_golden.py):
  * the reference's own expected arrays in test/test_preprocessing_fns.py:32-71 (prune False/True),
    kept as data in tests/golden/generate_edges_reference_tests.json;
  * outputs of the reference function itself, run in the build container on random structures
    (tests/golden/make_golden_edges.py -> tests/golden/edges_*.npz).
"""
import numpy as np


def cdist_euclidean(coords):
    """scipy.spatial.distance.cdist(coords, coords, 'euclidean') (preprocessing.py:108): float64,
    s = sum_k (a_k - b_k)^2 accumulated in index order, then sqrt."""
    c = np.asarray(coords, dtype=np.float64)
    s = np.zeros((len(c), len(c)), dtype=np.float64)
    for k in range(c.shape[1]):
        d = c[:, None, k] - c[None, :, k]
        s += d * d
    return np.sqrt(s)


def generate_edges(coords, bp, inter_radius=4.0, intra_radius=2.0, prune=True):
    """Returns (kept_node_index, (rows, cols), edge_attrs) like the reference returns
    (struct, edge_indices, edge_attrs); kept_node_index are the original row numbers that survive
    `prune` (all of them otherwise) and the edge indices refer to the renumbered survivors
    (reset_index, preprocessing.py:101)."""
    coords = np.asarray(coords, dtype=np.float64)
    bp = np.asarray(bp).astype(np.int64)
    keep = np.arange(len(coords))
    while True:
        d = cdist_euclidean(coords)
        adj_inter = (d < inter_radius) & (d > 1e-7)                      # :110
        r_i, c_i = np.where(adj_inter)                                    # :111 row-major
        mask = np.abs(bp[r_i] - bp[c_i]) != 0                             # :113-116
        r_i, c_i = r_i[mask], c_i[mask]
        adj_intra = (d < intra_radius) & (d > 1e-7)                      # :119
        r_a, c_a = np.where(adj_intra)                                    # :121
        attr_inter = np.ones(len(r_i), dtype=np.int32)                   # :132-133 (bp differ => 1)
        attr_intra = np.where((bp[r_a] == 1) & (bp[c_a] == 1), 2, 0).astype(np.int32)   # :135
        rows = np.concatenate([r_i, r_a])                                 # :139-142
        cols = np.concatenate([c_i, c_a])
        attrs = np.concatenate([attr_inter, attr_intra])
        if not (prune and len(r_i)):                                      # :144
            return keep, (rows, cols), attrs
        # :145-151 component of the first edge's row in the undirected graph of all edges
        n = len(coords)
        label = np.arange(n)
        changed = True
        while changed:
            new = label.copy()
            np.minimum.at(new, rows, label[cols])
            np.minimum.at(new, cols, label[rows])
            new = np.minimum(new, new[new])
            changed = bool((new != label).any())
            label = new
        sel = label == label[rows[0]]
        coords, bp, keep = coords[sel], bp[sel], keep[sel]
        prune = False                                                     # :152 recursion, prune=False
