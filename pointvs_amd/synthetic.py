"""Deterministic synthetic protein-ligand radius graphs (SURVEY.md §8d input generator).

Edge rule = the reference's generate_edges with estimate_bonds=False
(/root/reference/point_vs/preprocessing/preprocessing.py:108-142): block 1 = every ordered
ligand<->receptor pair with 1e-7 < d < r (class 1); block 2 = every ordered pair with
1e-7 < d < r (class 2 if both receptor, else 0), each block row-major - so inter-molecular pairs
appear twice. edge_attr is the int64 one-hot(3) the data loader emits
(data_loaders.py:370), edge_index int64.
"""
import numpy as np
import torch

from .graph import Batch, Data


def synthetic_graph(seed, n_nodes=2000, n_lig=30, edge_radius=10.0, density=0.05, n_feats=12):
    rng = np.random.default_rng(seed)
    big_r = (3.0 * n_nodes / (4.0 * np.pi * density)) ** (1.0 / 3.0)
    direction = rng.normal(size=(n_nodes, 3))
    direction /= np.linalg.norm(direction, axis=1, keepdims=True)
    pts = (direction * (big_r * rng.random(n_nodes) ** (1.0 / 3.0))[:, None]).astype(np.float32)
    order = np.argsort(np.linalg.norm(pts, axis=1), kind='stable')
    pts = pts[order]                       # ligand = the n_lig points nearest the centre
    bp = np.ones(n_nodes, dtype=np.int64)
    bp[:n_lig] = 0
    feats = np.zeros((n_nodes, n_feats), dtype=np.float32)
    feats[np.arange(n_nodes), rng.integers(0, n_feats - 1, n_nodes)] = 1.0
    feats[:, n_feats - 1] = bp
    p64 = pts.astype(np.float64)
    sq = (p64 ** 2).sum(1)
    d2 = np.maximum(sq[:, None] + sq[None, :] - 2.0 * p64 @ p64.T, 0.0)
    adj = (d2 < edge_radius ** 2) & (d2 > 1e-14)
    rows, cols = np.nonzero(adj)           # row-major, like np.where in the reference
    inter = bp[rows] != bp[cols]
    e_rows = np.concatenate([rows[inter], rows])
    e_cols = np.concatenate([cols[inter], cols])
    intra_type = np.where((bp[rows] == 1) & (bp[cols] == 1), 2, 0)
    e_type = np.concatenate([np.ones(int(inter.sum()), dtype=np.int64), intra_type])
    return Data(
        x=torch.from_numpy(feats), pos=torch.from_numpy(pts),
        edge_index=torch.from_numpy(np.vstack([e_rows, e_cols]).astype(np.int64)),
        edge_attr=torch.from_numpy(np.eye(3, dtype=np.int64)[e_type]),   # int64 one-hot(3)
        y=torch.tensor(seed % 2), lig_fname=f'lig_{seed}', rec_fname=f'rec_{seed}',
        edge_layout='generate_edges')     # inter block then intra block, each row-major: see graph.runs_layout


def synthetic_batch(cfg_id, batch_size, first_graph=0, **graph_kwargs):
    """Batch of graphs g = first_graph .. first_graph+batch_size-1 with seed 1000*cfg_id + g."""
    graphs = [synthetic_graph(1000 * cfg_id + g, **graph_kwargs)
              for g in range(first_graph, first_graph + batch_size)]
    return Batch.from_data_list(graphs)


def screening_set(seed=5000, n_nodes=2000, n_lig=30, density=0.05, n_feats=12):
    """BASELINE config 5: one receptor cloud (n_nodes - n_lig points) and one
    ligand (the n_lig points nearest the centre) whose rigid poses are screened. Returns
    (lig_pos [n_lig,3], rec_pos [n_rec,3], feats [n_nodes,F] with the ligand rows first)."""
    g = synthetic_graph(seed, n_nodes=n_nodes, n_lig=n_lig, edge_radius=1.0, density=density, n_feats=n_feats)
    return g.pos[:n_lig].clone(), g.pos[n_lig:].clone(), g.x.clone()


def random_poses(lig_pos, n_poses, seed, max_shift=6.0, device=None):
    """n_poses random rigid transforms of the ligand: uniform rotation about its centroid, centroid
    moved to a uniform point of the centre ball of radius max_shift. Returns [n_poses, n_lig, 3]."""
    gen = torch.Generator(device='cpu').manual_seed(seed)
    q = torch.randn(n_poses, 4, generator=gen)
    q = q / q.norm(dim=1, keepdim=True)
    w, x, y, z = q.unbind(1)
    rot = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
                       2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                       2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], dim=1).reshape(-1, 3, 3)
    direction = torch.randn(n_poses, 3, generator=gen)
    direction = direction / direction.norm(dim=1, keepdim=True)
    shift = direction * (max_shift * torch.rand(n_poses, 1, generator=gen) ** (1.0 / 3.0))
    centred = lig_pos - lig_pos.mean(0, keepdim=True)
    poses = torch.einsum('pij,nj->pni', rot, centred) + shift[:, None, :]
    return poses.to(device) if device is not None else poses


# BASELINE.json configs
CONFIGS = {
    'cfg2': dict(cfg_id=2, graph=dict(n_nodes=2000, n_lig=30, edge_radius=10.0),
                 model=dict(dim_input=12, k=32, dim_output=1, num_layers=3, residual=False,
                            edge_residual=False, edge_attention=False, normalize=False, tanh=False,
                            dropout=0.0, graphnorm=False, update_coords=True,
                            permutation_invariance=False, node_attention=False,
                            gated_residual=False, rezero=False, softmax_attention=False,
                            model_task='classification')),
    'cfg3': dict(cfg_id=3, graph=dict(n_nodes=2000, n_lig=30, edge_radius=6.0),
                 model=dict(dim_input=12, k=64, dim_output=1, num_layers=12, residual=False,
                            edge_residual=False, edge_attention=True, normalize=False, tanh=False,
                            dropout=0.0, graphnorm=False, update_coords=True,
                            permutation_invariance=False, node_attention=True,
                            gated_residual=False, rezero=False, softmax_attention=False,
                            model_task='classification')),
    # NOT a BASELINE configuration: the reference's own default CLI shape (parse_args.py:56,67,152: 32 channels, 6
    # layers, edge_radius 4 A; BASELINE config 1's real graphs have ~500 atoms and ~11 edges per atom) - the launch-bound
    # regime (bench.py --config real4A; ~100 launches per step of a few microseconds each). The generator's density of
    # 0.05 atoms / A^3 gives E / N = 11.2 at r = 4 A.
    'real4A': dict(cfg_id=6, graph=dict(n_nodes=500, n_lig=30, edge_radius=4.0),
                   model=dict(dim_input=12, k=32, dim_output=1, num_layers=6, residual=False,
                              edge_residual=False, edge_attention=False, normalize=False, tanh=False,
                              dropout=0.0, graphnorm=False, update_coords=True,
                              permutation_invariance=False, node_attention=False,
                              gated_residual=False, rezero=False, softmax_attention=False,
                              model_task='classification')),
}
