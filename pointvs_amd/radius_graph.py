"""Radius graphs on the GPU: `generate_edges` of the reference
(/root/reference/point_vs/preprocessing/preprocessing.py:68-155) as a HIP operator that goes from
coordinates straight to the library's CSR/CSC, plus the reference's return
format for callers that want the edge list itself.

`radius_graph`   coordinates of a batch -> PreparedGraph (what the layers consume; no COO, no one-hot)
`generate_edges` same name / arguments / result as the reference function, on device tensors
"""
import ctypes as C

import torch

from . import _lib
from .graph import PreparedGraph


def _stream(dev):
    return _lib.stream(dev)


def radius_graph(pos, bp, graph_ptr=None, inter_radius=4.0, intra_radius=None, max_graph_nodes=None,
                 need_backward=None, ligand_pairs_only=False):
    """pos [N,3] fp32, bp [N] (0 ligand / 1 receptor), graph_ptr [B+1] node offsets of the batch's
    graphs (None: one graph). Returns a PreparedGraph identical, array for array, to
    prepare_graph(edge_index, one_hot(edge_attr, 3)) of the reference's generate_edges output for
    each graph (prune=False), graphs concatenated PyG style. intra_radius=None means "no
    estimate_bonds": intra_radius = inter_radius (data_loaders.py:359-360). max_graph_nodes: size
    of the largest graph if the caller knows it (Batch.graph_node_counts), else read from graph_ptr.
    need_backward=False (default: `torch.is_grad_enabled()`) skips the by-column lists that only the
    backward reads (one radix sort less). ligand_pairs_only keeps only the pairs that touch a ligand
    atom (the pose-dependent part of a screening graph, pointvs_amd/screening.py)."""
    if need_backward is None:
        need_backward = torch.is_grad_enabled()
    _lib.require_hip(pos, bp)
    lib = _lib.lib()
    if pos.dtype != torch.float32:
        raise TypeError(f'pos must be float32 (got {pos.dtype})')
    pos = pos.contiguous()
    dev = pos.device
    n = int(pos.shape[0])
    bp8 = bp.to(device=dev, dtype=torch.uint8).contiguous()
    if graph_ptr is None:
        graph_ptr = torch.tensor([0, n], dtype=torch.int32)
        max_graph_nodes = n
    if max_graph_nodes is None:
        max_graph_nodes = int((graph_ptr[1:] - graph_ptr[:-1]).max().item())
    gp = graph_ptr.to(device=dev, dtype=torch.int32).contiguous()
    n_graphs = int(gp.numel()) - 1
    if intra_radius is None:
        intra_radius = inter_radius
    i32 = dict(dtype=torch.int32, device=dev)
    rowptr, inter_ptr, intra_ptr = (torch.empty(n + 1, **i32) for _ in range(3))
    st_bytes = lib.pvs_radius_graph_state_bytes(n, n_graphs, max_graph_nodes)
    state = torch.empty(st_bytes, dtype=torch.uint8, device=dev)
    _lib.check(lib.pvs_radius_graph_count(
        _lib.ptr(pos), _lib.ptr(bp8), _lib.ptr(gp), n_graphs, n, max_graph_nodes, float(inter_radius),
        float(intra_radius), 1 if ligand_pairs_only else 0, _lib.ptr(rowptr), _lib.ptr(inter_ptr),
        _lib.ptr(intra_ptr), _lib.ptr(state),
        st_bytes, _stream(dev)), 'pvs_radius_graph_count')
    n_edges = int(rowptr[n].item())        # E is data dependent: the one host sync of the builder
    e_alloc = max(n_edges, 1)
    t = {
        'rowptr': rowptr, 'row': torch.empty(e_alloc, **i32), 'col': torch.empty(e_alloc, **i32),
        'etype': torch.empty(e_alloc, dtype=torch.uint8, device=dev), 'perm': torch.empty(e_alloc, **i32),
        'inv_deg': torch.empty(n, dtype=torch.float32, device=dev),
        'status': torch.zeros(1, **i32), 'inter_ptr': inter_ptr, 'intra_ptr': intra_ptr,
    }
    if need_backward:
        t['colptr'], t['cedge'] = torch.empty(n + 1, **i32), torch.empty(e_alloc, **i32)
    ws_bytes = lib.pvs_radius_graph_workspace_bytes(n, n_graphs, n_edges)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    _lib.check(lib.pvs_radius_graph_fill(
        _lib.ptr(bp8), _lib.ptr(gp), n_graphs, n, max_graph_nodes, n_edges, _lib.ptr(rowptr),
        _lib.ptr(inter_ptr), _lib.ptr(intra_ptr), _lib.ptr(t['row']), _lib.ptr(t['col']), _lib.ptr(t['etype']),
        _lib.ptr(t['perm']), _lib.ptr(t.get('colptr')), _lib.ptr(t.get('cedge')), _lib.ptr(t['inv_deg']),
        _lib.ptr(state), st_bytes, _lib.ptr(ws), ws_bytes, _stream(dev)), 'pvs_radius_graph_fill')
    pg = PreparedGraph(n, n_edges, 3, t)
    pg._status_checked = True     # built, not parsed: nothing to validate
    return pg


def edges_in_reference_order(pg):
    """(edge_index [2,E] int64, edge_attrs [E] int64) of a radius_graph result, in the order
    generate_edges returns them (inter block then intra block, each row-major)."""
    e = pg.n_edges
    perm = pg.t['perm'][:e].long()
    edge_index = torch.empty((2, e), dtype=torch.int64, device=perm.device)
    edge_index[0, perm] = pg.t['row'][:e].long()
    edge_index[1, perm] = pg.t['col'][:e].long()
    attrs = torch.empty(e, dtype=torch.int64, device=perm.device)
    attrs[perm] = pg.t['etype'][:e].long()
    return edge_index, attrs


def _component_of(pg, start):
    """Boolean mask of the nodes connected to node `start` (min-label sweeps on the device)."""
    lib = _lib.lib()
    dev = pg.t['row'].device
    labels = torch.arange(pg.n_nodes, dtype=torch.int32, device=dev)
    changed = torch.zeros(1, dtype=torch.int32, device=dev)
    while True:
        changed.zero_()
        for _ in range(4):      # a few sweeps per host round trip
            _lib.check(lib.pvs_graph_min_label_step(_lib.ptr(pg.t['rowptr']), _lib.ptr(pg.t['col']),
                                                    pg.n_nodes, _lib.ptr(labels), _lib.ptr(changed),
                                                    _stream(dev)), 'pvs_graph_min_label_step')
        if int(changed.item()) == 0:
            break
    return labels == labels[start]


def generate_edges(pos, bp, inter_radius=4.0, intra_radius=2.0, prune=True):
    """The reference's generate_edges (preprocessing.py:68-155) for one structure on the GPU.
    Returns (keep, edge_index, edge_attrs): `keep` = indices of the atoms that survive `prune` (the
    reference returns the pruned DataFrame), edge_index [2,E] int64 over the renumbered survivors,
    edge_attrs [E] int64 in {0,1,2}; same order as the reference."""
    pg = radius_graph(pos, bp, None, inter_radius, intra_radius)
    keep = torch.arange(pos.shape[0], device=pos.device)
    n_inter = int(pg.t['inter_ptr'][pg.n_nodes].item())
    if prune and n_inter:
        # first edge of the reference's list = first inter edge: its row is the lowest node with one
        has_inter = (pg.t['inter_ptr'][1:] - pg.t['inter_ptr'][:-1]) > 0
        start = int(torch.nonzero(has_inter)[0].item())
        mask = _component_of(pg, start)
        keep = torch.nonzero(mask).reshape(-1)
        pg = radius_graph(pos[keep].contiguous(), bp.to(pos.device)[keep], None, inter_radius, intra_radius)
    edge_index, attrs = edges_in_reference_order(pg)
    return keep, edge_index, attrs


def attach_radius_graph(batch, inter_radius, intra_radius=None):
    """Builds the batch's graph on the GPU from `batch.pos`, the bp column of `batch.x` (last
    feature, preprocessing.make_bit_vector) and `batch.ptr`, and hangs it on the batch as
    `batch.prepared`: the models then skip edge_index / edge_attr altogether."""
    max_nodes = max(batch.graph_node_counts) if getattr(batch, 'graph_node_counts', None) else None
    batch.prepared = radius_graph(batch.pos, batch.x[:, -1], batch.ptr, inter_radius, intra_radius,
                                  max_graph_nodes=max_nodes)
    return batch


class PoseBatcher:
    """Virtual-screening batches (BASELINE config 5): B poses of one ligand against one receptor as
    a PyG-style batch whose graph is built on the GPU. The receptor rows, the features, `ptr` and
    `batch` are laid out once; per batch only the ligand coordinates are written."""

    def __init__(self, rec_pos, feats, n_lig, batch_size, edge_radius, intra_radius=None):
        from .graph import Batch
        dev = rec_pos.device
        n = n_lig + rec_pos.shape[0]
        self.n_lig, self.n, self.b = n_lig, n, batch_size
        self.edge_radius, self.intra_radius = edge_radius, intra_radius
        pos = torch.empty((batch_size, n, 3), dtype=torch.float32, device=dev)
        pos[:, n_lig:] = rec_pos
        self._pos = pos
        ptr = torch.arange(batch_size + 1, dtype=torch.int64) * n
        self.batch = Batch(
            x=feats.to(dev).repeat(batch_size, 1), pos=pos.view(-1, 3), edge_index=None, edge_attr=None,
            batch=torch.arange(batch_size, device=dev).repeat_interleave(n), ptr=ptr,
            y=torch.zeros(batch_size, device=dev), lig_fname=['pose'] * batch_size,
            rec_fname=['receptor'] * batch_size, num_graphs=batch_size,
            graph_node_counts=[n] * batch_size)

    def load(self, lig_poses):
        """lig_poses [B, n_lig, 3] (device) -> the batch with its graph attached."""
        self._pos[:, :self.n_lig] = lig_poses
        return attach_radius_graph(self.batch, self.edge_radius, self.intra_radius)
