"""Graph unpacking, input embedding wrapper, pooling + head.

Mirrors /root/reference/point_vs/models/geometric/pnn_geometric_base.py:
  PNNGeometricBase.forward                   :24-41
  PNNGeometricBase.unpack_input_data_and_predict :43-53
  PNNGeometricBase.unpack_graph              :55-58
  PygLinearPass                              :61-94
"""
import os
from abc import abstractmethod

import torch
from torch import nn

from . import functional as PF
from .global_objects import DEVICE
from .graph import prepared_for, runs_layout
from .point_neural_network_base import PointNeuralNetworkBase


class PNNGeometricBase(PointNeuralNetworkBase):
    """Base class of the geometric point networks."""

    @abstractmethod
    def get_embeddings(self, feats, edges, coords, edge_attributes, batch):
        """Input features -> final node embeddings."""

    def unpack_graph(self, graph):
        edges = getattr(graph, 'edge_index', None)      # None when the graph was built on the GPU
        attrs = getattr(graph, 'edge_attr', None)
        return (graph.x.float().to(DEVICE), None if edges is None else edges.to(DEVICE),
                graph.pos.float().to(DEVICE), None if attrs is None else attrs.to(DEVICE),
                graph.batch.to(DEVICE))

    def _embed_graph(self, graph):
        feats, edges, coords, edge_attributes, batch = self.unpack_graph(graph)
        n_nodes = feats.size(0)
        n_graphs = getattr(graph, 'num_graphs', None)
        if n_graphs is None:  # the reference syncs here too (torch.max(batch), :27)
            n_graphs = int(batch.max()) + 1
        ptr = getattr(graph, 'ptr', None)
        if ptr is None:
            counts = torch.bincount(batch, minlength=n_graphs)
            ptr = torch.cat([counts.new_zeros(1), counts.cumsum(0)])
        # (the node offsets as device int32: from the batch's cached layout tables where it has them, else converted
        # ONCE per batch object - two conversions and a gather per step were three launches of ~4 us each)
        layout = None if edges is None else runs_layout(graph, feats.device)
        if layout is not None:
            graph_ptr = layout[0]
        else:
            cached = graph.__dict__.get('_pvs_ptr32') if hasattr(graph, '__dict__') else None
            if cached is not None and cached[0] is ptr and cached[1] == ptr._version and cached[2].device == feats.device:
                graph_ptr = cached[2]
            else:
                graph_ptr = ptr.to(device=feats.device, dtype=torch.int32).contiguous()
                if hasattr(graph, '__dict__'):
                    graph.__dict__['_pvs_ptr32'] = (ptr, ptr._version, graph_ptr)
        pg = getattr(graph, 'prepared', None)     # built on the GPU (radius_graph.attach_radius_graph)
        dropping = self.training and float(getattr(self, 'dropout_p', 0.0) or 0.0) > 0.0
        if dropping:      # edge dropout ahead of the layer stack (egnn_satorras.py:320-323): a new edge list
            if pg is not None:    # graph built on the GPU: its sorted list back as the COO the dropout draws from
                e = pg.n_edges
                edges = torch.stack([pg.t['row'][:e].long(), pg.t['col'][:e].long()])
                edge_attributes = (None if pg.n_edge_attr == 0 else
                                   torch.nn.functional.one_hot(pg.t['etype'][:e].long(), pg.n_edge_attr))
            edges, edge_attributes = self.edge_dropout(edges, edge_attributes)
            pg = prepared_for(edges, edge_attributes, n_nodes)
        elif pg is None:
            pg = prepared_for(edges, edge_attributes, n_nodes, layout=layout)     # (sets graph_eptr itself from a layout)
        pg.poll_status()
        pg.set_graph_ptr(graph_ptr)
        feats, _, _ = self.embed_prepared(pg, feats, coords, need_coords=False)
        return feats, pg, graph_ptr, n_graphs

    @staticmethod
    def _pool(feats, graph_ptr, n_graphs):
        """global_mean_pool, or plain mean over nodes for a single graph (:29-33)."""
        if n_graphs == 1:
            whole = torch.tensor([0, feats.size(0)], dtype=torch.int32, device=feats.device)
            return PF.mean_pool(feats, whole).reshape(-1)
        return PF.mean_pool(feats, graph_ptr)

    @staticmethod
    def _run_head(head, pooled):
        out = pooled
        for mod in head:
            out = PF.linear(out, mod.weight, mod.bias) if isinstance(mod, nn.Linear) else mod(out)
        return out

    @classmethod
    def _pool_and_head(cls, head, feats, graph_ptr, n_graphs):
        """head(pool(feats)) (:29-36). A head that starts with a Linear takes the pooling and that Linear as one
        op (PF.pool_head: one launch forward, one backward)."""
        mods = list(head)
        if (not mods or not isinstance(mods[0], nn.Linear) or feats.size(1) > PF.POOL_HEAD_MAX_WIDTH
                or os.environ.get('PVS_FUSED_HEAD') == '0'):       # (=0: the two ops apart, for A/B)
            return cls._run_head(head, cls._pool(feats, graph_ptr, n_graphs))
        if n_graphs == 1:
            whole = torch.tensor([0, feats.size(0)], dtype=torch.int32, device=feats.device)
            out = PF.pool_head(feats, whole, mods[0].weight, mods[0].bias).reshape(-1)
        else:
            out = PF.pool_head(feats, graph_ptr, mods[0].weight, mods[0].bias)
        return cls._run_head(mods[1:], out)

    def forward(self, x):
        feats, _, graph_ptr, n_graphs = self._embed_graph(x)
        if self.feats_linear_layers is not None:
            feats = self._pool_and_head(self.feats_linear_layers, feats, graph_ptr, n_graphs)
        return feats

    def unpack_input_data_and_predict(self, input_data):
        y_true = input_data.y
        try:
            y_true = y_true.float()
        except (AttributeError, TypeError):
            pass
        y_pred = self(input_data).reshape(-1, )
        return y_pred, y_true, input_data.lig_fname, input_data.rec_fname


class PygLinearPass(nn.Module):
    """Linear input embedding with the layer-like call signature (:61-94)."""

    def __init__(self, module, feats_appended_to_coords=False, return_coords_and_edges=False):
        super().__init__()
        if feats_appended_to_coords:
            raise NotImplementedError('feats_appended_to_coords is only used by the lucid EGNN')
        self.m = module
        self.feats_appended_to_coords = feats_appended_to_coords
        self.return_coords_and_edges = return_coords_and_edges
        self._coords_src = None

    @property
    def intermediate_coords(self):
        return None if self._coords_src is None else self._coords_src().detach().cpu().numpy()

    @intermediate_coords.setter
    def intermediate_coords(self, value):
        self._coords_src = None if value is None else (lambda: torch.as_tensor(value))

    def embed(self, h, coord=None):
        if coord is not None:
            self._coords_src = lambda: coord
        return PF.linear(h, self.m.weight, self.m.bias)

    def forward(self, h, **kwargs):
        res = self.embed(h, kwargs.get('coord'))
        if self.return_coords_and_edges:
            return res, kwargs['coord'], kwargs['edge_attr'], kwargs.get('edge_messages', None)
        return res
