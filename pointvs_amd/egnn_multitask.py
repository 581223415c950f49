"""Two-headed EGNN (pose / affinity) with the reference's class surface.

Mirrors /root/reference/point_vs/models/geometric/egnn_multitask.py:11-166.
"""
from torch import nn

from . import functional as PF
from .egnn_satorras import SartorrasEGNN


class MultitaskSatorrasEGNN(SartorrasEGNN):
    """Same layer stack, per-layer attention placement flags, two heads."""

    def _layer_flags(self, idx, num_layers, kw):
        def placed(on, first_only, final_only):
            if not on:
                return False
            if not first_only and not final_only:
                return True
            return bool((first_only and idx == 0) or (final_only and idx == num_layers - 1))
        return (placed(kw['edge_attention'], kw['edge_attention_first_only'],
                       kw['edge_attention_final_only']),
                placed(kw['node_attention'], kw['node_attention_first_only'],
                       kw['node_attention_final_only']))

    def build_net(self, dim_input, k, dim_output, act_fn=nn.SiLU(), num_layers=4, residual=True,
                  edge_residual=False, edge_attention=False, normalize=True, tanh=True, dropout=0.0,
                  graphnorm=True, update_coords=True, permutation_invariance=False,
                  attention_activation_fn='sigmoid', node_attention=False,
                  node_attention_final_only=False, edge_attention_final_only=False,
                  node_attention_first_only=False, edge_attention_first_only=False,
                  gated_residual=False, rezero=False, model_task='classification',
                  final_softplus=False, softmax_attention=False, **kwargs):
        assert not (gated_residual and rezero), 'gated_residual and rezero are incompatible'
        if not 0.0 <= float(dropout or 0.0) < 1.0:
            raise ValueError(f'dropout must be in [0, 1), got {dropout}')
        self.n_layers = num_layers
        self.dropout_p = dropout
        self.residual, self.edge_residual = residual, edge_residual
        self.gated_residual, self.rezero = gated_residual, rezero
        self.model_task = model_task
        self.softmax_attention = softmax_attention
        kw = dict(residual=residual, edge_residual=edge_residual, edge_attention=edge_attention,
                  normalize=normalize, tanh=tanh, graphnorm=graphnorm, update_coords=update_coords,
                  permutation_invariance=permutation_invariance,
                  attention_activation_fn=attention_activation_fn, node_attention=node_attention,
                  gated_residual=gated_residual, rezero=rezero, softmax_attention=softmax_attention,
                  node_attention_final_only=node_attention_final_only,
                  edge_attention_final_only=edge_attention_final_only,
                  node_attention_first_only=node_attention_first_only,
                  edge_attention_first_only=edge_attention_first_only)
        layers = self._build_layers(dim_input, k, num_layers, act_fn, kw)
        affinity = [nn.Linear(k, dim_output), nn.Softplus() if final_softplus else nn.ReLU()]
        self.feats_linear_layers_pose = nn.Sequential(nn.Linear(k, 1))
        self.feats_linear_layers_affinity = nn.Sequential(*affinity)
        return nn.Sequential(*layers)

    def forward(self, graph):
        feats, pg, graph_ptr, n_graphs = self._embed_graph(graph)
        if 'classification' in self.model_task:
            return self._pool_and_head(self.feats_linear_layers_pose, feats, graph_ptr, n_graphs)
        return self._pool_and_head(self.feats_linear_layers_affinity, feats, graph_ptr, n_graphs)
