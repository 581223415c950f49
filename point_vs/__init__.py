"""Import-path aliases of the reference package for the EGNN hot path:
`point_vs.models.geometric.{egnn_satorras,egnn_multitask,pnn_geometric_base}`,
`point_vs.models.point_neural_network_base`, `point_vs.global_objects`, `point_vs.parse_args`
resolve to the MI355X-native implementation in `pointvs_amd`, so code written against the
reference's import paths (its tests, attribution scripts, `point_vs.py`) runs unchanged on the HIP
path. Only the modules on the path exist here; the reference's data-preparation, attribution and
analysis packages are out of scope."""
