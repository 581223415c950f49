"""`generate_edges` of /root/reference/point_vs/preprocessing/preprocessing.py:68-155 on the GPU
(pvs_radius_graph_*, csrc/radius_graph.hip)."""
from pointvs_amd.radius_graph import attach_radius_graph, generate_edges, radius_graph  # noqa: F401
