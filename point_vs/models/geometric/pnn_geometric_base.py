"""Alias of /root/reference/point_vs/models/geometric/pnn_geometric_base.py:11-94."""
from pointvs_amd.pnn_geometric_base import PNNGeometricBase, PygLinearPass  # noqa: F401
