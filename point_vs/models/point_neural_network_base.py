"""Alias of /root/reference/point_vs/models/point_neural_network_base.py:46-582."""
from pointvs_amd.point_neural_network_base import PointNeuralNetworkBase  # noqa: F401
