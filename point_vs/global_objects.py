"""Alias of /root/reference/point_vs/global_objects.py:14-25 (DEVICE)."""
from pointvs_amd.global_objects import DEVICE  # noqa: F401
