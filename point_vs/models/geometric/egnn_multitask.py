"""Alias of /root/reference/point_vs/models/geometric/egnn_multitask.py:11-166."""
from pointvs_amd.egnn_multitask import MultitaskSatorrasEGNN  # noqa: F401
