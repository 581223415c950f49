"""Alias of /root/reference/point_vs/models/geometric/egnn_satorras.py (EGNNLayer :23-206,
SartorrasEGNN :209-329, unsorted_segment_sum/mean :332-347) -> pointvs_amd.egnn_satorras."""
from pointvs_amd.egnn_satorras import (EGNNLayer, GraphNorm, SartorrasEGNN,  # noqa: F401
                                       unsorted_segment_mean, unsorted_segment_sum)
