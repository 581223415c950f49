"""The sampling / batching side of /root/reference/point_vs/preprocessing/data_loaders.py
(:170-186 class-balancing sampler, :512-520 loader) as the hot path needs it; parquet parsing is out of
scope."""
from pointvs_amd.data_loaders import GraphLoader, RankWeightedSampler, class_balance_weights  # noqa: F401
