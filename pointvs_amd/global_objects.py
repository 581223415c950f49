"""DEVICE global, as /root/reference/point_vs/global_objects.py:14-22 (HIP shows up as 'cuda')."""
import torch

DEVICE = torch.device('cuda') if torch.cuda.is_available() else torch.device('cpu')
