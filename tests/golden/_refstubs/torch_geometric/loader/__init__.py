import torch
from torch_geometric.data import Batch


class DataLoader(torch.utils.data.DataLoader):
    def __init__(self, dataset, batch_size=1, shuffle=False, **kwargs):
        kwargs.pop('collate_fn', None)
        super().__init__(dataset, batch_size, shuffle,
                         collate_fn=Batch.from_data_list, **kwargs)
