import torch


class Data:
    def __init__(self, **kwargs):
        self.__dict__.update(kwargs)

    def to(self, device):
        for key, val in list(self.__dict__.items()):
            if torch.is_tensor(val):
                self.__dict__[key] = val.to(device)
        return self

    def pin_memory(self):
        return self

    def keys(self):
        return list(self.__dict__.keys())


class Batch(Data):
    @staticmethod
    def from_data_list(items):
        merged, offset, batch_vec, shifted = {}, 0, [], []
        for gid, item in enumerate(items):
            n_nodes = item.x.size(0)
            batch_vec.append(torch.full((n_nodes,), gid, dtype=torch.long))
            if torch.is_tensor(item.edge_index) and item.edge_index.dim() == 2:
                shifted.append(item.edge_index + offset)
            offset += n_nodes
        for key in items[0].__dict__:
            vals = [getattr(item, key) for item in items]
            if key == 'edge_index':
                merged[key] = torch.cat(shifted, dim=1)
            elif torch.is_tensor(vals[0]):
                merged[key] = torch.cat(
                    [v.reshape(1) if v.dim() == 0 else v for v in vals], dim=0)
            elif vals[0] is None:
                merged[key] = None
            else:
                merged[key] = vals
        merged['batch'] = torch.cat(batch_vec)
        return Batch(**merged)


class Dataset(torch.utils.data.Dataset):
    def __init__(self, *args, **kwargs):
        super().__init__()
