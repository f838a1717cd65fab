"""Only the helpers the hot-path files of the reference touch (diffuser/utils/arrays.py:15-30,
progress.py): to_np / to_torch and a silent progress object."""
import numpy as np
import torch


def to_np(x):
    return x.detach().cpu().numpy() if torch.is_tensor(x) else x


def to_torch(x, dtype=None, device=None):
    dtype = dtype or torch.float32
    device = device or 'cuda:0'
    if type(x) is dict:
        return {k: to_torch(v, dtype, device) for k, v in x.items()}
    if torch.is_tensor(x):
        return x.to(device).type(dtype)
    return torch.tensor(np.asarray(x), dtype=dtype, device=device)


class Silent:
    def __init__(self, *args, **kwargs):
        pass

    def update(self, *args, **kwargs):
        pass

    def stamp(self, *args, **kwargs):
        pass


Progress = Silent
