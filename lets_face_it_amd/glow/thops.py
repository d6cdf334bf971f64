"""Tensor helpers with the call signatures of glow_pytorch/glow/thops.py (sum, mean, split_feature, cat_feature).

Only the stand-alone module API uses them; the fused engine never does.
"""
import torch


def _reduce(tensor, dim, keepdim, fn):
    if dim is None:
        return fn(tensor)
    dims = [dim] if isinstance(dim, int) else list(dim)
    return fn(tensor, dim=dims, keepdim=keepdim)


def sum(tensor, dim=None, keepdim=False):  # noqa: A001 - reference name (thops.py:4-17)
    return _reduce(tensor, dim, keepdim, torch.sum)


def mean(tensor, dim=None, keepdim=False):  # thops.py:20-33
    return _reduce(tensor, dim, keepdim, torch.mean)


def split_feature(tensor, type="split"):  # noqa: A002 - reference keyword (thops.py:36-44)
    C = tensor.size(1)
    if type == "split":
        return tensor[:, :C // 2, ...], tensor[:, C // 2:, ...]
    if type == "cross":
        return tensor[:, 0::2, ...], tensor[:, 1::2, ...]
    raise ValueError(type)


def cat_feature(tensor_a, tensor_b):  # thops.py:47-48
    return torch.cat((tensor_a, tensor_b), dim=1)
