"""Tensor / layer helpers (reference rltime/models/torch/utils.py)."""
import numpy as np
import torch
import torch.nn as nn


def init_weight(weight, mode="xavier"):
    """models/torch/utils.py:6-19: uniform(-1/sqrt(fan_in), +1/sqrt(fan_in))."""
    if mode == "default":
        return
    assert mode == "xavier", "Invalid init_weight mode: '%s'" % mode
    fan_in = int(np.prod(weight.shape[1:]))
    bound = (1.0 / fan_in) ** 0.5
    nn.init.uniform_(weight, -bound, bound)


def init_layer(layer):
    layer.bias.data.zero_()
    init_weight(layer.weight)


def conv2d(*a, **kw):
    layer = nn.Conv2d(*a, **kw)
    init_layer(layer)
    return layer


def linear(*a, **kw):
    layer = nn.Linear(*a, **kw)
    init_layer(layer)
    return layer


def conv_out_size(sz, kernel, stride, padding=0, dilation=1):
    """models/torch/utils.py:37-49 (valid padding by default)."""
    return int((sz + 2 * padding - dilation * (kernel - 1) - 1) / stride + 1)


def set_lr(optimizer, lr):
    for g in optimizer.param_groups:
        if isinstance(g["lr"], torch.Tensor):
            g["lr"].fill_(lr)          # a capturable optimizer keeps its learning rate on the device (graphed learner step)
        else:
            g["lr"] = lr


def make_tensor(x, device, non_blocking=False):
    """models/torch/utils.py:95-123: recursive; uint8 stays uint8 (image data),
    float32 stays, every other numpy dtype becomes float32; tensors only move."""
    if isinstance(x, (list, tuple)):
        return type(x)(make_tensor(v, device, non_blocking) for v in x)
    if isinstance(x, dict):
        return {k: make_tensor(v, device, non_blocking) for k, v in x.items()}
    if x is None:
        return None
    if not isinstance(x, torch.Tensor):
        x = np.asarray(x)
        if x.dtype != np.uint8 and x.dtype != np.float32:
            x = x.astype("float32")
        x = torch.from_numpy(x)
    return x.to(device, non_blocking=non_blocking)
