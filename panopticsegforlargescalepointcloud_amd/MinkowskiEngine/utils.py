"""ME.utils subset: kaiming_normal_ on ME-layout kernels (used by BaseMinkowski.weight_initialization,
torch_points3d/applications/minkowski.py:104-111)."""
import math

import torch


def _fans(tensor):
    if tensor.dim() < 2:
        raise ValueError("fan in/out needs at least 2 dimensions")
    if tensor.dim() == 2:  # [Cin, Cout] kernel of a 1x1x1 convolution, treated like a Linear weight by ME
        return tensor.size(1), tensor.size(0)
    rf = tensor.size(0)  # kernel volume first: [K, Cin, Cout]
    return tensor.size(1) * rf, tensor.size(2) * rf


def kaiming_normal_(tensor, a=0, mode="fan_in", nonlinearity="leaky_relu"):
    fan_in, fan_out = _fans(tensor)
    fan = fan_in if mode == "fan_in" else fan_out
    gain = torch.nn.init.calculate_gain(nonlinearity, a)
    std = gain / math.sqrt(fan)
    with torch.no_grad():
        return tensor.normal_(0, std)
