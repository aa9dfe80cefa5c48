"""nn.Module layers with MinkowskiEngine's names and parameter layout (the part reached from models/MEConv.py:221-228), backed by
libdd3d_hip.so.  Stand-alone they run op by op (conv -> instance norm -> ReLU -> pool); MEConvExp.forward fuses norm/ReLU/pool."""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from . import ops
from .sparse import SparseTensor


class MinkowskiNetwork(nn.Module):
    """ME.MinkowskiNetwork(D): an nn.Module that remembers the spatial dimension."""

    def __init__(self, D):
        super().__init__()
        self.D = D


def kaiming_normal_(tensor, a=0, mode="fan_in", nonlinearity="leaky_relu"):
    """ME.utils.kaiming_normal_ on a [K, Cin, Cout] kernel (models/MEConv.py:237-238): fan_out = Cout*K, fan_in = Cin*K."""
    K, cin, cout = tensor.shape
    fan = cin * K if mode == "fan_in" else cout * K
    std = nn.init.calculate_gain(nonlinearity, a) / math.sqrt(fan)
    with torch.no_grad():
        return tensor.normal_(0, std)


class MinkowskiConvolution(nn.Module):
    """kernel_size=3, stride=1, dilation=1, bias=False; parameter ``kernel`` [27, Cin, Cout]."""

    def __init__(self, in_channels, out_channels, kernel_size=-1, stride=1, dilation=1, bias=False, dimension=None):
        super().__init__()
        if kernel_size != 3 or stride != 1 or dilation != 1 or bias or dimension != 3:
            raise NotImplementedError("only kernel_size=3, stride=1, dilation=1, bias=False, dimension=3 is on the dd3d hot path")
        self.in_channels, self.out_channels, self.kernel_size = in_channels, out_channels, kernel_size
        self.kernel = nn.Parameter(torch.empty(27, in_channels, out_channels))
        stdv = 1.0 / math.sqrt(in_channels * 27)
        with torch.no_grad():
            self.kernel.uniform_(-stdv, stdv)

    def forward(self, x: SparseTensor) -> SparseTensor:
        cm, ts = x.coordinate_manager, x.tensor_stride
        nbr, ld = cm.kernel_map(ts, ts, 3)
        M = cm.num_rows(ts)
        return SparseTensor(ops.SparseConvFn.apply(x.F, self.kernel, nbr, ld, M), cm, ts)

    def __repr__(self):
        return (f"MinkowskiConvolution(in={self.in_channels}, out={self.out_channels}, kernel_size=[3, 3, 3], "
                f"stride=[1, 1, 1], dilation=[1, 1, 1])")


class MinkowskiInstanceNorm(nn.Module):
    """weight = ones(1,C), bias = zeros(1,C) (2-D); the affine stays torch-native, as in MinkowskiEngine."""

    def __init__(self, num_features):
        super().__init__()
        self.num_features = num_features
        self.weight = nn.Parameter(torch.ones(1, num_features))
        self.bias = nn.Parameter(torch.zeros(1, num_features))

    def forward(self, x: SparseTensor) -> SparseTensor:
        cm, ts = x.coordinate_manager, x.tensor_stride
        out = ops.InstanceNormFn.apply(x.F, cm.seg_off(ts), cm.B)
        return SparseTensor(out * self.weight + self.bias, cm, ts)

    def __repr__(self):
        return f"MinkowskiInstanceNorm(nchannels={self.num_features})"


class MinkowskiReLU(nn.Module):
    def __init__(self, inplace=False):
        super().__init__()
        self.inplace = inplace

    def forward(self, x: SparseTensor) -> SparseTensor:
        return SparseTensor(torch.relu(x.F), x.coordinate_manager, x.tensor_stride)

    def __repr__(self):
        return "MinkowskiReLU()"


class MinkowskiMaxPooling(nn.Module):
    def __init__(self, kernel_size, stride=1, dilation=1, dimension=None):
        super().__init__()
        if kernel_size != 2 or stride != 2 or dilation != 1 or dimension != 3:
            raise NotImplementedError("only kernel_size=2, stride=2, dimension=3 is on the dd3d hot path")
        self.kernel_size, self.stride = kernel_size, stride

    def forward(self, x: SparseTensor) -> SparseTensor:
        cm, ts = x.coordinate_manager, x.tensor_stride
        kids, ld = cm.kernel_map(ts, 2 * ts, 2)
        out = ops.MaxPoolFn.apply(x.F, kids, ld, cm.parent(2 * ts), cm.num_rows(2 * ts))
        return SparseTensor(out, cm, 2 * ts)

    def __repr__(self):
        return "MinkowskiMaxPooling(kernel_size=[2, 2, 2], stride=[2, 2, 2], dilation=[1, 1, 1])"
