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
    """ME.utils.kaiming_normal_ on a [K, Cin, Cout] kernel (models/MEConv.py:237-238): fan_out = Cout*K, fan_in = Cin*K
    (a kernel_size=1 kernel is [Cin, Cout]: K = 1)."""
    K, cin, cout = tensor.shape if tensor.dim() == 3 else (1, *tensor.shape)
    fan = cin * K if mode == "fan_in" else cout * K
    std = nn.init.calculate_gain(nonlinearity, a) / math.sqrt(fan)
    with torch.no_grad():
        return tensor.normal_(0, std)


class MinkowskiConvolution(nn.Module):
    """kernel_size=3, stride=1, dilation=1, bias=False; parameter ``kernel`` [27, Cin, Cout]."""

    def __init__(self, in_channels, out_channels, kernel_size=-1, stride=1, dilation=1, bias=False, dimension=None):
        super().__init__()
        if kernel_size not in (1, 3) or stride != 1 or dilation != 1 or bias or dimension != 3:
            raise NotImplementedError("only kernel_size in (1, 3), stride=1, dilation=1, bias=False, dimension=3 (what models/MEConv.py builds)")
        self.in_channels, self.out_channels, self.kernel_size = in_channels, out_channels, kernel_size
        vol = kernel_size ** 3
        # kernel_size=1 keeps MinkowskiEngine's 2-D parameter [Cin, Cout] (kernel volume 1); kernel_size=3 the [27, Cin, Cout] one
        self.kernel = nn.Parameter(torch.empty(in_channels, out_channels) if vol == 1 else torch.empty(vol, in_channels, out_channels))
        stdv = 1.0 / math.sqrt(in_channels * vol)
        with torch.no_grad():
            self.kernel.uniform_(-stdv, stdv)

    def forward(self, x: SparseTensor) -> SparseTensor:
        cm, ts = x.coordinate_manager, x.tensor_stride
        M = cm.num_rows(ts)
        if self.kernel_size == 1:   # (only the reference's unused variants: models/MEConv.py:39,52,116,130)
            return SparseTensor(ops.sparse_conv_any(x.F, self.kernel[None], None, 0, M), cm, ts)
        nbr, ld = cm.kernel_map(ts, ts, 3)
        return SparseTensor(ops.sparse_conv_any(x.F, self.kernel, nbr, ld, M), cm, ts)   # every shape on hand-written kernels

    def __repr__(self):
        k = self.kernel_size
        return (f"MinkowskiConvolution(in={self.in_channels}, out={self.out_channels}, kernel_size=[{k}, {k}, {k}], "
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
        if self.num_features == 128 and not ops.true_second_order_enabled():
            out = ops.InstanceNormFn.apply(x.F, cm.seg_off(ts), cm.B)
        else:   # channel counts of the reference's unused MEConv stack (64, 32, 16), and the true-second-order mode: torch ops on the device
            out = ops.instance_norm_torch(x.F, x.C[:, 0].to(torch.int64), cm.B)
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
        if ops.true_second_order_enabled():
            out = ops.max_pool_true(x.F, kids, ld, cm.num_rows(2 * ts))
        else:
            out = ops.MaxPoolFn.apply(x.F, kids, ld, cm.parent(2 * ts), cm.num_rows(2 * ts))
        return SparseTensor(out, cm, 2 * ts)

    def __repr__(self):
        return "MinkowskiMaxPooling(kernel_size=[2, 2, 2], stride=[2, 2, 2], dilation=[1, 1, 1])"


# ---- layers that only the reference's UNUSED models name (models/MEConv.py:10-164: MEConv, MEConvImage) --------------------------
class MinkowskiAvgPooling(nn.Module):
    """kernel_size=2, stride=2: mean over the existing children of every output voxel."""

    def __init__(self, kernel_size, stride=1, dilation=1, dimension=None):
        super().__init__()
        if kernel_size != 2 or stride != 2 or dilation != 1 or dimension != 3:
            raise NotImplementedError("only kernel_size=2, stride=2, dimension=3 (what models/MEConv.py builds)")
        self.kernel_size, self.stride = kernel_size, stride

    def forward(self, x: SparseTensor) -> SparseTensor:
        cm, ts = x.coordinate_manager, x.tensor_stride
        kids, ld = cm.kernel_map(ts, 2 * ts, 2)
        if ops.true_second_order_enabled():
            out = ops.avg_pool_torch(x.F, kids, cm.num_rows(2 * ts))
        else:
            out = ops.AvgPoolFn.apply(x.F, kids, ld, cm.parent(2 * ts), cm.num_rows(2 * ts))
        return SparseTensor(out, cm, 2 * ts)

    def __repr__(self):
        return "MinkowskiAvgPooling(kernel_size=[2, 2, 2], stride=[2, 2, 2], dilation=[1, 1, 1])"


class PooledBatch:
    """What MinkowskiGlobalAvgPooling returns here: one feature row per cloud (``.F`` [B, C]), in batch order."""

    def __init__(self, F, coordinate_manager):
        self.F, self.coordinate_manager = F, coordinate_manager


class MinkowskiGlobalAvgPooling(nn.Module):
    def forward(self, x: SparseTensor) -> PooledBatch:
        cm, ts = x.coordinate_manager, x.tensor_stride
        if ops.true_second_order_enabled():   # differentiable torch ops on the device
            b = x.C[:, 0].to(torch.int64)
            cnt = torch.bincount(b, minlength=cm.B).to(x.F.dtype).clamp_(min=1)[:, None]
            return PooledBatch(torch.zeros((cm.B, x.F.shape[1]), dtype=x.F.dtype, device=x.F.device).index_add(0, b, x.F) / cnt, cm)
        return PooledBatch(ops.GlobalAvgPoolFn.apply(x.F, cm.seg_off(ts), cm.B), cm)

    def __repr__(self):
        return "MinkowskiGlobalAvgPooling()"


class MinkowskiLinear(nn.Module):
    """A torch Linear on the feature rows, kept under ``.linear`` as in MinkowskiEngine (state_dict keys ``linear.weight`` / ``linear.bias``)."""

    def __init__(self, in_features, out_features, bias=True):
        super().__init__()
        self.linear = nn.Linear(in_features, out_features, bias=bias)

    def forward(self, x):
        out = self.linear(x.F)
        if isinstance(x, SparseTensor):
            return SparseTensor(out, x.coordinate_manager, x.tensor_stride)
        return PooledBatch(out, x.coordinate_manager)

    def __repr__(self):
        return f"MinkowskiLinear(in_features={self.linear.in_features}, out_features={self.linear.out_features}, bias={self.linear.bias is not None})"


class MinkowskiBatchNorm(nn.Module):
    """Named by weight_initialization (models/MEConv.py:64,142) but built by no model of the reference."""

    def __init__(self, num_features, eps=1e-5, momentum=0.1):
        super().__init__()
        self.bn = nn.BatchNorm1d(num_features, eps=eps, momentum=momentum)

    def forward(self, x: SparseTensor) -> SparseTensor:
        return SparseTensor(self.bn(x.F), x.coordinate_manager, x.tensor_stride)
