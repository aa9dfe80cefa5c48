"""Drop-in for the classes of the reference's ``models/MEConv.py``: MEConvExp (:208-258), MEPytorch (:166-207) and
create_input_batch (:260-269) -- the hot path -- and the unused MEConv / MEConvImage (:10-164), on hand-written gfx950 kernels
instead of MinkowskiEngine.

Same constructor signatures, module tree (``features.{0,1,4,5}``, ``classifier``), parameter order, state_dict keys and
initialisation as the reference.  Keyword-only extensions with reference defaults (``num_classes=10, dense_extent=(1,28,28)``)
open the hard-coded MNIST geometry (Linear(100352,10), dense shape [B,128,1,28,28]) for other cloud shapes."""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import nn as ME
from .. import ops
from ..sparse import SparseTensor, TensorField


class MEConvExp(ME.MinkowskiNetwork):
    def __init__(self, in_channel, out_channel, embedding_channel=1024, dimension=3, full_minkowski=False, *,
                 num_classes=10, dense_extent=(1, 28, 28), fused=True):
        super().__init__(dimension)
        if full_minkowski:
            raise NotImplementedError("full_minkowski=True is never used by the reference (models/MEConv.py:249)")
        self.full_minkowski = full_minkowski
        self.fused = fused
        self.dense_extent = tuple(dense_extent)
        self.features = self._make_equal_layers(in_channel, 128, embedding_channel)
        self.weight_initialization()
        self.classifier = nn.Linear(128 * self.dense_extent[0] * self.dense_extent[1] * self.dense_extent[2], num_classes)

    def _make_equal_layers(self, in_channel, out_channel, embedding_channel):
        layers = []
        for _ in range(2):
            layers += [ME.MinkowskiConvolution(in_channel, out_channel, kernel_size=3, dimension=self.D),
                       ME.MinkowskiInstanceNorm(128), ME.MinkowskiReLU(inplace=True),
                       ME.MinkowskiMaxPooling(kernel_size=2, stride=2, dimension=self.D)]
            in_channel = 128
        return nn.Sequential(*layers)

    def weight_initialization(self):
        for m in self.modules():
            if isinstance(m, ME.MinkowskiConvolution):
                ME.kaiming_normal_(m.kernel, mode="fan_out", nonlinearity="relu")

    def forward(self, x: TensorField):
        out = x.sparse()
        cm = out.coordinate_manager
        if not self.fused:
            # op-by-op path: every torch-native piece of the reference stays torch-native (dense tensor included)
            out = self.features(out)
            B = cm.B
            shape = torch.Size([B, 128, *self.dense_extent])
            min_coord = out.C.min(0, keepdim=True)[0][:, 1:]
            dense = out.dense(shape=shape, min_coordinate=min_coord)[0]
            return self.classifier(dense.view(B, -1))
        # fused path: build every coordinate map first (one host sync), then conv -> [norm+ReLU+pool] x2 -> sparse classifier
        f, ts = out.F, 1
        plan = []
        for _ in range(2):
            plan.append((cm.kernel_map(ts, ts, 3), cm.kernel_map(ts, 2 * ts, 2), cm.seg_off(ts), cm.parent(2 * ts)))
            ts *= 2
        dp = ops.DensePlacement(cm, ts, self.dense_extent)
        cm.sync()
        ts = 1
        for blk, ((nbr, ld), (kids, kld), seg_off, parent) in zip((0, 4), plan):
            conv, inorm = self.features[blk], self.features[blk + 1]
            M_in, M_out = cm.num_rows(ts), cm.num_rows(2 * ts)
            y = ops.SparseConvFn.apply(f, conv.kernel, nbr, ld, M_in)
            f = ops.NormActPoolFn.apply(y, inorm.weight, inorm.bias, seg_off, cm.B, kids, kld, parent, M_out,
                                        ops._single_group(M_in, f.device), 1)
            ts *= 2
        return ops.sparse_linear(f, self.classifier.weight, self.classifier.bias, dp)


class _UnusedVariant(ME.MinkowskiNetwork):
    """Shared body of the reference's two UNUSED models (models/MEConv.py:10-164; nothing in mnist_distill.py builds them): layer
    stacks with MinkowskiAvgPooling, and either a dense Linear head or, with ``full_minkowski=True``, MinkowskiGlobalAvgPooling +
    MinkowskiLinear(128, 10).  Constructors, module trees, parameter names and initialisation follow the reference line by line --
    including what does not work there: ``MEConv`` never defines ``dense_shape`` (its dense head raises AttributeError, :76) and its
    full-Minkowski head feeds 16 channels into Linear(128, 10) (:21,81-82); ``MEConvImage``'s dense head asks ``dense()`` for one
    channel of a 128-channel tensor (:152-156).  The one configuration that runs in the reference, MEConvImage(full_minkowski=True),
    runs here on the HIP kernels (convolutions, instance norm, average pooling, global average pooling)."""

    def _make_layers(self, in_channel, out_channel, embedding_channel):
        layers = []
        layers += [ME.MinkowskiConvolution(in_channel, 128, kernel_size=3, dimension=self.D)]
        layers += [ME.MinkowskiInstanceNorm(128)]
        layers += [ME.MinkowskiReLU(inplace=True)]
        layers += [ME.MinkowskiAvgPooling(kernel_size=2, stride=2, dimension=self.D)]
        in_channel = 128
        out_channel = 64
        for d in range(3):
            layers += [ME.MinkowskiConvolution(in_channel, out_channel, kernel_size=3, dimension=self.D)]
            layers += [ME.MinkowskiInstanceNorm(out_channel)]
            layers += [ME.MinkowskiReLU(inplace=True)]
            layers += [ME.MinkowskiAvgPooling(kernel_size=2, stride=2, dimension=self.D)]
            in_channel //= 2
            out_channel //= 2
        if not self.full_minkowski:
            layers += [ME.MinkowskiConvolution(out_channel * 2, 1, kernel_size=1, dimension=self.D)]
        return nn.Sequential(*layers)

    def weight_initialization(self):
        for m in self.modules():
            if isinstance(m, ME.MinkowskiLinear):
                ME.kaiming_normal_(m.kernel, mode="fan_out", nonlinearity="relu")   # (as in the reference: MinkowskiLinear has no .kernel)
            if isinstance(m, ME.MinkowskiConvolution):
                ME.kaiming_normal_(m.kernel, mode="fan_out", nonlinearity="relu")
            if isinstance(m, ME.MinkowskiBatchNorm):
                nn.init.constant_(m.bn.weight, 1)
                nn.init.constant_(m.bn.bias, 0)


class MEConv(_UnusedVariant):
    """models/MEConv.py:10-84."""

    def __init__(self, in_channel, out_channel, embedding_channel=1024, dimension=3, full_minkowski=False):
        super().__init__(dimension)
        self.full_minkowski = full_minkowski
        self.features = self._make_layers(1, out_channel, embedding_channel)
        if not full_minkowski:
            self.classifier = nn.Linear(4500, 10)
        if full_minkowski:
            self.global_avg_pool = ME.MinkowskiGlobalAvgPooling()
            self.classifier = ME.MinkowskiLinear(128, 10)

    def _make_equal_layers(self, in_channel, out_channel, embedding_channel):
        layers = []
        for d in range(3):
            layers += [ME.MinkowskiConvolution(in_channel, out_channel, kernel_size=3, dimension=self.D)]
            layers += [ME.MinkowskiInstanceNorm(out_channel)]
            layers += [ME.MinkowskiReLU(inplace=True)]
            layers += [ME.MinkowskiAvgPooling(kernel_size=2, stride=2, dimension=self.D)]
            in_channel = 128
        if not self.full_minkowski:
            layers += [ME.MinkowskiConvolution(out_channel * 2, 1, kernel_size=1, dimension=self.D)]
        return nn.Sequential(*layers)

    def forward(self, x: TensorField):
        out = x.sparse()
        out = self.features(out)
        if not self.full_minkowski:
            min_coord, _ = out.C.min(0, keepdim=True)
            min_coord = min_coord[:, 1:].cpu()
            out = out.dense(shape=self.dense_shape, min_coordinate=min_coord)[0]   # AttributeError, as in the reference (:18,76)
            out = out.view(out.shape[0], -1)
            out = self.classifier(out)
        if self.full_minkowski:
            out = self.global_avg_pool(out)
            out = self.classifier(out).F   # 16 channels into Linear(128, 10): shape error, as in the reference (:21,82)
        return out


class MEConvImage(_UnusedVariant):
    """models/MEConv.py:86-164."""

    def __init__(self, in_channel, out_channel, embedding_channel=1024, dimension=3, full_minkowski=False):
        super().__init__(dimension)
        self.full_minkowski = full_minkowski
        self.features = self._make_equal_layers(in_channel, out_channel, embedding_channel)
        self.weight_initialization()
        if not full_minkowski:
            self.classifier = nn.Linear(100352, 10)
        if full_minkowski:
            self.global_avg_pool = ME.MinkowskiGlobalAvgPooling()
            self.classifier = ME.MinkowskiLinear(128, 10)

    def _make_equal_layers(self, in_channel, out_channel, embedding_channel):
        layers = []
        for d in range(3):
            layers += [ME.MinkowskiConvolution(in_channel, 128, kernel_size=3, dimension=self.D)]
            layers += [ME.MinkowskiInstanceNorm(128)]
            layers += [ME.MinkowskiReLU(inplace=True)]
            layers += [ME.MinkowskiAvgPooling(kernel_size=2, stride=2, dimension=self.D)]
            in_channel = 128
        if not self.full_minkowski:
            layers += [ME.MinkowskiConvolution(in_channel, 128, kernel_size=1, dimension=self.D)]
        return nn.Sequential(*layers)

    def forward(self, x: TensorField):
        out = x.sparse()
        B = out.coordinate_manager.B
        out = self.features(out)
        if not self.full_minkowski:
            dense_shape = torch.Size([B, 1, 1, 28, 28])   # one channel asked of a 128-channel tensor, as in the reference (:152)
            min_coord, _ = out.C.min(0, keepdim=True)
            min_coord = min_coord[:, 1:].cpu()
            out = out.dense(shape=dense_shape, min_coordinate=min_coord)[0]
            out = out.view(-1, 100352)
            out = self.classifier(out)
        if self.full_minkowski:
            out = self.global_avg_pool(out)
            out = self.classifier(out).F
        return out


class MEPytorch(ME.MinkowskiNetwork):
    """``--pytorch`` model: sparse() on the HIP kernels, then the reference's dense torch layers (on ROCm these are MIOpen)."""

    def __init__(self, in_channel, out_channel, dimension=3):
        super().__init__(dimension)
        self.net_depth = 3
        self.pyt = self._make_pytorch_layers(in_channel, 128)
        self.classifier = nn.Linear(1152, 10)

    def _make_pytorch_layers(self, in_channel, out_channel):
        layers = []
        for _ in range(self.net_depth):
            layers += [nn.Conv3d(in_channel, 128, kernel_size=(1, 3, 3), padding=1), nn.ReLU(inplace=True),
                       nn.MaxPool3d(kernel_size=2, stride=2)]
            in_channel = 128
        return nn.Sequential(*layers)

    def forward(self, x: TensorField):
        out = x.sparse()
        B = out.coordinate_manager.B
        min_coord = out.C.min(0, keepdim=True)[0][:, 1:]
        dense = out.dense(shape=torch.Size([B, 3, 1, 28, 28]), min_coordinate=min_coord)[0]
        out = self.pyt(dense)
        return self.classifier(out.view(out.size(0), -1))


def create_input_batch(batch, is_minknet, device="cuda", quantization_size=0.05):
    """models/MEConv.py:260-269: divides the spatial columns IN PLACE, then wraps the batch in a TensorField on ``device``."""
    if is_minknet:
        batch["coordinates"][:, 1:] = batch["coordinates"][:, 1:] / quantization_size
        return TensorField(coordinates=batch["coordinates"], features=batch["features"], device=device,
                           num_clouds=batch.get("num_clouds"))
    return batch["coordinates"].permute(0, 2, 1).to(device)
