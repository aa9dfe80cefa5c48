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
        if not self.fused or ops.true_second_order_enabled():
            # op-by-op path: every torch-native piece of the reference stays torch-native (dense tensor included); the only path of the
            # true-second-order research mode (ops.true_second_order)
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


def _avg_block(cin, cout, D):
    """conv k3 -> instance norm -> ReLU -> average pooling k2 s2: the repeating unit of both unused stacks."""
    return [ME.MinkowskiConvolution(cin, cout, kernel_size=3, dimension=D), ME.MinkowskiInstanceNorm(cout), ME.MinkowskiReLU(inplace=True),
            ME.MinkowskiAvgPooling(kernel_size=2, stride=2, dimension=D)]


class _UnusedVariant(ME.MinkowskiNetwork):
    """Shared body of the reference's two UNUSED models (models/MEConv.py:10-164; nothing in mnist_distill.py builds them): stacks of
    the block above, closed by a 1x1 convolution and a dense Linear head or, with ``full_minkowski=True``, by
    MinkowskiGlobalAvgPooling + MinkowskiLinear(128, 10).  Constructors, module trees, parameter names and initialisation are the
    reference's -- including what does not work there: ``MEConv`` never defines ``dense_shape`` (its dense head raises
    AttributeError, :76) and its full-Minkowski head feeds 16 channels into Linear(128, 10) (:21,81-82); ``MEConvImage``'s dense head
    asks ``dense()`` for one channel of a 128-channel tensor (:152-156).  The one configuration that runs in the reference,
    MEConvImage(full_minkowski=True), runs here on the HIP kernels (convolutions, instance norm, average pooling, global average
    pooling).  The two layer builders are tables of (in, out) widths; both classes carry both builders, as in the reference."""

    def _stack(self, widths, tail):
        layers = [m for cin, cout in widths for m in _avg_block(cin, cout, self.D)]
        if not self.full_minkowski:   # the dense heads sit behind a closing kernel-size-1 convolution
            layers.append(ME.MinkowskiConvolution(tail[0], tail[1], kernel_size=1, dimension=self.D))
        return nn.Sequential(*layers)

    def _make_layers(self, in_channel, out_channel, embedding_channel):
        # (:22-41, :99-118) 128 channels, then three halvings; the arguments beyond the first do not reach the widths
        return self._stack([(in_channel, 128), (128, 64), (64, 32), (32, 16)], (16, 1))

    def weight_initialization(self):
        for m in self.modules():
            if isinstance(m, (ME.MinkowskiLinear, ME.MinkowskiConvolution)):   # (a MinkowskiLinear has no .kernel: it would raise, as in the reference)
                ME.kaiming_normal_(m.kernel, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, ME.MinkowskiBatchNorm):
                nn.init.constant_(m.bn.weight, 1)
                nn.init.constant_(m.bn.bias, 0)

    def _heads(self, full_minkowski, dense_in):
        if full_minkowski:
            self.global_avg_pool = ME.MinkowskiGlobalAvgPooling()
            self.classifier = ME.MinkowskiLinear(128, 10)
        else:
            self.classifier = nn.Linear(dense_in, 10)

    def _pooled_head(self, out):
        return self.classifier(self.global_avg_pool(out)).F


class MEConv(_UnusedVariant):
    """models/MEConv.py:10-84 (no weight_initialization in its constructor; one input channel whatever ``in_channel`` says, :15)."""

    def __init__(self, in_channel, out_channel, embedding_channel=1024, dimension=3, full_minkowski=False):
        super().__init__(dimension)
        self.full_minkowski = full_minkowski
        self.features = self._make_layers(1, out_channel, embedding_channel)
        self._heads(full_minkowski, 4500)

    def _make_equal_layers(self, in_channel, out_channel, embedding_channel):
        # (:43-54) three blocks of ``out_channel`` channels, closed by 2 * out_channel -> 1
        return self._stack([(in_channel, out_channel), (128, out_channel), (128, out_channel)], (out_channel * 2, 1))

    def forward(self, x: TensorField):
        out = self.features(x.sparse())
        if self.full_minkowski:
            return self._pooled_head(out)   # 16 channels into Linear(128, 10): shape error, as in the reference (:21,82)
        min_coord = out.C.min(0, keepdim=True)[0][:, 1:].cpu()
        dense = out.dense(shape=self.dense_shape, min_coordinate=min_coord)[0]   # AttributeError, as in the reference (:18,76)
        return self.classifier(dense.view(dense.shape[0], -1))


class MEConvImage(_UnusedVariant):
    """models/MEConv.py:86-164."""

    def __init__(self, in_channel, out_channel, embedding_channel=1024, dimension=3, full_minkowski=False):
        super().__init__(dimension)
        self.full_minkowski = full_minkowski
        self.features = self._make_equal_layers(in_channel, out_channel, embedding_channel)
        self.weight_initialization()   # before the head exists (:92): the convolutions only
        self._heads(full_minkowski, 100352)

    def _make_equal_layers(self, in_channel, out_channel, embedding_channel):
        # (:120-132) three blocks of 128 channels, closed by 128 -> 128
        return self._stack([(in_channel, 128), (128, 128), (128, 128)], (128, 128))

    def forward(self, x: TensorField):
        out = x.sparse()
        B = out.coordinate_manager.B
        out = self.features(out)
        if self.full_minkowski:
            return self._pooled_head(out)
        min_coord = out.C.min(0, keepdim=True)[0][:, 1:].cpu()
        # one channel asked of a 128-channel tensor, as in the reference (:152-156)
        dense = out.dense(shape=torch.Size([B, 1, 1, 28, 28]), min_coordinate=min_coord)[0]
        return self.classifier(dense.view(-1, 100352))


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


from .._lib import true_divide as _true_divide

def create_input_batch(batch, is_minknet, device="cuda", quantization_size=0.05):
    """models/MEConv.py:260-269: divides the spatial columns IN PLACE, then wraps the batch in a TensorField on ``device``."""
    if is_minknet:
        # a correctly rounded division, as the reference's host-side ``/=`` is (torch's device kernel for tensor / scalar is not: _lib.true_divide)
        batch["coordinates"][:, 1:] = _true_divide(batch["coordinates"][:, 1:], float(quantization_size))
        return TensorField(coordinates=batch["coordinates"], features=batch["features"], device=device,
                           num_clouds=batch.get("num_clouds"))
    return batch["coordinates"].permute(0, 2, 1).to(device)
