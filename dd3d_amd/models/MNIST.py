"""The 2-D evaluation classifier of the reference (``models/MNIST.py:5-79``, instantiated at ``mnist_distill.py:326`` as
``ConvNet(1, 10, 128, 3, 'relu', 'instancenorm', 'avgpooling')``): ``net_depth`` blocks of conv3x3 -> norm -> activation -> pooling
and one linear layer.  Same constructor signature, module order (so a seeded construction draws the same initial weights) and
``state_dict`` keys (``features.{0,1,4,5,8,9}.*``, ``classifier.*``) as the reference; dense torch layers on ROCm (MIOpen), no HIP
kernel of ours is involved.  SURVEY.md section 8f row 3."""
from __future__ import annotations

import torch.nn as nn


class Swish(nn.Module):  # the reference names it (models/MNIST.py:32) without defining it
    def forward(self, x):
        return x * x.sigmoid()


_ACT = {"sigmoid": nn.Sigmoid, "relu": lambda: nn.ReLU(inplace=True), "leakyrelu": lambda: nn.LeakyReLU(negative_slope=0.01), "swish": Swish}
_POOL = {"maxpooling": lambda: nn.MaxPool2d(kernel_size=2, stride=2), "avgpooling": lambda: nn.AvgPool2d(kernel_size=2, stride=2)}


def _norm(kind, chw):
    c = chw[0]
    if kind == "batchnorm":
        return nn.BatchNorm2d(c, affine=True)
    if kind == "layernorm":
        return nn.LayerNorm(list(chw), elementwise_affine=True)
    if kind == "instancenorm":
        return nn.GroupNorm(c, c, affine=True)
    if kind == "groupnorm":
        return nn.GroupNorm(4, c, affine=True)
    exit("unknown net_norm: %s" % kind)  # models/MNIST.py:59


class ConvNet(nn.Module):
    def __init__(self, channel, num_classes, net_width, net_depth, net_act, net_norm, net_pooling, im_size=(32, 32)):
        super().__init__()
        if net_act not in _ACT:
            exit("unknown activation function: %s" % net_act)  # :34
        if net_pooling not in _POOL and net_pooling != "none":
            exit("unknown net_pooling: %s" % net_pooling)  # :44
        # 28x28 inputs are padded to 32x32 by the first convolution (padding 3 when channel == 1), :64-68
        chw = [channel, 32 if im_size[0] == 28 else im_size[0], 32 if im_size[0] == 28 else im_size[1]]
        blocks, cin = [], channel
        for depth in range(net_depth):
            blocks.append(nn.Conv2d(cin, net_width, kernel_size=3, padding=3 if (channel == 1 and depth == 0) else 1))
            chw[0] = cin = net_width
            if net_norm != "none":
                blocks.append(_norm(net_norm, chw))
            blocks.append(_ACT[net_act]())
            if net_pooling != "none":
                blocks.append(_POOL[net_pooling]())
                chw[1] //= 2
                chw[2] //= 2
        self.features = nn.Sequential(*blocks)
        self.classifier = nn.Linear(chw[0] * chw[1] * chw[2], num_classes)

    def embed(self, x):
        return self.features(x).flatten(1)

    def forward(self, x):
        return self.classifier(self.embed(x))
