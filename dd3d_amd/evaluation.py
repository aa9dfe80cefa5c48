"""Evaluation block of the reference loop (``mnist_distill.py:65-198`` and ``:320-330``): rasterise the
synthetic clouds into 28x28 occupancy images, train a fresh 2-D ``ConvNet`` on them and report its test accuracy.

Same function names, arguments and behaviour as the reference -- including two of its quirks, kept so that results match:
``epoch`` ignores its ``aug`` argument and augments every batch, test batches included (:138), and ``augment`` draws from numpy's
GLOBAL generator, one ``permutation`` per image plus the chosen transform's draws, in image order (:113-122).

What runs where:
* ``populate_img``  : HIP kernel ``dd3d_populate_img`` (one thread per point) instead of 25 000 Python assignments (:163-168).
* ``augment``       : the batch is brought to the host ONCE, transformed there image by image (28x28 images; crop / nearest-neighbour
                      rescale / spline rotation -- the rotation is scipy's, as in the reference, :102) and sent back in one copy.  The
                      reference does this per image on the device with a host round trip inside every rotation.
* ``epoch`` / ``evaluate_synset`` : dense torch layers on ROCm; loss and accuracy are accumulated on the device and read back once
                      per epoch (the reference syncs twice per batch, :144-146).
"""
from __future__ import annotations

import os
import time

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
from scipy.ndimage import rotate as _spline_rotate

from . import _lib as L
from .configs import settings
from .utils.Mnist2D import TensorDataset

DC_AUG_PARAM = {"crop": 4, "scale": 0.2, "rotate": 45, "noise": 0.001, "strategy": "crop_scale_rotate"}  # mnist_distill.py:67


def populate_img(occ_grid: torch.Tensor, img_array: torch.Tensor) -> torch.Tensor:
    """img_array[i, 0, int(p[1]), int(p[2])] = 1 for every point p of cloud i (mnist_distill.py:163-168).  Device tensors only."""
    if not (occ_grid.is_cuda and img_array.is_cuda):
        raise RuntimeError("populate_img runs on the GPU (libdd3d_hip.so); there is no CPU path")
    n, pts = occ_grid.shape[0], occ_grid.shape[1]
    if img_array.shape[0] < n or img_array.shape[1] != 1 or not img_array.is_contiguous() or img_array.dtype != torch.float32:
        raise ValueError("img_array must be a contiguous float32 [n_clouds,1,H,W] tensor")
    xyz = occ_grid.detach().to(torch.float32).contiguous()
    err = torch.zeros(1, dtype=torch.int32, device=xyz.device)
    L.call("dd3d_populate_img", L.ptr(xyz), n, pts, img_array.shape[2], img_array.shape[3], L.ptr(img_array), L.ptr(err), L.stream())
    if int(err.item()):
        raise IndexError("populate_img: a point lies outside the %dx%d image" % (img_array.shape[2], img_array.shape[3]))
    return img_array


def _centre(canvas: torch.Tensor, h: int, w: int) -> torch.Tensor:
    r, c = int((canvas.shape[-2] - h) / 2), int((canvas.shape[-1] - w) / 2)
    return canvas[..., r:r + h, c:c + w]


def augment(images: torch.Tensor, device, dc_aug_param=DC_AUG_PARAM) -> torch.Tensor:
    """One random transform per image, in place (mnist_distill.py:65-124).  Returns ``images``."""
    if dc_aug_param is None or dc_aug_param["strategy"] == "none":
        return images
    pad, scale, max_angle, sigma = dc_aug_param["crop"], dc_aug_param["scale"], dc_aug_param["rotate"], dc_aug_param["noise"]
    kinds = dc_aug_param["strategy"].split("_")
    n, ch, H, W = images.shape
    host = images.detach().to("cpu", torch.float32)
    fill = [float(torch.mean(images[:, c])) for c in range(ch)]  # per-channel mean of the batch BEFORE any transform (:76-78)
    out = host.clone()
    for i in range(n):
        kind = np.random.permutation(kinds)[0]
        img = host[i]
        if kind == "crop":      # paste into a mean-filled frame `pad` wider on every side, cut a window at a random shift (:80-86)
            frame = torch.empty(ch, H + 2 * pad, W + 2 * pad)
            for c in range(ch):
                frame[c] = fill[c]
            frame[:, pad:pad + H, pad:pad + W] = img
            r, c = np.random.permutation(2 * pad)[0], np.random.permutation(2 * pad)[0]
            out[i] = frame[:, r:r + H, c:c + W]
        elif kind == "scale":   # nearest-neighbour resize to a random size, centred on a zero canvas, centre window (:88-99)
            h = int(np.random.uniform(1 - scale, 1 + scale) * H)
            w = int(np.random.uniform(1 - scale, 1 + scale) * H)
            small = F.interpolate(img[None], [h, w])[0]
            side = max(h, w, H, W)
            canvas = torch.zeros(ch, side, side)
            r, c = int((side - h) / 2), int((side - w) / 2)
            canvas[:, r:r + h, c:c + w] = small
            out[i] = _centre(canvas, H, W)
        elif kind == "rotate":  # cubic-spline rotation about the centre, enlarged frame filled with the mean, centre window (:101-105)
            turned = _spline_rotate(img.numpy(), angle=np.random.randint(-max_angle, max_angle), axes=(-2, -1), cval=np.mean(fill))
            out[i] = _centre(torch.from_numpy(np.ascontiguousarray(turned)).float(), H, W)
        elif kind == "noise":   # (:107-108)
            out[i] = img + sigma * torch.randn(ch, H, W)
    with torch.no_grad():
        images.copy_(out.to(images.device))
    return images


def epoch(mode, dataloader, net, optimizer, criterion, aug):
    """One pass over ``dataloader`` (mnist_distill.py:126-158).  Returns (mean loss, accuracy).  ``aug`` is ignored, as upstream."""
    dev = settings.device
    net, criterion = net.to(dev), criterion.to(dev)
    net.train() if mode == "train" else net.eval()
    loss_sum = torch.zeros((), dtype=torch.float64, device=dev)
    hits = torch.zeros((), dtype=torch.int64, device=dev)
    seen = 0
    for datum in dataloader:
        img = augment(datum[0].float().to(dev), dev)
        lab = datum[1].long().to(dev)
        output = net(img)
        loss = criterion(output, lab)
        loss_sum += loss.detach().double() * lab.shape[0]
        hits += (output.detach().argmax(-1) == lab).sum()
        seen += lab.shape[0]
        if mode == "train":
            optimizer.zero_grad()
            loss.backward()
            optimizer.step()
    return loss_sum.item() / seen, hits.item() / seen


def get_time():
    return str(time.strftime("[%Y-%m-%d %H:%M:%S]", time.localtime()))


def evaluate_synset(it_eval, net, images_train, labels_train, testloader, iteration, Epoch=1000, export_cad_dir=None):
    """Train ``net`` on the rasterised synthetic set for ``Epoch``+1 epochs (lr / 10 after Epoch//2 + 1, fresh SGD as upstream) and
    test it (mnist_distill.py:169-198).  Returns (net, acc_train, acc_test)."""
    dev = settings.device
    net = net.to(dev)
    images_train, labels_train = images_train.to(dev), labels_train.to(dev)
    lr = float(settings.modelconfig.getfloat("classifier_lr"))
    optimizer = torch.optim.SGD(net.parameters(), lr=lr, momentum=0.9, weight_decay=0.0005)
    criterion = nn.CrossEntropyLoss().to(dev)
    mnist_imgs = populate_img(images_train, torch.zeros((images_train.shape[0], 1, 28, 28), dtype=torch.float, device=dev))
    if export_cad_dir is not None:
        from .export import save_image
        save_image(mnist_imgs, os.path.join(export_cad_dir, "vis_iter%d.png" % iteration), nrow=settings.cad_per_class)
    trainloader = torch.utils.data.DataLoader(TensorDataset(mnist_imgs, labels_train), batch_size=256, shuffle=True, num_workers=0)
    start = time.time()
    loss_train = acc_train = 0.0
    for ep in range(Epoch + 1):
        loss_train, acc_train = epoch("train", trainloader, net, optimizer, criterion, aug=False)
        if ep == Epoch // 2 + 1:
            lr *= 0.1
            optimizer = torch.optim.SGD(net.parameters(), lr=lr, momentum=0.9, weight_decay=0.0005)
    time_train = time.time() - start
    _, acc_test = epoch("test", testloader, net, optimizer, criterion, aug=False)
    print("%s Evaluate_%02d: epoch = %04d train time = %d s train loss = %.6f train acc = %.4f, test acc = %.4f"
          % (get_time(), it_eval, Epoch, int(time_train), loss_train, acc_train, acc_test))
    return net, acc_train, acc_test
