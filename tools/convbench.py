"""Micro-benchmark of the 128-channel sparse convolution on the cfg2 real batch (10 x 256 stroke digits) and the synthetic batch.
usage: python tools_convbench.py   (set DD3D_CONV128_DENSE=1 to time the dense-tile kernel instead)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dd3d_amd import _lib as L
from dd3d_amd.sparse import CoordinateManager
from dd3d_amd.synthetic import mnist3d_like, init_synthetic
from dd3d_amd.utils.Mnist2D import batch_from_images

dev = torch.device("cuda:0")
images, labels, _ = mnist3d_like(10, 256, seed=0)
b = batch_from_images(images.to(dev), labels.to(dev), 250)
syn = init_synthetic(10, 10, 250, seed=0).to(dev).clamp_(0, 27)
syn[:, :, 0] = 0
sc = torch.cat([torch.arange(100, device=dev, dtype=torch.float32).repeat_interleave(250)[:, None], syn.reshape(-1, 3)], 1)
for name, coords, B in (("real", b["coordinates"], 2560), ("syn", sc, 100)):
    cm = CoordinateManager(coords, B)
    nbr, ld = cm.kernel_map(2, 2, 3)
    M = cm.num_rows(2)
    pairs = int((nbr[:, :M] >= 0).sum())
    X = torch.randn(M, 128, device=dev)
    assert int(nbr[:, :M].max()) < X.shape[0] and ld >= M  # every gathered row exists
    W = torch.randn(27, 128, 128, device=dev) * 0.05
    Y = torch.empty(M, 128, device=dev)
    def run():
        L.call("dd3d_conv_fwd", L.ptr(X), M, 128, L.ptr(W), 128, L.ptr(nbr), ld, 27, M, None, 0, L.ptr(Y), L.stream())
    for _ in range(5): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 50
    e0.record()
    for _ in range(n): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    act = int((nbr[:, :M] >= 0).any(1).sum())
    print(f"{name}: M={M} pairs={pairs} active offsets={act} density={pairs/(M*act):.2f}  {us:8.1f} us  {2*pairs*128*128/us/1e6:6.1f} TFLOP/s (alg)")
