"""Cin = Cout = 128 sparse convolution forward over growing prefixes of the cfg2 real batch: which tile configuration wins at which
size.  usage: DD3D_CONV_TILE={32,64,128} [DD3D_CONV_ROWS=0] python tools/convsweep.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dd3d_amd import _lib as L
from dd3d_amd.sparse import CoordinateManager
from dd3d_amd.synthetic import mnist3d_like
from dd3d_amd.utils.Mnist2D import batch_from_images

dev = torch.device("cuda:0")
images, labels, _ = mnist3d_like(10, 256, seed=0)
out = []
for nc in (160, 320, 640, 960, 1280, 1920, 2560):
    b = batch_from_images(images[:nc].to(dev), labels[:nc].to(dev), 250)
    cm = CoordinateManager(b["coordinates"], nc)
    nbr, ld = cm.kernel_map(2, 2, 3)
    M = cm.num_rows(2)
    X = torch.randn(M, 128, device=dev)
    W = torch.randn(27, 128, 128, device=dev) * 0.05
    Y = torch.empty(M, 128, device=dev)
    def run():
        L.call("dd3d_conv_fwd", L.ptr(X), M, 128, L.ptr(W), 128, L.ptr(nbr), ld, 27, M, None, 0, L.ptr(Y), L.stream())
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): run()
    e1.record(); torch.cuda.synchronize()
    out.append(f"{M}:{e0.elapsed_time(e1) / 30 * 1e3:.0f}us")
print("tile", os.environ.get("DD3D_CONV_TILE"), "rows", os.environ.get("DD3D_CONV_ROWS", "1"), " ".join(out))
