"""Timing of the general-purpose convolution kernels (conv_any.hip) on a cfg2-shaped kernel map: python tools/anybench.py
Prints per shape: rows, pairs, microseconds and algorithmic TFLOP/s of forward / input gradient / kernel gradient."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dd3d_amd import _lib as L
from dd3d_amd.sparse import CoordinateManager
from dd3d_amd.synthetic import stroke_digits
from dd3d_amd.utils.Mnist2D import batch_from_images

DEV = "cuda:0"
imgs = stroke_digits(2560, seed=1).to(DEV)
batch = batch_from_images(imgs, torch.zeros(2560, dtype=torch.long, device=DEV), 250)
cm = CoordinateManager(batch["coordinates"].to(DEV), None)
nbr, ld = cm.kernel_map(1, 1, 3)
cm.sync()
M = cm.num_rows(1)
pairs = int((nbr[:, :M] >= 0).sum())
st = L.stream()
print(f"rows {M}, pairs {pairs}")
for K, Cin, Cout in ((27, 1, 128), (27, 128, 64), (27, 64, 32), (27, 32, 16), (1, 128, 128), (27, 128, 128)):
    X, W, g = torch.randn(M, Cin, device=DEV), torch.randn(K, Cin, Cout, device=DEV), torch.randn(M, Cout, device=DEV)
    Y, dX, dW = torch.empty(M, Cout, device=DEV), torch.empty(M, Cin, device=DEV), torch.empty(K, Cin, Cout, device=DEV)
    ws = torch.empty(int(L.lib.dd3d_conv_any_bwd_weight_ws_bytes(K, Cin, Cout)), dtype=torch.uint8, device=DEV)
    tab, ldx = (nbr, ld) if K == 27 else (None, 0)
    calls = {"fwd": lambda: L.call("dd3d_conv_any_fwd", L.ptr(X), M, Cin, L.ptr(W), Cout, L.ptr(tab), ldx, K, M, None, L.ptr(Y), st),
             "bwd_data": lambda: L.call("dd3d_conv_any_bwd_data", L.ptr(g), M, None, Cout, L.ptr(W), Cin, L.ptr(tab), ldx, K, L.ptr(dX), st),
             "bwd_weight": lambda: L.call("dd3d_conv_any_bwd_weight", L.ptr(X), Cin, L.ptr(g), Cout, L.ptr(tab), ldx, K, M, None, L.ptr(dW), L.ptr(ws), ws.numel(), st)}
    p = pairs if K == 27 else M
    out = []
    for name, f in calls.items():
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            f()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100
        out.append(f"{name} {us:8.1f} us {2 * p * Cin * Cout / us / 1e6:6.2f} TF/s")
    print(f"K={K:2d} {Cin:3d}->{Cout:3d}: " + " | ".join(out), flush=True)
