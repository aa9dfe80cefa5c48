"""Isolated timings (HIP events, one kernel family at a time, nothing else on the GPU) of the hot kernels on the cfg2 real batch
(10 classes x 256 stroke digits) -- the per-kernel view that the overlapped step profile cannot give.
usage: python tools/kbench.py [filter-substring]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dd3d_amd import _lib as L
from dd3d_amd.sparse import CoordinateManager
from dd3d_amd.ops import DensePlacement, site_major, _site_strides, sl_fwd_call, sl_bwd_x_call
from dd3d_amd.synthetic import mnist3d_like
from dd3d_amd.utils.Mnist2D import batch_from_images

flt = sys.argv[1] if len(sys.argv) > 1 else ""
dev = torch.device("cuda:0")
G, per = 10, 256
images, labels, _ = mnist3d_like(G, per, seed=0)
b = batch_from_images(images.to(dev), labels.to(dev), 250)
B = G * per
cg = torch.arange(G, device=dev, dtype=torch.int32).repeat_interleave(per)
gco = (torch.arange(G + 1, device=dev) * per).to(torch.int32)
cm = CoordinateManager(b["coordinates"], B)
k0, ld0 = cm.kernel_map(1, 1, 3); p0, pl0 = cm.kernel_map(1, 2, 2)
k1, ld1 = cm.kernel_map(2, 2, 3); p1, pl1 = cm.kernel_map(2, 4, 2)
seg1, seg2 = cm.seg_off(1), cm.seg_off(2)
dp = DensePlacement(cm, 4, (1, 28, 28), cg, G, gco)
cm.sync()
M0, M1, M2 = cm.num_rows(1), cm.num_rows(2), cm.num_rows(4)
gro1, gro2 = seg1[gco.long()].contiguous(), seg2[gco.long()].contiguous()
N = cm.N
print(f"N={N} M0={M0} M1={M1} M2={M2} pairs0={int((k0[:, :M0] >= 0).sum())} pairs1={int((k1[:, :M1] >= 0).sum())}")
R = lambda *s: torch.randn(*s, device=dev)
F0, W0, Y0, xh0, is0 = R(M0, 3), R(27, 3, 128) * .1, R(M0, 128), torch.empty(M0, 128, device=dev), torch.empty(B, 128, device=dev)
w, bb = torch.ones(128, device=dev), torch.zeros(128, device=dev)
P0, arg0 = torch.empty(M1, 128, device=dev), torch.empty(M1, 128, device=dev, dtype=torch.int32)
W1, Y1 = R(27, 128, 128) * .05, torch.empty(M1, 128, device=dev)
P1, arg1 = torch.empty(M2, 128, device=dev), torch.empty(M2, 128, device=dev, dtype=torch.int32)
xh1, is1 = torch.empty(M1, 128, device=dev), torch.empty(B, 128, device=dev)
J, S = 10, 784
Wc = R(J, 128 * S) * .01
Wsm = site_major(Wc, 128, S); sst = _site_strides(J, 128, S)
logits, dl = torch.empty(B, J, device=dev), R(B, J)
gP1 = R(M2, 128)
gP0, gY0, gY1, dW1, dW0 = R(M1, 128), torch.empty(M0, 128, device=dev), R(M1, 128), torch.empty(G, 27, 128, 128, device=dev), torch.empty(G, 27, 3, 128, device=dev)
gw_w, gw_b = torch.empty(G, 128, device=dev), torch.empty(G, 128, device=dev)
import ctypes
gcw, cwst = torch.empty(G, J, S, 128, device=dev), (ctypes.c_int64 * 4)(J * 128 * S, 128 * S, 1, 128)
st = L.stream()
ws = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
nbr_tmp = torch.empty_like(k0)
dF0 = torch.empty(M0, 3, device=dev)
lv1, lv2 = cm.levels[1], cm.levels[2]

K = {
 "voxel_mean_fwd": (lambda: L.call("dd3d_voxel_mean_fwd", L.ptr(b["features"]), N, 3, L.ptr(cm.vox_off), L.ptr(cm.vox_pts), M0, None, L.ptr(F0), st), 28 * N),
 "kernel_map k3 level1": (lambda: L.call("dd3d_kernel_map", L.ptr(lv1.C), None, M0, L.ptr(lv1.table), lv1.cap, L.ptr(lv1.C), 3, 1, L.ptr(nbr_tmp), ld0, st), 16 * M0 + 108 * M0),
 "conv3_fwd": (lambda: L.call("dd3d_conv_fwd", L.ptr(F0), M0, 3, L.ptr(W0), 128, L.ptr(k0), ld0, 27, M0, None, 0, L.ptr(Y0), st), 512 * M0 + 12 * M0 + 108 * M0),
 "inorm_fwd L1": (lambda: L.call("dd3d_inorm_fwd", L.ptr(Y0), L.ptr(seg1), B, 128, L.ptr(xh0), L.ptr(is0), st), 1024 * M0),
 "arp_fwd L1": (lambda: L.call("dd3d_affine_relu_pool_fwd", L.ptr(xh0), L.ptr(w), L.ptr(bb), 1, 128, L.ptr(p0), pl0, 8, M1, None, L.ptr(P0), L.ptr(arg0), st), 512 * M0 + 1024 * M1),
 "conv128_fwd": (lambda: L.call("dd3d_conv_fwd", L.ptr(P0), M1, 128, L.ptr(W1), 128, L.ptr(k1), ld1, 27, M1, None, 0, L.ptr(Y1), st), 0),
 "inorm_fwd L2": (lambda: L.call("dd3d_inorm_fwd", L.ptr(Y1), L.ptr(seg2), B, 128, L.ptr(xh1), L.ptr(is1), st), 1024 * M1),
 "arp_fwd L2": (lambda: L.call("dd3d_affine_relu_pool_fwd", L.ptr(xh1), L.ptr(w), L.ptr(bb), 1, 128, L.ptr(p1), pl1, 8, M2, None, L.ptr(P1), L.ptr(arg1), st), 512 * M1 + 1024 * M2),
 "sl_fwd": (lambda: sl_fwd_call(P1, dp, None, Wsm, sst, J, None, logits), 512 * M2 * (1 + J)),
 "sl_bwd_x": (lambda: sl_bwd_x_call(dl, dp, None, Wsm, sst, J, 0, P1), 512 * M2 * (1 + J)),
 "sl_bwd_w": (lambda: L.call("dd3d_sparse_linear_bwd_w", L.ptr(dl), L.ptr(P1), L.ptr(dp.vox_at), S, B, L.ptr(gco), G, 128, J, L.ptr(gcw), cwst, gcw.numel(), st), 512 * M2 + 4 * S * B + 4 * G * J * S * 128),
 "arp_inorm_bwd L1": (lambda: L.call("dd3d_arp_inorm_bwd", L.ptr(gP0), L.ptr(arg0), L.ptr(cm.parent(2)), L.ptr(xh0), L.ptr(is0), L.ptr(w), L.ptr(bb), 128, L.ptr(seg1), B,
                                     L.ptr(cg), L.ptr(gco), G, L.ptr(gY0), None, 0, None, None, L.ptr(gw_w), L.ptr(gw_b), L.ptr(ws), ws.numel(), st), 1024 * M0 + 1024 * M1),
 "arp_inorm_bwd L2": (lambda: L.call("dd3d_arp_inorm_bwd", L.ptr(gP1), L.ptr(arg1), L.ptr(cm.parent(4)), L.ptr(xh1), L.ptr(is1), L.ptr(w), L.ptr(bb), 128, L.ptr(seg2), B,
                                     L.ptr(cg), L.ptr(gco), G, L.ptr(gY1), None, 0, None, None, L.ptr(gw_w), L.ptr(gw_b), L.ptr(ws), ws.numel(), st), 1024 * M1 + 1024 * M2),
 "conv128_bww": (lambda: L.call("dd3d_conv_bwd_weight", L.ptr(P0), 128, L.ptr(gY1), 128, L.ptr(k1), ld1, 27, M1, L.ptr(gro2), G, L.ptr(dW1), L.ptr(ws), ws.numel(), st), 0),
 "conv128_bww_clouds": (lambda: L.call("dd3d_conv128_bwd_weight_clouds", L.ptr(P0), L.ptr(gY1), L.ptr(k1), ld1, 27, M1, L.ptr(seg2), B, L.ptr(gro2), G, L.ptr(dW1), L.ptr(cm.err), L.ptr(ws), ws.numel(), st), 0),
 "conv3_bwd_data (21.6k rows)": (lambda: L.call("dd3d_conv_bwd_data", L.ptr(gY0), 21600, None, 128, L.ptr(W0), 3, L.ptr(k0), ld0, 27, L.ptr(dF0), st), 21600 * (512 + 12 + 108)),
 "conv3_bww": (lambda: L.call("dd3d_conv_bwd_weight", L.ptr(F0), 3, L.ptr(gY0), 128, L.ptr(k0), ld0, 27, M0, L.ptr(gro1), G, L.ptr(dW0), L.ptr(ws), ws.numel(), st), 512 * M0 + 120 * M0),
}
from dd3d_amd.sparse import CloudCoordinateManager
from dd3d_amd.synthetic import init_synthetic
_syn = init_synthetic(10, 10, 250, seed=0).to(dev).clamp_(0, 27)
_syn[:, :, 0] = 0
_sc = torch.cat([torch.arange(100, device=dev, dtype=torch.float32).repeat_interleave(250)[:, None], _syn.reshape(-1, 3)], 1).contiguous()
_spt = (torch.arange(101, device=dev) * 250).to(torch.int32)
_scg = torch.arange(10, device=dev, dtype=torch.int32).repeat_interleave(10)
_sgco = (torch.arange(11, device=dev) * 10).to(torch.int32)
K["cloud_maps real (2560 clouds)"] = (lambda: CloudCoordinateManager(b["coordinates"], b["pt_off"], B, 250, (1, 28, 28), cg, G, gco), 0)
K["cloud_maps syn (100 clouds)"] = (lambda: CloudCoordinateManager(_sc, _spt, 100, 250, (1, 28, 28), _scg, 10, _sgco), 0)
# forward chain once so that arg/xhat hold sane values
for k in ("conv3_fwd", "inorm_fwd L1", "arp_fwd L1", "conv128_fwd", "inorm_fwd L2", "arp_fwd L2"):
    K[k][0]()
torch.cuda.synchronize()
pairs1 = int((k1[:, :M1] >= 0).sum())
for name, (fn, nbytes) in K.items():
    if flt not in name:
        continue
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 30
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    extra = f"{nbytes / us / 1e6:6.2f} TB/s (alg bytes)" if nbytes else (f"{2 * pairs1 * 128 * 128 / us / 1e6:6.1f} TFLOP/s (alg)" if "conv128" in name else "")
    print(f"{name:24s} {us:8.1f} us   {extra}")
