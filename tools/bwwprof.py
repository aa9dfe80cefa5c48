"""Per-workgroup timestamps of k_conv128_bww on the cfg2 real batch.  Needs: make -C dd3d_amd/csrc prof && DD3D_LIB=dd3d_amd/libdd3d_hip_prof.so"""
import os, sys, ctypes, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dd3d_amd import _lib as L
from dd3d_amd.sparse import CoordinateManager
from dd3d_amd.synthetic import mnist3d_like
from dd3d_amd.utils.Mnist2D import batch_from_images
dev = torch.device("cuda:0")
G, per = 10, 256
images, labels, _ = mnist3d_like(G, per, seed=0)
b = batch_from_images(images.to(dev), labels.to(dev), 250)
B = G * per
gco = (torch.arange(G + 1, device=dev) * per).to(torch.int32)
cm = CoordinateManager(b["coordinates"], B)
k1, ld1 = cm.kernel_map(2, 2, 3)
seg2 = cm.seg_off(2)
cm.sync()
M1 = cm.num_rows(2)
gro2 = seg2[gco.long()].contiguous()
P0, gY1 = torch.randn(M1, 128, device=dev), torch.randn(M1, 128, device=dev)
dW1 = torch.empty(G, 27, 128, 128, device=dev)
ws = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
run = lambda: L.call("dd3d_conv_bwd_weight", L.ptr(P0), 128, L.ptr(gY1), 128, L.ptr(k1), ld1, 27, M1, L.ptr(gro2), G, L.ptr(dW1), L.ptr(ws), ws.numel(), L.stream())
buf = np.zeros(4096 * 8, dtype=np.int64)
L.lib.dd3d_prof_read(ctypes.c_void_p(buf.ctypes.data))   # (clears nothing: read once to page the symbol in)
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); run(); e1.record(); torch.cuda.synchronize()
print("call us", e0.elapsed_time(e1) * 1e3, "M1", M1)
L.lib.dd3d_prof_read(ctypes.c_void_p(buf.ctypes.data))
p = buf.reshape(4096, 8)
p = p[p[:, 3] > 0]
t0 = p[:, 0].min()
st, en = (p[:, 0] - t0) / 100.0, (p[:, 1] - t0) / 100.0
loop, steps = p[:, 2] / 100.0, p[:, 3]
print(f"productive WGs {len(p)}; kernel span {en.max():.1f} us; steps total {steps.sum()} mean {steps.mean():.1f}")
print(f"per WG: life {np.mean(en - st):.1f} us, in step loops {loop.mean():.1f} us ({(loop / steps).mean():.2f} us/step), outside {np.mean(en - st - loop):.1f} us")
print("per step us: multiply+issue %.2f, wait+store dY chunk %.2f, barrier+A wait %.2f" % (((p[:, 6] & 0xFFFFFFFF) / 100.0 / steps).mean(), ((p[:, 6] >> 32) / 100.0 / steps).mean(), (p[:, 7] / 100.0 / steps).mean()))
hw, xcc = p[:, 4], p[:, 5]
cu = ((xcc & 0xF) << 8) | ((hw >> 13) & 7) << 5 | ((hw >> 12) & 1) << 4 | ((hw >> 8) & 0xF)
ucu = np.unique(cu)
tot = np.array([steps[cu == c].sum() for c in ucu]); end = np.array([en[cu == c].max() for c in ucu])
print(f"CUs {len(ucu)}; steps per CU min {tot.min()} mean {tot.mean():.1f} max {tot.max()}; CU end min {end.min():.1f} mean {end.mean():.1f} max {end.max():.1f}; us per CU-step {np.mean(end / tot):.2f}")
grid = np.arange(0, en.max(), 10.0)
print("WGs alive at t (10 us grid):", [int(((st <= t) & (en > t)).sum()) for t in grid])
