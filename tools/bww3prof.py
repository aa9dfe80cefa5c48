"""Per-workgroup timestamps of k_conv3_bww on the cfg2 real batch.  Needs: make -C dd3d_amd/csrc prof && DD3D_LIB=dd3d_amd/libdd3d_hip_prof.so"""
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
k0, ld0 = cm.kernel_map(1, 1, 3)
seg1 = cm.seg_off(1)
cm.sync()
M0 = cm.num_rows(1)
gro1 = seg1[gco.long()].contiguous()
F0, gY0 = torch.randn(M0, 3, device=dev), torch.randn(M0, 128, device=dev)
dW0 = torch.empty(G, 27, 3, 128, device=dev)
ws = torch.empty(1 << 28, dtype=torch.uint8, device=dev)
run = lambda: L.call("dd3d_conv_bwd_weight", L.ptr(F0), 3, L.ptr(gY0), 128, L.ptr(k0), ld0, 27, M0, L.ptr(gro1), G, L.ptr(dW0), L.ptr(ws), ws.numel(), L.stream())
buf = np.zeros(4096 * 8, dtype=np.int64)
L.lib.dd3d_prof_read(ctypes.c_void_p(buf.ctypes.data))
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); run(); e1.record(); torch.cuda.synchronize()
print("call us", e0.elapsed_time(e1) * 1e3, "M0", M0)
L.lib.dd3d_prof_read(ctypes.c_void_p(buf.ctypes.data))
p = buf.reshape(4096, 8)
p = p[p[:, 3] > 0]
t0 = p[:, 0].min()
st, en, ls = (p[:, 0] - t0) / 100.0, (p[:, 1] - t0) / 100.0, (p[:, 2] - t0) / 100.0
its = p[:, 3]
seg = [(p[:, 6] & 0xFFFFFFFF) / 100.0, (p[:, 6] >> 32) / 100.0, (p[:, 7] & 0xFFFFFFFF) / 100.0, (p[:, 7] >> 32) / 100.0]
loop = sum(seg)
print(f"productive WGs {len(p)}; kernel span {en.max():.1f} us; iterations total {its.sum()} mean {its.mean():.1f}")
print(f"per WG: start {st.mean():.1f} (max {st.max():.1f}), prologue {np.mean(ls - st):.1f} us, loop {loop.mean():.1f} us ({(loop / its).mean():.2f} us/iteration), epilogue {np.mean(en - ls - loop):.1f} us, life {np.mean(en - st):.1f}")
print("per iteration us: issue + dY wait %.2f, multiply %.2f, gather wait + store %.2f, barrier %.2f" % tuple((s / its).mean() for s in seg))
hw, xcc = p[:, 4], p[:, 5]
cu = ((xcc & 0xF) << 8) | ((hw >> 13) & 7) << 5 | ((hw >> 12) & 1) << 4 | ((hw >> 8) & 0xF)
ucu = np.unique(cu)
cnt = np.array([(cu == c).sum() for c in ucu])
print(f"CUs {len(ucu)}; WGs per CU min {cnt.min()} mean {cnt.mean():.2f} max {cnt.max()}; histogram {np.bincount(cnt).tolist()}")
grid = np.arange(0, en.max(), 5.0)
print("WGs alive at t (5 us grid):", [int(((st <= t) & (en > t)).sum()) for t in grid])
