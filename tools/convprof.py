"""Per-workgroup timestamps of k_conv128_rows on the merged cfg2 batch (scheduled launch), and its per-step segment times with 1, 2, 3
workgroups per CU.  Needs the instrumented build:  make -C dd3d_amd/csrc prof && DD3D_LIB=dd3d_amd/libdd3d_hip_prof.so python tools/convprof.py"""
import os, sys, ctypes, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dd3d_amd import _lib as L
from dd3d_amd.sparse import CoordinateManager
from dd3d_amd.synthetic import mnist3d_like, init_synthetic
from dd3d_amd.utils.Mnist2D import batch_from_images
dev = torch.device("cuda:0")
images, labels, _ = mnist3d_like(10, 256, seed=0)
b = batch_from_images(images.to(dev), labels.to(dev), 250)
syn = init_synthetic(10, 10, 250, seed=0).to(dev).clamp_(0, 27); syn[:, :, 0] = 0
sc = torch.cat([torch.arange(100, device=dev, dtype=torch.float32).repeat_interleave(250)[:, None], syn.reshape(-1, 3)], 1)
rc = b["coordinates"].clone(); rc[:, 0] += 100
cm = CoordinateManager(torch.cat([sc, rc]), 2660)
nbr, ld = cm.kernel_map(2, 2, 3); M = cm.num_rows(2)
X = torch.randn(M, 128, device=dev); W = torch.randn(27, 128, 128, device=dev) * 0.05; Y = torch.empty(M, 128, device=dev)
nblk = int(L.lib.dd3d_conv_schedule_blocks(M))
nbr_s = torch.empty_like(nbr); tmask = torch.empty((M + 31) // 32, dtype=torch.int32, device=dev)
perm = torch.empty(M, dtype=torch.int32, device=dev); tmap = torch.empty(nblk, dtype=torch.int32, device=dev)
ws = torch.empty(int(L.lib.dd3d_conv_schedule_ws_bytes(M)), dtype=torch.uint8, device=dev)
L.call("dd3d_conv_schedule", L.ptr(nbr), ld, 27, M, None, L.ptr(perm), L.ptr(tmap), L.ptr(nbr_s), L.ptr(tmask), L.ptr(ws), ws.numel(), L.stream())
def run(): L.call("dd3d_conv128_ordered", L.ptr(X), M, L.ptr(W), L.ptr(nbr), ld, 27, M, None, 0, L.ptr(perm), L.ptr(tmap), L.ptr(nbr_s), L.ptr(tmask), L.ptr(Y), L.stream())
for _ in range(5): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); run(); e1.record(); torch.cuda.synchronize()
print("launch us", e0.elapsed_time(e1) * 1e3, "blocks", nblk, "M", M)
buf = np.zeros(4096 * 8, dtype=np.int64)
rc_ = L.lib.dd3d_prof_read(ctypes.c_void_p(buf.ctypes.data)); assert rc_ == 0
p = buf.reshape(1024, 4, 8)[:nblk].copy()
p[:, :, 4] &= 0xFF; p[:, :, 5] &= 0xFF
tm = tmap.cpu().numpy()
live = tm >= 0
p = p[live]
t0 = p[:, :, 0].min()
us = lambda a: (a - t0) / 100.0   # 100 MHz wall clock
st, pe, le, en = us(p[:, :, 0]), us(p[:, :, 1]), us(p[:, :, 2]), us(p[:, :, 3])
print("kernel span us", en.max(), " first start", st.min(), "last start", st.max())
print("prologue us: mean %.2f max %.2f" % ((pe - st).mean(), (pe - st).max()))
print("loop us: mean %.2f max %.2f" % ((le - pe).mean(), (le - pe).max()))
print("epilogue us: mean %.2f max %.2f" % ((en - le).mean(), (en - le).max()))
ksteps, wsteps = p[:, :, 4] * 4, p[:, :, 5] * 4
print("us per tile step (loop/ksteps) mean %.3f ; ideal single-wave 4096 cyc @2.4GHz = 1.707" % ((le - pe) / np.maximum(ksteps, 1)).mean())
hw, xcc = p[:, 0, 6], p[:, 0, 7]
cu = ((xcc & 0xF) << 8) | ((hw >> 13) & 7) << 5 | ((hw >> 12) & 1) << 4 | ((hw >> 8) & 0xF)
ucu = np.unique(cu)
print("distinct CUs", len(ucu))
busy, endt, steps, nwg = [], [], [], []
for c in ucu:
    m = cu == c
    endt.append(en[m].max()); steps.append(ksteps[m][:, 0].sum()); nwg.append(m.sum())
endt, steps, nwg = np.array(endt), np.array(steps), np.array(nwg)
print("per-CU end time: min %.1f mean %.1f max %.1f" % (endt.min(), endt.mean(), endt.max()))
print("per-CU steps: min %d mean %.1f max %d ; WGs/CU min %d max %d" % (steps.min(), steps.mean(), steps.max(), nwg.min(), nwg.max()))
print("corr(end, steps) %.3f" % np.corrcoef(endt, steps)[0, 1])
print("us per CU-step (end/steps): mean %.3f" % (endt / steps).mean())
# time histogram of how many WGs are in their loop
grid = np.arange(0, en.max(), 5.0)
act = [(((pe[:, 0] <= t) & (le[:, 0] > t)).sum()) for t in grid]
print("WGs in main loop at t (5 us grid):", act)
# a sample CU timeline
c = ucu[np.argmax(endt)]
m = cu == c
print("slowest CU", hex(c), [(round(float(a), 1), round(float(b_), 1), round(float(c_), 1), round(float(d), 1), int(k)) for a, b_, c_, d, k in zip(st[m][:, 0], pe[m][:, 0], le[m][:, 0], en[m][:, 0], ksteps[m][:, 0])])
# ---- co-residency experiment: n heavy tiles (all 9 offsets) per CU
present = (nbr[:, :M] >= 0)[:, perm.long()]
pm = torch.cat([present, present.new_zeros(27, (-M) % 128)], 1).reshape(27, -1, 128).any(2).sum(0).cpu().numpy()
order = np.argsort(-pm, kind="stable")
print("tile cost histogram (offsets: tiles)", {int(c): int((pm == c).sum()) for c in np.unique(pm)})
for per_cu in (1, 2, 3):
    n = 256 * per_cu
    n = min(n, len(order))
    t2 = torch.full((nblk,), -1, dtype=torch.int32)
    t2[:n] = torch.from_numpy(order[:n].astype(np.int32))
    tmap.copy_(t2.to(dev))
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0.record(); run(); e1.record(); torch.cuda.synchronize()
    L.lib.dd3d_prof_read(ctypes.c_void_p(buf.ctypes.data))
    raw = buf.reshape(1024, 4, 8)[:n].copy()
    sg = np.stack([(raw[:, :, 4] >> 8) & 0xFFFFFF, raw[:, :, 4] >> 32, (raw[:, :, 5] >> 8) & 0xFFFFFF, raw[:, :, 5] >> 32], -1) / 100.0
    p = raw; p[:, :, 4] &= 0xFF; p[:, :, 5] &= 0xFF
    print("   per-step segments us (issue loads, multiply, wait+store W, barrier):", (sg / np.maximum(p[:, :, 4:5] * 4 - 1, 1)).mean((0, 1)).round(3))
    t0 = p[:, :, 0].min()
    pe, le, en = us(p[:, :, 1]), us(p[:, :, 2]), us(p[:, :, 3])
    ks = np.maximum(p[:, :, 4] * 4, 1)
    print(f"{per_cu} WG/CU ({n} heaviest tiles): launch {e0.elapsed_time(e1)*1e3:.1f} us; prologue end {pe.mean():.1f}; us/step/WG {np.mean((le-pe)/ks):.3f}; first 256: {np.mean(((le-pe)/ks)[:256]):.3f}; span {en.max():.1f}")
