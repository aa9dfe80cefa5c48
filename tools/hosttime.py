"""Where does the host spend its time inside one fused step?  (wall-clock marks, no extra syncs)"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dd3d_amd.fused_step as fs
from dd3d_amd import _lib as L
from dd3d_amd.models.MEConv import MEConvExp
from dd3d_amd.synthetic import init_synthetic, mnist3d_like

dev = torch.device("cuda:0")
images, _, idx = mnist3d_like(10, 512, seed=0)
d = fs.FusedDistiller(images, idx, MEConvExp(in_channel=3, out_channel=10), ipc=10, device=dev, seed=1, image_syn=init_synthetic(10, 10, 250, 0))
marks = []
orig_init, orig_run, orig_loss, orig_b2p = fs._Pass.__init__, fs._Pass.run, fs.grouped_match_loss, fs._Pass.backward_to_points
def wrap(name, f):
    def g(*a, **k):
        t = time.perf_counter(); r = f(*a, **k); marks.append((name, time.perf_counter() - t)); return r
    return g
fs._Pass.__init__ = wrap("prepare", orig_init); fs._Pass.run = wrap("run", orig_run)
fs.grouped_match_loss = wrap("loss", orig_loss); fs._Pass.backward_to_points = wrap("bwd2", orig_b2p)
ncalls = [0]
oc = L.call
def cc(name, *a):
    ncalls[0] += 1; oc(name, *a)
L.call = cc
for it in range(6):
    marks.clear(); ncalls[0] = 0
    torch.cuda.synchronize(); t0 = time.perf_counter()
    d.outer_step()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("host enqueue ms", (t1 - t0) * 1e3, "total ms", (t2 - t0) * 1e3, "C-ABI calls", ncalls[0])
for n, t in marks: print(f"  {n:8s} {t*1e3:7.3f} ms")
