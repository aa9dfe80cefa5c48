"""How well do two independent step graphs overlap on one GPU?  A = a real-heavy job (cfg2 with ipc 1), B = a synthetic-heavy small job (10
classes, ipc 10, FOUR real clouds per class: about the synthetic pass alone).  Both replayed every step on two streams vs one after the
other.  (Probe for a pipelined real / synthetic split of the step.)  Every stage prints when it is done."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dd3d_amd.fused_step import FusedDistiller
from dd3d_amd.models.MEConv import MEConvExp
from dd3d_amd.synthetic import init_synthetic, mnist3d_like
dev = torch.device("cuda:0")
torch.manual_seed(0)
images, _, idx = mnist3d_like(10, 512, seed=0)
net = MEConvExp(in_channel=3, out_channel=10)
P = lambda *a: print(*a, flush=True)
A = FusedDistiller(images, idx, net, ipc=1, device=dev, seed=1, image_syn=init_synthetic(10, 1, 250, seed=0), graph=True)
B = FusedDistiller(images, idx, net, ipc=10, batch_size=4, device=dev, seed=2, image_syn=init_synthetic(10, 10, 250, seed=1), graph=True)
for i in range(3):
    A.outer_step(); torch.cuda.synchronize(); P("A default stream step", i)
for i in range(3):
    B.outer_step(); torch.cuda.synchronize(); P("B default stream step", i)
sA, sB = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
def run(which, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        if "A" in which:
            sA.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(sA):
                A.outer_step()
        if "B" in which:
            sB.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(sB):
                B.outer_step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
for w in ("A", "B", "AB"):
    P("warm", w, run(w, 3))
for w in ("A", "B", "AB", "A", "B", "AB"):
    P(f"{w:2s}: {run(w, 100):.3f} ms per step")
