"""Decay of mean |xyz-grad| (sum of squares) over the reference's 2500 iterations (default.ini total_iterations), block means of 250."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dd3d_amd.fused_step import FusedDistiller
from dd3d_amd.models.MEConv import MEConvExp
from dd3d_amd.synthetic import init_synthetic, mnist3d_like
DEV = "cuda:0"
images, _, idx = mnist3d_like(10, 256, seed=0)
for metric in ("modmse", "ours"):
    torch.manual_seed(0)
    net = MEConvExp(in_channel=3, out_channel=10).to(DEV).train()
    with FusedDistiller(images, idx, net, ipc=1, device=DEV, seed=1, image_syn=init_synthetic(10, 1, 250, seed=0), dist_opt=metric, graph=True) as f:
        vals, losses = [], []
        for it in range(2500):
            l = f.outer_step()
            vals.append(f.last["grad"].abs().mean().double())
            losses.append(l.double().clone())
        v, l = torch.stack(vals).cpu(), torch.stack(losses).cpu()
    print(metric, "block means of 250 iterations:", " ".join(f"{float(v[i:i+250].mean()):.3e}" for i in range(0, 2500, 250)))
    print(metric, "loss block means:", " ".join(f"{float(l[i:i+250].mean()):.4g}" for i in range(0, 2500, 250)))
    print(metric, "mean of iterations 1250..2499:", f"{float(v[1250:].mean()):.3e}", " mean of all 2500:", f"{float(v.mean()):.3e}")
