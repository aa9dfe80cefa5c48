"""cfg2 'ours' step 2: is the first-layer discrepancy a ReLU sign flip of a near-zero pre-activation?  (diagnostic, GPU box)"""
import numpy as np, torch, sys
sys.path.insert(0, ".")
from dd3d_amd.synthetic import init_synthetic, mnist3d_like
from dd3d_amd.fused_step import FusedDistiller
from dd3d_amd.models.MEConv import MEConvExp
from tests.detgen import det_state_dict
DEV = "cuda:0"
torch.manual_seed(0)
net = MEConvExp(in_channel=3, out_channel=10); net.load_state_dict(det_state_dict(net, "net.MEConvExp")); net = net.to(DEV).train()
images, _, idx = mnist3d_like(10, 256, seed=0)
f = FusedDistiller(images, idx, net, ipc=10, device=DEV, seed=7, image_syn=init_synthetic(10, 10, 250, seed=0), dist_opt="ours", graph=False)
for step in range(3):
    f.outer_step()
p = f.last["passes"][0]
k0, ld0 = p.maps[0], p.maps[1]
M0 = p.M[0]
seg1 = p.maps[8].cpu().tolist()
F0 = p.F0.double(); W0 = net.features[0].kernel.detach().double()
Y0 = torch.zeros(M0, 128, dtype=torch.float64, device=DEV)
for k in range(27):
    nb = k0[k, :M0].long(); has = nb >= 0
    Y0[has] += F0[nb[has]] @ W0[k]
w = net.features[1].weight.detach().double().reshape(-1); b = net.features[1].bias.detach().double().reshape(-1)
B = p.B
y64 = torch.empty_like(Y0)
for c in range(B):
    a, e = seg1[c], seg1[c + 1]
    if e > a:
        x = Y0[a:e]; mu = x.mean(0); var = ((x - mu) ** 2).mean(0)
        y64[a:e] = (x - mu) / torch.sqrt(var + 1e-8) * w + b
y32 = p.xh0[:M0] * net.features[1].weight.detach().reshape(-1) + net.features[1].bias.detach().reshape(-1)
flip = (y64 > 0) != (y32 > 0)
print("rows", M0, "elements whose ReLU sign differs between fp64 and the fp32 xhat*w+b:", int(flip.sum()))
rows, ch = torch.nonzero(flip, as_tuple=True)
cloud = torch.bucketize(rows.cpu(), torch.tensor(seg1[1:]), right=True)
par = p.cm.parent(2)[:M0].long()
for r, c_, cl in zip(rows.tolist(), ch.tolist(), cloud.tolist()):
    routed = int(p.arg0[par[r], c_]) == r
    print(f" row {r} cloud {cl} channel {c_}: y64 {float(y64[r, c_]):.3e} y32 {float(y32[r, c_]):.3e} routed(argmax) {routed}")
