"""Where does the cfg2 'ours' step differ from the fp64 oracle?  (diagnostic, GPU box)"""
import numpy as np, torch, sys
sys.path.insert(0, ".")
from oracle import ref_model as R
from dd3d_amd.synthetic import init_synthetic, mnist3d_like
from dd3d_amd.fused_step import FusedDistiller
from dd3d_amd.models.MEConv import MEConvExp
from tests.detgen import det_state_dict
DEV = "cuda:0"
def rel(a, b):
    a = torch.as_tensor(a).detach().double().cpu(); b = torch.as_tensor(b).detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))
torch.manual_seed(0)
net = MEConvExp(in_channel=3, out_channel=10); net.load_state_dict(det_state_dict(net, "net.MEConvExp")); net = net.to(DEV).train()
images, _, idx = mnist3d_like(10, 256, seed=0)
f = FusedDistiller(images, idx, net, ipc=10, device=DEV, seed=7, image_syn=init_synthetic(10, 10, 250, seed=0), dist_opt="ours", graph=True)
rng = np.random.RandomState(7)
for dt in (torch.float64,):
    onet = R.MEConvExp(in_channel=3, out_channel=10); onet.load_state_dict(det_state_dict(onet, "net.MEConvExp")); onet = onet.to(dt).train()
for step in range(3):
    osyn = f.image_syn.detach().cpu().double().clone(); R.clamp_synthetic_(osyn); osyn.requires_grad_()
    order = [rng.permutation(np.asarray(idx[c]))[:256] for c in range(10)]
    ref = R.matching_step(onet, osyn, [images[o] for o in order], 10, "ours", False)
    o32 = osyn.detach().float().requires_grad_()
    onet32 = R.MEConvExp(in_channel=3, out_channel=10); onet32.load_state_dict(det_state_dict(onet32, "net.MEConvExp")); onet32 = onet32.train()
    ref32 = R.matching_step(onet32, o32, [images[o] for o in order], 10, "ours", False)
    f.outer_step()
    g, rg = f.last["grad"].cpu().double(), ref["grad"]
    print("step", step, "grad rel", rel(g, rg), "fp32-oracle vs fp64-oracle", rel(ref32["grad"], rg), "gpu vs fp32 oracle", rel(g, ref32["grad"]))
    pc = [(c, rel(g[c*10:(c+1)*10], rg[c*10:(c+1)*10])) for c in range(10)]
    print(" per class grad err", ["%d:%.1e" % x for x in pc])
    print(" per class loss err", [("%.1e" % abs((a - b) / b)) for a, b in zip(f.last["per_class"].cpu().double().tolist(), ref["per_class"].tolist())])
    bad = max(pc, key=lambda x: x[1])[0]
    for i in range(8):
        print("  class", bad, "param", i, "gw_syn err %.2e" % rel(f.last["gw_syn"][i][bad].reshape(-1), ref["gw_syn"][bad][i].reshape(-1)),
              "gw_real err %.2e" % rel(f.last["gw_real"][i][bad].reshape(-1), ref["gw_real"][bad][i].reshape(-1)))
    pcl = [(j, rel(g[j], rg[j])) for j in range(bad*10, bad*10+10)]
    print("  per cloud of bad class (scaled to cloud max)", ["%d:%.1e" % x for x in pcl])
