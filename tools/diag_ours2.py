"""cfg2 'ours' step 2, class 3: eager vs oracle, cloud-local maps vs generic maps (diagnostic, GPU box)"""
import numpy as np, torch, sys
sys.path.insert(0, ".")
from oracle import ref_model as R
from dd3d_amd.synthetic import init_synthetic, mnist3d_like
from dd3d_amd.fused_step import FusedDistiller
from dd3d_amd.models.MEConv import MEConvExp
from dd3d_amd.sparse import CloudCoordinateManager, CoordinateManager
from dd3d_amd.ops import DensePlacement
from tests.detgen import det_state_dict
DEV = "cuda:0"
def rel(a, b):
    a = torch.as_tensor(a).detach().double().cpu(); b = torch.as_tensor(b).detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))
torch.manual_seed(0)
net = MEConvExp(in_channel=3, out_channel=10); net.load_state_dict(det_state_dict(net, "net.MEConvExp")); net = net.to(DEV).train()
images, _, idx = mnist3d_like(10, 256, seed=0)
f = FusedDistiller(images, idx, net, ipc=10, device=DEV, seed=7, image_syn=init_synthetic(10, 10, 250, seed=0), dist_opt="ours", graph=False)
rng = np.random.RandomState(7)
for step in range(2):
    f.outer_step(); [rng.permutation(np.asarray(idx[c])) for c in range(10)]
torch.save(f.image_syn.cpu(), "gpurun_out/syn_step2.pt")
syn = f.image_syn.clone().clamp_(0, 27); syn[:, :, 0] = 0
# maps of the synthetic batch of class 3 alone
xyz = syn[30:40].reshape(-1, 3)
coords = torch.cat([torch.arange(10, device=DEV).repeat_interleave(250)[:, None].float(), xyz], 1).contiguous()
pt_off = (torch.arange(11, device=DEV) * 250).to(torch.int32)
cg = torch.zeros(10, dtype=torch.int32, device=DEV); gco = torch.tensor([0, 10], dtype=torch.int32, device=DEV)
ref = CoordinateManager(coords, 10); dref = DensePlacement(ref, 4, (1, 7, 7), cg, 1, gco)
cm = CloudCoordinateManager(coords, pt_off, 10, 250, (1, 7, 7), cg, 1, gco); dcm = DensePlacement(cm, 4, (1, 7, 7), cg, 1, gco)
ref.sync(); cm.sync()
Ms = [ref.num_rows(s) for s in (1, 2, 4)]
print("rows", Ms, [cm.num_rows(s) for s in (1, 2, 4)])
for s, M in zip((1, 2, 4), Ms):
    print(" stride", s, "C", torch.equal(cm.coordinates(s), ref.coordinates(s)), "seg", torch.equal(cm.seg_off(s), ref.seg_off(s)), "uidx", torch.equal(cm.levels[s].uidx[:M], ref.levels[s].uidx[:M]))
print(" par2", torch.equal(cm.parent(2)[:Ms[0]], ref.parent(2)[:Ms[0]]), "par4", torch.equal(cm.parent(4)[:Ms[1]], ref.parent(4)[:Ms[1]]))
for key in ((1, 1, 3), (2, 2, 3), (1, 2, 2), (2, 4, 2)):
    Mo = ref.num_rows(key[1])
    print(" kmap", key, torch.equal(cm.kernel_map(*key)[0][:, :Mo], ref.kernel_map(*key)[0][:, :Mo]))
print(" gmin", dcm.gmin.tolist(), dref.gmin.tolist(), "site", torch.equal(dcm.site[:Ms[2]], dref.site[:Ms[2]]), "vox_at", torch.equal(dcm.vox_at, dref.vox_at))
print(" C4 min/max of the class", ref.coordinates(4).min(0).values.tolist(), ref.coordinates(4).max(0).values.tolist())
# the eager fused step vs the oracle at step 2
onet = R.MEConvExp(in_channel=3, out_channel=10); onet.load_state_dict(det_state_dict(onet, "net.MEConvExp")); onet = onet.double().train()
osyn = f.image_syn.detach().cpu().double().clone(); R.clamp_synthetic_(osyn); osyn.requires_grad_()
order = [rng.permutation(np.asarray(idx[c]))[:256] for c in range(10)]
rf = R.matching_step(onet, osyn, [images[o] for o in order], 10, "ours", False)
f.outer_step()
print("eager step 2: grad rel", rel(f.last["grad"], rf["grad"]))
for i in range(8):
    print("  class 3 param", i, "gw_syn err %.2e" % rel(f.last["gw_syn"][i][3].reshape(-1), rf["gw_syn"][3][i].reshape(-1)))
# which sites of the classifier gradient differ
a = f.last["gw_syn"][6][3].reshape(10, 128, 28, 28).cpu().double(); b = rf["gw_syn"][3][6].reshape(10, 128, 28, 28)
d = (a - b).abs().amax((0, 1)); print(" site err map rows with error:", [(int(y), int(x), float(d[y, x])) for y, x in torch.nonzero(d > 1e-6 * b.abs().max())][:20])
print(" oracle nonzero sites:", sorted(set((int(y), int(x)) for y, x in torch.nonzero(b.abs().amax((0, 1)) > 0)))[:60])
print(" gpu nonzero sites:", sorted(set((int(y), int(x)) for y, x in torch.nonzero(a.abs().amax((0, 1)) > 0)))[:60])
