"""cfg4 full shape: which buffers differ between two eager distillers / between graph and eager (diagnostic, GPU box)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dd3d_amd.fused_step import FusedDistiller
from dd3d_amd.models.MEConv import MEConvExp
from dd3d_amd.synthetic import shell_clouds
from tests.detgen import det_state_dict
DEV = "cuda:0"
nclass, per, ipc, npts = int(os.environ.get("NCLASS", "40")), 256, 10, 1024
clouds = shell_clouds(nclass * per, npts, seed=0)
idx = [list(range(c * per, (c + 1) * per)) for c in range(nclass)]
syn = (torch.rand((nclass * ipc, npts, 3), generator=torch.Generator().manual_seed(0)) * 1.8 - 0.9).float()
def net():
    torch.manual_seed(0)
    m = MEConvExp(3, nclass, num_classes=nclass, dense_extent=(12, 12, 12))
    m.load_state_dict(det_state_dict(m, "net.cfg4"))
    return m.to(DEV).train()
kw = dict(ipc=ipc, num_classes=nclass, num_points=npts, batch_size=per, device=DEV, seed=5, image_syn=syn, real_clouds=clouds,
          quantization_size=0.05, clamp_range=(-0.95, 0.95), zero_first_axis=False, dist_lr=1e-6)
ds = {"eager1": FusedDistiller(None, idx, net(), graph=False, **kw), "eager2": FusedDistiller(None, idx, net(), graph=False, **kw),
      "graph": FusedDistiller(None, idx, net(), graph=True, **kw)}
def fields(f):
    p = f.last["passes"][0]
    M0, M1, M2 = f.row_counts()[0]
    out = {"loss": f.last["per_class"], "grad": f.last["grad"], "F0": p.F0[:M0], "Y0": p.Y0[:M0], "P0": p.P0[:M1], "Y1": p.Y1[:M1], "P1": p.P1[:M2],
           "logits": p.logits, "dl": p.dl, "ga0": p.ga0, "ga1": p.ga1}
    for i in range(8):
        out[f"gw_real{i}"] = f.last["gw_real"].compact(i) if hasattr(f.last["gw_real"], "compact") else f.last["gw_real"][i]
        out[f"gw_syn{i}"] = f.last["gw_syn"].compact(i)
    if getattr(p, "early_out", None):
        out["hP0"] = p.early_out[1]
    return out
for step in range(3):
    res = {}
    for n, f in ds.items():
        f.outer_step()
        torch.cuda.synchronize()
        res[n] = {k: v.clone() for k, v in fields(f).items()}
    for a, b in (("eager1", "eager2"), ("graph", "eager1")):
        diff = [(k, float((res[a][k].double() - res[b][k].double()).abs().max())) for k in res[a] if res[a][k].shape == res[b][k].shape and not torch.equal(res[a][k], res[b][k])]
        shp = [k for k in res[a] if res[a][k].shape != res[b][k].shape]
        print(f"step {step} {a} vs {b}: differ {diff} shapes {shp}", flush=True)
