import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from detgen import det_state_dict
from dd3d_amd.fused_step import FusedDistiller
from dd3d_amd.models.MEConv import MEConvExp
from dd3d_amd.synthetic import init_synthetic, mnist3d_like
DEV = torch.device("cuda:0")
def net():
    torch.manual_seed(0)
    m = MEConvExp(in_channel=3, out_channel=10)
    m.load_state_dict(det_state_dict(m, "net.MEConvExp"))
    return m.to(DEV).train()
images, _, idx = mnist3d_like(4, 30, seed=5)
syn = init_synthetic(4, 3, 250, seed=2)
for poison in (0x7f, 0xff):
    junk = [torch.full((1 << 28,), poison, dtype=torch.uint8, device=DEV) for _ in range(8)]   # 2 GiB of garbage back to the allocator
    del junk
nbad = 0
for trial in range(60):
    kw = dict(ipc=3, num_classes=4, batch_size=24, dist_opt="modmse", device=DEV, seed=11, image_syn=syn, classes=None)
    fg = FusedDistiller(images, idx, net(), graph=True, **kw)
    fe = FusedDistiller(images, idx, net(), graph=False, **kw)
    for step in range(3):
        lg, le = fg.outer_step().item(), fe.outer_step().item()
        bad = [i for i in range(8) if not torch.equal(fg.last["gw_real"][i], fe.last["gw_real"][i])]
        bads = [i for i in range(8) if not torch.equal(fg.last["gw_syn"][i], fe.last["gw_syn"][i])]
        p_g, p_e = fg.last["passes"][0], fe.last["passes"][0]
        M0, M1, M2 = fe.row_counts()[0]
        for name, n in (() if lg == le else (("F0", M0), ("P0", M1), ("P1", M2), ("logits", None), ("dl", None), ("xh0", M0), ("xh1", M1))):
            a, b = getattr(p_g, name), getattr(p_e, name)
            if n is not None: a, b = a[:n], b[:n]
            if not torch.equal(a, b): print("   differs:", name, float((a - b).abs().max()))
        if lg == le and not bad and not bads: continue
        nbad += 1
        print(trial, step, lg, le, "gw_real differ:", bad, "gw_syn differ:", bads, "logits eq", torch.equal(fg.last["logits"][0], fe.last["logits"][0]))
print("mismatching steps:", nbad)
