"""GPU parity tests of the whole hot path: match_loss, MEConvExp / MEPytorch forward + gradients, and one full matching step,
against (a) the golden vectors produced by the reference's own files and (b) the CPU oracle on seeded inputs.
Tolerance: rel 1e-4 on losses and coordinate gradients (BASELINE.json north_star), fp32."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import ref_model as R
from dd3d_amd.synthetic import init_synthetic, stroke_digits
from tests.detgen import det_state_dict, det_tensor, sample_idx

G = os.path.join(os.path.dirname(__file__), "golden")
EXP_SHAPES = [(27, 3, 128), (1, 128), (1, 128), (27, 128, 128), (1, 128), (1, 128), (10, 100352), (10,)]
PYT_SHAPES = [(128, 3, 1, 3, 3), (128,), (128, 128, 1, 3, 3), (128,), (128, 128, 1, 3, 3), (128,), (10, 1152), (10,)]
DEV = "cuda:0"


def rel_err(a, b):
    a = torch.as_tensor(a).detach().double().cpu()
    b = torch.as_tensor(b).detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


@pytest.mark.parametrize("fam,shapes", [("exp", EXP_SHAPES), ("pyt", PYT_SHAPES)])
@pytest.mark.parametrize("pyt", [False, True])
@pytest.mark.parametrize("metric", ["ours", "modmse", "mse", "cos"])
def test_match_loss_vs_reference_golden(fam, shapes, pyt, metric):
    from dd3d_amd.configs import settings
    from dd3d_amd.distillation_loss import match_loss
    g = np.load(os.path.join(G, "g1_match_loss.npz"))
    real = [det_tensor(s, f"g1.{fam}.real.{i}", 0.05).to(DEV) for i, s in enumerate(shapes)]
    syn = [det_tensor(s, f"g1.{fam}.syn.{i}", 0.05).to(DEV).requires_grad_() for i, s in enumerate(shapes)]
    settings.Pyt = pyt
    try:
        v = match_loss(syn, real, metric, DEV)
        grads = torch.autograd.grad(v * 1.0, syn)
    finally:
        settings.Pyt = False
    key = f"{fam}.{int(pyt)}.{metric}"
    assert v.item() == pytest.approx(float(g[key + ".value"]), rel=1e-5)
    for i, gr in enumerate(grads):
        assert gr.shape == syn[i].shape
        assert gr.double().norm().item() == pytest.approx(g[key + ".gnorm"][i], rel=1e-4, abs=1e-10)
        idx = sample_idx(gr.numel(), f"{fam}.{i}")
        np.testing.assert_allclose(gr.reshape(-1)[idx].cpu().numpy(), g[key + ".gsample"][i][:len(idx)], rtol=1e-4, atol=1e-8)


def test_match_loss_unknown_metric_exits():
    from dd3d_amd.distillation_loss import match_loss
    with pytest.raises(SystemExit):
        match_loss([torch.zeros(3, device=DEV)], [torch.zeros(3, device=DEV)], "nope", DEV)


def test_match_loss_upstream_gradient_scales():
    from dd3d_amd.distillation_loss import match_loss
    real = [det_tensor(s, f"ug.real.{i}", 0.05).to(DEV) for i, s in enumerate(EXP_SHAPES)]
    syn = [det_tensor(s, f"ug.syn.{i}", 0.05).to(DEV).requires_grad_() for i, s in enumerate(EXP_SHAPES)]
    g1 = torch.autograd.grad(match_loss(syn, real, "ours", DEV), syn)
    g3 = torch.autograd.grad(3.0 * match_loss(syn, real, "ours", DEV), syn)
    for a, b in zip(g1, g3):
        assert rel_err(b, 3.0 * a) < 1e-6


def _load(net_cls, oracle_cls, pyt):
    torch.manual_seed(0)
    net = net_cls(3, 10) if pyt else net_cls(in_channel=3, out_channel=10)
    sd = det_state_dict(net, "net." + oracle_cls.__name__)
    net.load_state_dict(sd)
    return net.to(DEV).train()


def _distiller(net, reals, ipc, metric, nclass):
    from dd3d_amd.mnist_distill import Distiller
    from dd3d_amd.utils.Mnist2D import batch_from_images
    image_syn = init_synthetic(nclass, ipc, num_points=250, seed=5)
    images_all = torch.cat(reals, 0)
    per = reals[0].shape[0]
    d = Distiller(images_all, [list(range(c * per, (c + 1) * per)) for c in range(nclass)], net, ipc=ipc, num_classes=nclass,
                  batch_size=per, dist_opt=metric, device=DEV, image_syn=image_syn)
    batches = [batch_from_images(r.to(DEV), torch.full((per,), c, device=DEV), 250) for c, r in enumerate(reals)]
    return d, batches


@pytest.mark.parametrize("metric", ["modmse", "ours"])
@pytest.mark.parametrize("fused", [True, False])
def test_meconvexp_step_vs_reference_golden(metric, fused):
    """One matching step of MEConvExp: reference topology code (run over the ME restatement) vs the HIP path."""
    from dd3d_amd.models.MEConv import MEConvExp
    g = np.load(os.path.join(G, "g4_meconvexp.npz"))
    net = _load(MEConvExp, R.MEConvExp, False)
    net.fused = fused
    reals = [stroke_digits(4, seed=100 + c) for c in range(2)]
    d, batches = _distiller(net, reals, 2, metric, 2)
    loss = d.outer_step(real_batches=batches, apply_update=False)
    assert loss.item() == pytest.approx(float(g[metric + ".loss"]), rel=1e-4)
    np.testing.assert_allclose(torch.stack(d.last["per_class"]).cpu().numpy(), g[metric + ".per_class"], rtol=1e-4)
    assert rel_err(d.last["grad"], g[metric + ".grad"]) < 1e-4
    for c in range(2):
        assert rel_err(d.last["logits"][c][0], g[metric + ".logits_real"][c]) < 1e-4
        assert rel_err(d.last["logits"][c][1], g[metric + ".logits_syn"][c]) < 1e-4
        np.testing.assert_allclose([t.double().norm().item() for t in d.last["gw_real"][c]], g[metric + ".gwr_norm"][c], rtol=1e-4)
        np.testing.assert_allclose([t.double().norm().item() for t in d.last["gw_syn"][c]], g[metric + ".gws_norm"][c], rtol=1e-4)


@pytest.mark.parametrize("metric", ["modmse", "ours"])
def test_mepytorch_step_vs_reference_golden(metric):
    from dd3d_amd.configs import settings
    from dd3d_amd.models.MEConv import MEPytorch
    g = np.load(os.path.join(G, "g3_mepytorch.npz"))
    net = _load(MEPytorch, R.MEPytorch, True)
    reals = [stroke_digits(4, seed=100 + c) for c in range(2)]
    d, batches = _distiller(net, reals, 2, metric, 2)
    settings.Pyt = True
    try:
        loss = d.outer_step(real_batches=batches, apply_update=False)
    finally:
        settings.Pyt = False
    e_loss = abs(loss.item() - float(g[metric + ".loss"])) / abs(float(g[metric + ".loss"]))
    e_grad = rel_err(d.last["grad"], g[metric + ".grad"])
    print(f"MEPytorch {metric}: loss rel {e_loss:.2e}, xyz-grad rel {e_grad:.2e}")
    assert e_loss < 1e-4 and e_grad < 1e-4   # north_star tolerance
    for c in range(2):
        assert rel_err(d.last["logits"][c][0], g[metric + ".logits_real"][c]) < 1e-4


@pytest.mark.parametrize("metric", ["modmse", "ours", "cos"])
def test_meconvexp_step_vs_oracle_midsize(metric):
    """3 classes x 24 real clouds, ipc 3, fp64 oracle on the CPU vs the HIP path; then the SGD update itself."""
    from dd3d_amd.models.MEConv import MEConvExp
    net = _load(MEConvExp, R.MEConvExp, False)
    onet = R.MEConvExp(in_channel=3, out_channel=10)
    onet.load_state_dict(det_state_dict(onet, "net.MEConvExp"))
    onet = onet.double().train()
    reals = [stroke_digits(24, seed=200 + c) for c in range(3)]
    d, batches = _distiller(net, reals, 3, metric, 3)
    osyn = d.image_syn.detach().cpu().double().clone()
    R.clamp_synthetic_(osyn)
    osyn.requires_grad_()
    opt = torch.optim.SGD([osyn], lr=0.1, momentum=0.5)
    ref = R.matching_step(onet, osyn, reals, 3, metric, False, optimizer=opt)
    loss = d.outer_step(real_batches=batches, apply_update=True)
    assert loss.item() == pytest.approx(ref["loss"].item(), rel=1e-4)
    assert rel_err(d.last["grad"], ref["grad"]) < 1e-4
    assert rel_err(d.image_syn, osyn) < 1e-5
    for c in range(3):
        for a, b in zip(d.last["gw_real"][c], ref["gw_real"][c]):
            assert rel_err(a, b) < 1e-4
        for a, b in zip(d.last["gw_syn"][c], ref["gw_syn"][c]):
            assert rel_err(a, b) < 1e-4


def test_true_second_order_mode_vs_oracle():
    """The true-second-order research mode (NOT the reference's semantics): inside ``ops.true_second_order()``
    the matching loss is differentiated through every sparse op twice.  Same loss as the default mode (the forward is the same), the xyz
    gradient equals plain fp64 torch autograd over the restated ops (``oracle.minkowski_ref.true_second_order``) -- and differs from the
    MinkowskiEngine-faithful gradient, which is what makes the mode worth a name."""
    from oracle import minkowski_ref as ME
    from dd3d_amd import ops
    from dd3d_amd.models.MEConv import MEConvExp
    net = _load(MEConvExp, R.MEConvExp, False)
    onet = R.MEConvExp(in_channel=3, out_channel=10)
    onet.load_state_dict(det_state_dict(onet, "net.MEConvExp"))
    onet = onet.double().train()
    reals = [stroke_digits(6, seed=400 + c) for c in range(2)]
    res = {}
    for mode in ("me", "true"):
        d, batches = _distiller(net, reals, 2, "modmse", 2)
        osyn = d.image_syn.detach().cpu().double().clone()
        R.clamp_synthetic_(osyn)
        osyn.requires_grad_()
        if mode == "true":
            with ME.true_second_order():
                ref = R.matching_step(onet, osyn, reals, 2, "modmse", False)
            with ops.true_second_order():
                loss = d.outer_step(real_batches=batches, apply_update=False)
        else:
            ref = R.matching_step(onet, osyn, reals, 2, "modmse", False)
            loss = d.outer_step(real_batches=batches, apply_update=False)
        assert loss.item() == pytest.approx(ref["loss"].item(), rel=1e-4)
        assert rel_err(d.last["grad"], ref["grad"]) < 1e-4
        res[mode] = (loss.item(), d.last["grad"].clone())
    assert res["me"][0] == pytest.approx(res["true"][0], rel=1e-6)
    assert rel_err(res["me"][1], res["true"][1]) > 1e-3
    assert not ops.true_second_order_enabled()


def test_inner_network_update_vs_oracle_and_class_shards():
    """The inner-loop network update (no reference behaviour: mnist_distill.py:262 declares the loop and never runs it) follows the loop's
    per-class synthetic forward (:351-355).  Checked against the fp64 oracle restating the same thing, and: the parameter gradients of two
    class shards sum to the whole set's (what the all-reduce of sharding.all_reduce_gradients adds up over RCCL)."""
    from oracle import minkowski_ref as ME
    from dd3d_amd.mnist_distill import inner_network_update
    from dd3d_amd.models.MEConv import MEConvExp
    nclass, ipc = 4, 2
    net = _load(MEConvExp, R.MEConvExp, False)
    onet = R.MEConvExp(in_channel=3, out_channel=10)
    onet.load_state_dict(det_state_dict(onet, "net.MEConvExp"))
    onet = onet.double().train()
    image_syn = init_synthetic(nclass, ipc, num_points=250, seed=9)
    R.clamp_synthetic_(image_syn)                  # mnist_distill.py:311-317, as at the top of every iteration
    image_syn = image_syn.to(DEV)
    label_syn = torch.arange(nclass, device=DEV).repeat_interleave(ipc)
    osyn = image_syn.detach().cpu().double()
    oopt = torch.optim.SGD(onet.parameters(), lr=0.01, momentum=0.5)
    oopt.zero_grad()
    oloss = 0.0
    for c in range(nclass):
        items = [osyn[c * ipc + i] for i in range(ipc)]
        lab = torch.full((ipc,), c, dtype=torch.long)
        c_, f_, l_ = ME.sparse_collate(items, items, [lab[i].reshape(1) for i in range(ipc)], dtype=torch.float64)
        out = onet(R.create_input_batch({"coordinates": c_, "features": f_, "labels": l_}, True, device="cpu", quantization_size=1,
                                        coord_dtype=torch.float32))
        l = torch.nn.functional.cross_entropy(out, l_) / nclass
        l.backward()
        oloss += float(l.detach())
    ograd = [p.grad.clone() for p in onet.parameters()]
    oopt.step()
    # class shards first (lr 0: the weights stay), then the whole set with the real step
    shard = []
    for mine in ([0, 2], [1, 3]):
        opt0 = torch.optim.SGD(net.parameters(), lr=0.0)
        ls = inner_network_update(net, image_syn, label_syn, opt0, ipc, classes=mine, device=DEV)
        shard.append(([p.grad.clone() for p in net.parameters()], float(ls)))
    opt = torch.optim.SGD(net.parameters(), lr=0.01, momentum=0.5)
    loss = inner_network_update(net, image_syn, label_syn, opt, ipc, device=DEV)
    assert float(loss) == pytest.approx(oloss, rel=1e-4) and shard[0][1] + shard[1][1] == pytest.approx(float(loss), rel=1e-5)
    for p, g0, g1, og in zip(net.parameters(), shard[0][0], shard[1][0], ograd):
        assert rel_err(p.grad, og) < 1e-4
        assert rel_err(g0 + g1, p.grad) < 1e-5
    for p, op in zip(net.parameters(), onet.parameters()):
        assert rel_err(p.detach(), op.detach()) < 1e-5


def test_batch_from_images_equals_per_item_collate():
    from dd3d_amd.utils.Mnist2D import Mnist2Dreal, batch_from_images, get_mnist_dataloader
    g = np.load(os.path.join(G, "g2_plumbing.npz"))
    imgs = stroke_digits(4, seed=11)
    imgs[3] = 0
    imgs[3, 0, 5, 7] = 1
    labels = torch.tensor([3, 1, 4, 1])
    b = batch_from_images(imgs.to(DEV), labels.to(DEV), 250)
    np.testing.assert_array_equal(b["coordinates"].cpu().numpy(), g["real.coordinates"])
    np.testing.assert_array_equal(b["features"].cpu().numpy(), g["real.features"])
    np.testing.assert_array_equal(b["labels"].cpu().numpy(), g["real.labels"])
    b20 = batch_from_images(imgs.to(DEV), labels.to(DEV), 20)
    np.testing.assert_array_equal(b20["coordinates"].cpu().numpy(), g["real20.coordinates"])
    (pb,) = list(get_mnist_dataloader(Mnist2Dreal(imgs, labels, False), 4))
    np.testing.assert_array_equal(pb["coordinates"].numpy(), g["real.coordinates"])


def test_report_table2_anchor_over_1250_iterations():
    """Report Table 2's protocol on the GPU path (1 synthetic cloud per class, Kaiming init, 1250 iterations, MinkowskiEngine layers):
    mean |gradient| on the synthetic xyz (mnist_distill.py:367) against the report's 4e-4 (layer-wise cosine) and 0.033 (sum of squares).
    Stroke digits stand in for MNIST, so this is an anchor for the unpinned MinkowskiEngine semantics, not a parity bound; the factors were
    taken apart one at a time (tools/table2.py, tools/table2_decay.py):
      * SURVEY.md section 6's 0.039 (its own restatement, ipc 10, iteration 0) is REPRODUCED: 0.0396 -- asserted here to 5 %;
      * at the report's ipc 1 the same quantity is 0.55 at iteration 0 (x14: the cloud's gradient is no longer a mean over ten clouds) and
        decays to 0.15 averaged over 1250 iterations (0.08 at the end of the reference's 2500); Kaiming seed, synthetic seed, real batch
        size and stroke density move it by +-2 % / 2 % / 4 % / -13 % -- the remaining factor 4.6 to the report's 0.033 is not explained by
        anything that can be varied offline (MNIST itself cannot): asserted as a band around the measured factor, not a decade;
      * the layer-wise cosine sits 19 % above the report (6 % at MNIST-like stroke density): asserted within 40 %."""
    from dd3d_amd.fused_step import FusedDistiller
    from dd3d_amd.models.MEConv import MEConvExp
    from dd3d_amd.synthetic import mnist3d_like
    images, _, idx = mnist3d_like(10, 256, seed=0)
    got = {}
    for metric, report, lo, hi in (("ours", 4e-4, 0.9, 1.4), ("modmse", 0.033, 3.0, 6.5)):
        torch.manual_seed(0)
        net = MEConvExp(in_channel=3, out_channel=10).to(DEV).train()
        with FusedDistiller(images, idx, net, ipc=1, device=DEV, seed=1, image_syn=init_synthetic(10, 1, 250, seed=0), dist_opt=metric, graph=True) as f:
            acc = torch.zeros((), device=DEV, dtype=torch.float64)
            first = None
            for it in range(1250):
                f.outer_step()
                m = f.last["grad"].abs().mean()
                first = float(m) if it == 0 else first
                acc += m.double()
            f.check()
        got[metric] = (first, float(acc) / 1250)
        print(f"{metric}: mean |xyz-grad| iteration 0 {first:.3e}, average over 1250 iterations {got[metric][1]:.3e} (report {report}: x{got[metric][1] / report:.2f})")
        assert report * lo < got[metric][1] < report * hi
    assert got["ours"][0] == pytest.approx(4.92e-4, rel=0.05) and got["modmse"][0] == pytest.approx(0.552, rel=0.05)   # the oracle's iteration 0
    # "mean |xyz-grad| for sum-of-squares came out at 0.039" -- its configuration was ipc 10, first iteration
    torch.manual_seed(0)
    net = MEConvExp(in_channel=3, out_channel=10).to(DEV).train()
    with FusedDistiller(images, idx, net, ipc=10, device=DEV, seed=1, image_syn=init_synthetic(10, 10, 250, seed=0), dist_opt="modmse", graph=False) as f:
        f.outer_step()
        v = float(f.last["grad"].abs().mean())
    print(f"modmse, ipc 10, iteration 0: {v:.4f} (SURVEY.md section 6: 0.039)")
    assert v == pytest.approx(0.039, rel=0.05)
