"""Generate the golden fixtures in tests/golden/ by running the REFERENCE'S OWN files.

Runs only in the build container (needs /root/reference, which never travels).  It imports the reference's
``distillation_loss.py``, ``models/MEConv.py``, ``utils/Mnist2D.py`` and ``utils/MinkowskiCollate.py`` unchanged,
with in-memory stand-ins for the modules that are absent offline:

* ``configs.settings``  -> a namespace carrying only ``Pyt``, ``num_points``, ``device`` (configs/settings.py:137,183,167)
* ``MinkowskiEngine``   -> oracle/minkowski_ref.py (MinkowskiEngine 0.5.4 is not installable offline)
* ``torchvision``       -> name-only module (utils/Mnist2D.py:2,27 need ``datasets.MNIST`` as a base class only)

Fixtures (data only; inputs are regenerated from tests/detgen.py seeds by the tests):
  g1_match_loss.npz   reference match_loss / distance_* values and grads, 4 metrics x Pyt in {F,T}
  g2_plumbing.npz     Mnist2Dreal / Mnist2Dsyn / minkowski_collate_fn / create_input_batch outputs
  g3_mepytorch.npz    reference MEPytorch: one matching step (2 classes x 4 real, ipc 2)
  g4_meconvexp.npz    reference MEConvExp topology over the ME restatement: one matching step, modmse + ours
  g5_ctor.npz         constructor parameter names / shapes / order
  g7_unused_models.npz  the unused MEConv / MEConvImage: names / shapes; one forward + backward of MEConvImage(full_minkowski=True)
  g6_eval.npz         evaluation block: reference models/MNIST.py ConvNet (names, shapes, seeded forward) and
                      mnist_distill.py populate_img / augment / epoch on seeded inputs (open3d, torchinfo, torchvision.utils are
                      name-only stand-ins: none of them is reached by these functions)

usage:  python tests/golden/make_golden.py
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, REPO)

from oracle import minkowski_ref as MEref  # noqa: E402
from tests.detgen import det_state_dict, det_tensor, eval_inputs, sample_idx  # noqa: E402
from dd3d_amd.synthetic import init_synthetic, stroke_digits  # noqa: E402


def install_stubs():
    settings = types.ModuleType("configs.settings")
    settings.Pyt = False
    settings.num_points = 250
    settings.device = torch.device("cpu")
    configs = types.ModuleType("configs")
    configs.__path__ = []
    configs.settings = settings
    sys.modules["configs"] = configs
    sys.modules["configs.settings"] = settings

    me = types.ModuleType("MinkowskiEngine")
    me.__path__ = []
    for n in ("MinkowskiNetwork", "TensorField", "SparseTensor", "MinkowskiConvolution", "MinkowskiInstanceNorm",
              "MinkowskiReLU", "MinkowskiMaxPooling", "MinkowskiAvgPooling", "MinkowskiGlobalAvgPooling",
              "MinkowskiLinear", "MinkowskiBatchNorm"):
        setattr(me, n, getattr(MEref, n))
    utils = types.ModuleType("MinkowskiEngine.utils")
    utils.sparse_collate = MEref.sparse_collate
    utils.kaiming_normal_ = MEref.kaiming_normal_
    me.utils = utils
    ops = types.ModuleType("MinkowskiEngine.MinkowskiOps")
    ops.to_sparse = ops.MinkowskiToSparseTensor = ops.MinkowskiToDenseTensor = object
    me.MinkowskiOps = ops
    sys.modules["MinkowskiEngine"] = me
    sys.modules["MinkowskiEngine.utils"] = utils
    sys.modules["MinkowskiEngine.MinkowskiOps"] = ops

    tv = types.ModuleType("torchvision")
    tv.__path__ = []
    ds = types.ModuleType("torchvision.datasets")
    ds.MNIST = object
    tr = types.ModuleType("torchvision.transforms")
    tv.datasets, tv.transforms = ds, tr
    sys.modules["torchvision"] = tv
    sys.modules["torchvision.datasets"] = ds
    sys.modules["torchvision.transforms"] = tr
    return settings


MECONVEXP_SHAPES = [(27, 3, 128), (1, 128), (1, 128), (27, 128, 128), (1, 128), (1, 128), (10, 100352), (10,)]
MEPYTORCH_SHAPES = [(128, 3, 1, 3, 3), (128,), (128, 128, 1, 3, 3), (128,), (128, 128, 1, 3, 3), (128,), (10, 1152), (10,)]


def grad_lists(shapes, tag):
    real = [det_tensor(s, f"{tag}.real.{i}", 0.05) for i, s in enumerate(shapes)]
    syn = [det_tensor(s, f"{tag}.syn.{i}", 0.05).requires_grad_() for i, s in enumerate(shapes)]
    return syn, real


def g1(settings, dl):
    out = {}
    for fam, shapes in (("exp", MECONVEXP_SHAPES), ("pyt", MEPYTORCH_SHAPES)):
        for pyt in (False, True):
            settings.Pyt = pyt
            for metric in ("ours", "modmse", "mse", "cos"):
                syn, real = grad_lists(shapes, "g1." + fam)
                v = dl.match_loss(syn, real, metric, "cpu")
                grads = torch.autograd.grad(v, syn)
                key = f"{fam}.{int(pyt)}.{metric}"
                out[key + ".value"] = np.float64(v.item())
                out[key + ".gnorm"] = np.array([g.double().norm().item() for g in grads])
                out[key + ".gsample"] = np.stack([
                    np.pad(g.reshape(-1)[sample_idx(g.numel(), f"{fam}.{i}")].numpy().astype(np.float64),
                           (0, 64 - min(64, g.numel()))) for i, g in enumerate(grads)])
            # the per-layer helpers as well
            syn, real = grad_lists(shapes, "g1." + fam)
            out[f"{fam}.{int(pyt)}.wb"] = np.array([dl.distance_wb(r, s).item() for r, s in zip(real, syn)])
            out[f"{fam}.{int(pyt)}.mse"] = np.array([dl.distance_mse(r, s).item() for r, s in zip(real, syn)])
    settings.Pyt = False
    np.savez_compressed(os.path.join(HERE, "g1_match_loss.npz"), **out)


def g2(settings, Mnist2D, MEConv):
    imgs = stroke_digits(4, seed=11)
    imgs[3] = 0
    imgs[3, 0, 5, 7] = 1  # a single-pixel image
    labels = torch.tensor([3, 1, 4, 1])
    ds = Mnist2D.Mnist2Dreal(imgs, labels, False)
    (batch,) = list(Mnist2D.get_mnist_dataloader(ds, 4))
    out = {"real.coordinates": batch["coordinates"].numpy(), "real.features": batch["features"].numpy(),
           "real.labels": batch["labels"].numpy()}
    settings.num_points = 20  # the cap
    (batch20,) = list(Mnist2D.get_mnist_dataloader(Mnist2D.Mnist2Dreal(imgs, labels, False), 4))
    settings.num_points = 250
    out["real20.coordinates"] = batch20["coordinates"].numpy()
    syn = init_synthetic(2, 2, num_points=5, seed=3)
    (sb,) = list(Mnist2D.get_mnist_dataloader(Mnist2D.Mnist2Dsyn(syn, torch.tensor([0, 0, 1, 1]), False), 4))
    out["syn.coordinates"] = sb["coordinates"].detach().numpy()
    out["syn.features"] = sb["features"].detach().numpy()
    out["syn.labels"] = sb["labels"].numpy()
    tf = MEConv.create_input_batch({k: v.clone() for k, v in batch.items()}, True, device="cpu", quantization_size=0.5)
    out["field.C.q05"] = tf.C.numpy()
    np.savez_compressed(os.path.join(HERE, "g2_plumbing.npz"), **out)


def reference_step(settings, Mnist2D, MEConv, dl, network, image_syn, reals, ipc, metric):
    """mnist_distill.py:333-365 with the reference's own callables."""
    criterion = torch.nn.CrossEntropyLoss()
    params = list(network.parameters())
    loss = torch.tensor(0.0)
    rec = {"per_class": [], "logits_real": [], "logits_syn": [], "gwr_norm": [], "gws_norm": []}
    for c, imgs in enumerate(reals):
        labels = torch.ones((imgs.shape[0],), dtype=torch.long) * c
        loss_real = torch.tensor(0.0)
        for batch in Mnist2D.get_mnist_dataloader(Mnist2D.Mnist2Dreal(imgs, labels, False), imgs.shape[0]):
            out = network(MEConv.create_input_batch(batch, True, device="cpu", quantization_size=1))
            loss_real = loss_real + criterion(out, batch["labels"])
        gw_real = [g.detach().clone() for g in torch.autograd.grad(loss_real, params)]
        img_syn = image_syn[c * ipc:(c + 1) * ipc]
        lab_syn = torch.ones((ipc,), dtype=torch.long) * c
        for batch in Mnist2D.get_mnist_dataloader(Mnist2D.Mnist2Dsyn(img_syn, lab_syn, False), ipc):
            out_s = network(MEConv.create_input_batch(batch, True, device="cpu", quantization_size=1))
            loss_syn = criterion(out_s, batch["labels"])
            gw_syn = torch.autograd.grad(loss_syn, params, create_graph=True)
        m = dl.match_loss(gw_syn, gw_real, metric, "cpu")
        loss = loss + m
        rec["per_class"].append(m.item())
        rec["logits_real"].append(out.detach().numpy())
        rec["logits_syn"].append(out_s.detach().numpy())
        rec["gwr_norm"].append([g.double().norm().item() for g in gw_real])
        rec["gws_norm"].append([g.double().norm().item() for g in gw_syn])
    loss.backward()
    rec["loss"] = loss.item()
    rec["grad"] = image_syn.grad.detach().numpy().copy()
    return rec


def run_model_fixture(settings, Mnist2D, MEConv, dl, cls, pyt, fname, metrics):
    settings.Pyt = pyt
    torch.manual_seed(0)
    net = cls(3, 10) if pyt else cls(in_channel=3, out_channel=10)
    net.load_state_dict(det_state_dict(net, "net." + cls.__name__))
    net.train()
    reals = [stroke_digits(4, seed=100 + c) for c in range(2)]
    out = {}
    for metric in metrics:
        image_syn = init_synthetic(2, 2, num_points=250, seed=5)
        image_syn[image_syn < 0] = 0      # the iteration prologue, mnist_distill.py:311-317
        image_syn[image_syn > 27] = 27
        image_syn[:, :, 0] = 0
        image_syn.requires_grad_()
        rec = reference_step(settings, Mnist2D, MEConv, dl, net, image_syn, reals, 2, metric)
        for k, v in rec.items():
            out[f"{metric}.{k}"] = np.asarray(v)
    settings.Pyt = False
    np.savez_compressed(os.path.join(HERE, fname), **out)


def g5(MEConv):
    out = {}
    for name, net in (("MEConvExp", MEConv.MEConvExp(in_channel=3, out_channel=10)), ("MEPytorch", MEConv.MEPytorch(3, 10))):
        names = [n for n, _ in net.named_parameters()]
        out[name + ".names"] = np.array(names)
        out[name + ".shapes"] = np.array([str(tuple(p.shape)) for p in net.parameters()])
        out[name + ".sd_keys"] = np.array(list(net.state_dict().keys()))
    net = MEConv.MEConvExp(in_channel=3, out_channel=10)
    out["MEConvExp.kernel_std"] = np.array([net.features[0].kernel.std().item(), net.features[4].kernel.std().item()])
    np.savez_compressed(os.path.join(HERE, "g5_ctor.npz"), **out)


def g6(settings):
    for name in ("open3d", "torchinfo"):
        m = types.ModuleType(name)
        m.summary = lambda *a, **k: None
        sys.modules[name] = m
    tvu = types.ModuleType("torchvision.utils")
    tvu.save_image = lambda *a, **k: None
    tvu.make_grid = lambda *a, **k: None
    sys.modules["torchvision.utils"] = tvu
    sys.modules["torchvision"].utils = tvu
    import mnist_distill as md   # the reference's loop module; its __main__ block does not run
    from models.MNIST import ConvNet
    out = {}
    torch.manual_seed(0)
    net = ConvNet(1, 10, 128, 3, "relu", "instancenorm", "avgpooling")
    out["convnet.names"] = np.array([n for n, _ in net.named_parameters()])
    out["convnet.shapes"] = np.array([str(tuple(p.shape)) for p in net.parameters()])
    out["convnet.sd_keys"] = np.array(list(net.state_dict().keys()))
    out["convnet.repr"] = np.array(repr(net))
    clouds, imgs, labels = eval_inputs()
    net.eval()
    with torch.no_grad():
        out["convnet.out"] = net(imgs).numpy()
    # populate_img (mnist_distill.py:163-168)
    canvas = torch.zeros(clouds.shape[0], 1, 28, 28)
    out["populate.img"] = md.populate_img(clouds, canvas).numpy()
    # augment (mnist_distill.py:65-124): one of crop / scale / rotate per image, numpy global RNG
    np.random.seed(5)
    torch.manual_seed(5)
    out["augment.out"] = md.augment(imgs.clone(), torch.device("cpu")).numpy()
    # epoch (mnist_distill.py:126-158): three training epochs + one test epoch, SGD as evaluate_synset :176
    settings.device = torch.device("cpu")
    np.random.seed(7)
    torch.manual_seed(7)
    net = ConvNet(1, 10, 128, 3, "relu", "instancenorm", "avgpooling")
    opt = torch.optim.SGD(net.parameters(), lr=0.01, momentum=0.9, weight_decay=0.0005)
    crit = torch.nn.CrossEntropyLoss()
    from utils.Mnist2D import TensorDataset
    loader = torch.utils.data.DataLoader(TensorDataset(imgs.clone(), labels), batch_size=256, shuffle=True, num_workers=0)
    rec = [md.epoch("train", loader, net, opt, crit, aug=False) for _ in range(3)]
    rec.append(md.epoch("test", loader, net, opt, crit, aug=False))
    out["epoch.loss_acc"] = np.array(rec, dtype=np.float64)
    out["epoch.w_norms"] = np.array([p.detach().norm().item() for p in net.parameters()])
    np.savez_compressed(os.path.join(HERE, "g6_eval.npz"), **out)


def g7(settings, Mnist2D, MEConv):
    """The reference's UNUSED models (models/MEConv.py:10-164) over the ME restatement: constructor names / shapes of MEConv and
    MEConvImage in both modes, and one forward + backward of the only configuration that runs, MEConvImage(full_minkowski=True)."""
    out = {}
    for name, cls in (("MEConv", MEConv.MEConv), ("MEConvImage", MEConv.MEConvImage)):
        for full in (False, True):
            net = cls(in_channel=3, out_channel=10, full_minkowski=full)
            tag = f"{name}.{'full' if full else 'dense'}"
            out[tag + ".names"] = np.array([n for n, _ in net.named_parameters()])
            out[tag + ".shapes"] = np.array([str(tuple(p.shape)) for p in net.parameters()])
            out[tag + ".sd_keys"] = np.array(list(net.state_dict().keys()))
    torch.manual_seed(0)
    net = MEConv.MEConvImage(in_channel=3, out_channel=10, full_minkowski=True)
    out["MEConvImage.kernel_std"] = np.array([net.features[i].kernel.std().item() for i in (0, 4, 8)])
    net.load_state_dict(det_state_dict(net, "net.MEConvImage"))
    net.train()
    imgs = torch.cat([stroke_digits(3, seed=300 + c) for c in range(2)])
    labels = torch.tensor([0, 0, 0, 1, 1, 1])
    for batch in Mnist2D.get_mnist_dataloader(Mnist2D.Mnist2Dreal(imgs, labels, False), imgs.shape[0]):
        tf = MEConv.create_input_batch(batch, True, device="cpu", quantization_size=1)
        logits = net(tf)
        loss = torch.nn.CrossEntropyLoss()(logits, batch["labels"])
    grads = torch.autograd.grad(loss, list(net.parameters()))
    out["fwd.logits"] = logits.detach().numpy()
    out["fwd.loss"] = np.array(loss.item())
    out["fwd.grad_norm"] = np.array([g.double().norm().item() for g in grads])
    out["fwd.grad_kernel0"] = grads[0].detach().numpy()
    np.savez_compressed(os.path.join(HERE, "g7_unused_models.npz"), **out)


def main():
    settings = install_stubs()
    sys.path.insert(0, REF)
    import distillation_loss as dl
    from models import MEConv
    from utils import Mnist2D

    if "--only-g7" in sys.argv:
        g7(settings, Mnist2D, MEConv)
        print("g7 written to", HERE)
        return
    g1(settings, dl)
    g2(settings, Mnist2D, MEConv)
    run_model_fixture(settings, Mnist2D, MEConv, dl, MEConv.MEPytorch, True, "g3_mepytorch.npz", ["modmse", "ours"])
    run_model_fixture(settings, Mnist2D, MEConv, dl, MEConv.MEConvExp, False, "g4_meconvexp.npz", ["modmse", "ours"])
    g5(MEConv)
    g6(settings)
    g7(settings, Mnist2D, MEConv)
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
