"""The gradient-matching loop of the reference's ``mnist_distill.py`` (:294-374) on the MI355X kernels.

``Distiller.outer_step()`` is one iteration of the reference loop without evaluation / logging / export (SURVEY.md section 8a
row a1; those live in ``evaluation.py`` / ``export.py`` and are driven by ``main``): clamp (:311-317), per-class real and synthetic forward + gradients (:335-359), ``match_loss`` (:360), backward into the
synthetic xyz (:365) and the SGD(momentum 0.5) step (:369).  Run as a script it takes the reference's flags
(``--log/--no-log --exp --pytorch``) and INI keys, on MNIST-shaped synthetic digits (no dataset download offline).
"""
from __future__ import annotations

import time

import numpy as np
import torch
import torch.nn as nn

from .configs import settings
from .distillation_loss import match_loss
from .models.MEConv import MEConvExp, MEPytorch, create_input_batch
from .utils.Mnist2D import Mnist2Dsyn, batch_from_images, get_mnist_dataloader


class Distiller:
    def __init__(self, images_all, indices_class, network, ipc, num_classes=10, num_points=250, batch_size=256, dist_opt="modmse",
                 dist_lr=0.1, device=None, seed=0, image_syn=None, classes=None):
        self.device = torch.device(device or settings.device)
        self.images_all = images_all.to(self.device)
        self.indices_class = [np.asarray(ix) for ix in indices_class]
        self.network = network.to(self.device).train()
        self.net_parameters = list(self.network.parameters())
        self.num_classes, self.ipc, self.num_points, self.batch_size = num_classes, ipc, num_points, batch_size
        self.dist_opt, self.rng = dist_opt, np.random.RandomState(seed)
        # classes owned by this process (class sharding): every class is independent inside a step
        self.classes = list(range(num_classes)) if classes is None else list(classes)
        settings.num_points, settings.cad_per_class = num_points, ipc
        if image_syn is None:  # :295-299
            g = torch.Generator().manual_seed(seed)
            image_syn = 28 * torch.rand((num_classes * ipc, num_points, 3), generator=g, dtype=torch.float)
        self.image_syn = image_syn.to(self.device).clone()
        self.label_syn = torch.arange(num_classes, device=self.device).repeat_interleave(ipc)
        self.criterion = nn.CrossEntropyLoss().to(self.device)
        self.optimizer_img = torch.optim.SGD([self.image_syn], lr=dist_lr, momentum=0.5)  # :303
        self.last = {}

    # -- mnist_distill.py:311-317
    def clamp(self):
        s = self.image_syn
        s.requires_grad_(False)
        s[s < 0] = 0
        s[s > 27] = 27
        s[:, :, 0] = 0
        s.requires_grad_()

    # -- mnist_distill.py:200-208 (the dataset + dataloader + collate chain collapsed into one device-side gather)
    def get_images(self, c, n):
        idx = self.rng.permutation(self.indices_class[c])[:n]
        img_real = self.images_all[torch.as_tensor(idx, device=self.device)]
        labels = torch.full((img_real.shape[0],), c, device=self.device, dtype=torch.long)
        return batch_from_images(img_real, labels, self.num_points)

    # -- mnist_distill.py:241-249
    def synthetic_batch(self, c):
        img_syn = self.image_syn[c * self.ipc:(c + 1) * self.ipc]
        lab_syn = torch.full((self.ipc,), c, device=self.device, dtype=torch.long)
        (batch,) = list(get_mnist_dataloader(Mnist2Dsyn(img_syn, lab_syn, False), self.ipc))
        return batch

    def class_loss(self, c, real_batch=None):
        """mnist_distill.py:336-360 for one class; returns (matched_loss, gw_real, gw_syn)."""
        net, dev = self.network, self.device
        batch = real_batch if real_batch is not None else self.get_images(c, self.batch_size)
        output = net(create_input_batch(batch, True, device=dev, quantization_size=1))
        loss_real = self.criterion(output, batch["labels"])
        gw_real = [g.detach().clone() for g in torch.autograd.grad(loss_real, self.net_parameters)]
        sbatch = self.synthetic_batch(c)
        output_s = net(create_input_batch(sbatch, True, device=dev, quantization_size=1))
        loss_syn = self.criterion(output_s, sbatch["labels"])
        gw_syn = torch.autograd.grad(loss_syn, self.net_parameters, create_graph=True)
        self.last.setdefault("logits", []).append((output.detach(), output_s.detach()))
        return match_loss(gw_syn, gw_real, self.dist_opt, dev), gw_real, gw_syn

    def outer_step(self, real_batches=None, apply_update=True):
        """One iteration of mnist_distill.py:309-374 (no evaluation / logging / export).  Returns the summed matching loss (0-d,
        on the device, no host sync)."""
        self.clamp()
        self.last = {"per_class": [], "gw_real": [], "gw_syn": []}
        loss = torch.zeros((), device=self.device)
        for c in self.classes:
            m, gwr, gws = self.class_loss(c, None if real_batches is None else real_batches[c])
            loss = loss + m
            self.last["per_class"].append(m.detach())
            self.last["gw_real"].append(gwr)
            self.last["gw_syn"].append([g.detach() for g in gws])
        self.optimizer_img.zero_grad()
        if loss.requires_grad:
            loss.backward()
        self.last["grad"] = self.image_syn.grad.detach().clone() if self.image_syn.grad is not None else None
        if apply_update:
            self.optimizer_img.step()
        return loss.detach()


def inner_network_update(network, image_syn, label_syn, optimizer_net, ipc, classes=None, device=None, quantization_size=1, group=None):
    """One inner-loop update of the NETWORK on the synthetic set -- the other half of gradient matching (Zhao et al.), which the reference
    declares (``outer_loop, inner_loop = 1, 1``, mnist_distill.py:262) and never runs: its network stays at its initialisation, so there is
    no reference behaviour to match here.  What it does follows the loop's own synthetic forward (mnist_distill.py:351-355): one batch per
    class -- that class's ``ipc`` synthetic clouds through ``create_input_batch`` and the network, placed by that batch's own minimum corner
    (MEConv.py:251-254) -- mean cross entropy per class, averaged over all classes, then ``optimizer_net.step()``.
    Class-sharded (``classes`` = the classes this process owns; default all): every rank sums its own classes' terms, and the parameter
    gradients are summed over the ranks (``sharding.all_reduce_gradients``: one 5.83 MB all-reduce over RCCL) before the step, so every
    rank takes the step a single process would take -- exactly, since a class's batch never spans ranks.
    A FusedDistiller notices the changed weights by itself (parameter version counters) and re-captures its graph.  Returns this rank's
    share of the loss (detached)."""
    from .sharding import all_reduce_gradients
    from .utils.MinkowskiCollate import minkowski_collate_fn
    dev = torch.device(device or settings.device)
    num_classes = int(image_syn.shape[0]) // ipc
    classes = list(range(num_classes)) if classes is None else list(classes)
    optimizer_net.zero_grad()
    total = None
    for c in classes:
        syn = image_syn.detach()[c * ipc:(c + 1) * ipc]
        lab = label_syn[c * ipc:(c + 1) * ipc]
        batch = minkowski_collate_fn([{"coordinates": syn[i], "features": syn[i], "label": lab[i].reshape(1)} for i in range(ipc)])
        out = network(create_input_batch(batch, True, device=dev, quantization_size=quantization_size))
        loss = nn.functional.cross_entropy(out, batch["labels"].to(out.device).long()) / num_classes
        loss.backward()
        total = loss.detach() if total is None else total + loss.detach()
    all_reduce_gradients(network.parameters(), group)
    optimizer_net.step()
    return total


def main(argv=None, ini_path="configs/default.ini"):
    """The reference's ``__main__`` (mnist_distill.py:251-410) on MNIST-shaped synthetic digits: matching steps, the evaluation block at
    the reference's iterations (:276, :320-330), checkpoint + .ply export every ``save_cad_every`` iterations (:383-402).
    MEConvExp runs on the class-batched HIP-graph step (``FusedDistiller``, the path ``bench.py`` times); ``--pytorch`` (MEPytorch) keeps the
    autograd loop.  Two optional INI keys beyond the reference's, for short runs: [DISTILLATION] eval_epochs (1000) and num_eval (5)."""
    import copy
    import math
    import os

    from . import evaluation, export
    from .models.MNIST import ConvNet
    from .synthetic import mnist3d_like
    from .utils.Mnist2D import TensorDataset
    settings.init(argv, ini_path)
    dev = settings.device
    torch.random.manual_seed(int(time.time() * 1000) % 100000)
    images_all, _labels, indices_class = mnist3d_like(10, 512, seed=0)
    test_images, test_labels, _ = mnist3d_like(10, 100, seed=1)  # held-out digits standing in for the MNIST test split
    testloader = torch.utils.data.DataLoader(TensorDataset(test_images.reshape(-1, 1, 28, 28), test_labels), batch_size=256, shuffle=False)
    network = MEPytorch(3, 10) if settings.Pyt else MEConvExp(in_channel=3, out_channel=10)
    settings.log_string(network)
    from .fused_step import FusedDistiller
    make = Distiller if settings.Pyt else FusedDistiller
    d = make(images_all, indices_class, network, ipc=settings.cad_per_class, num_points=settings.num_points,
             batch_size=settings.modelconfig.getint("batch_size"), dist_opt=settings.modelconfig.get("dist_opt"),
             dist_lr=settings.modelconfig.getfloat("dist_lr"), device=dev)
    eval_epochs = settings.distillationconfig.getint("eval_epochs", fallback=1000)
    num_eval = settings.distillationconfig.getint("num_eval", fallback=5)
    total = settings.distillationconfig.getint("total_iterations")
    eval_iteration_pool = [0, 250, total // 4, total // 2, total // 4 + total // 2, total - 1]
    export_cad_dir = os.path.join(settings.logging_folder_name, settings.distillationconfig.get("export_folder_name"))
    os.makedirs(export_cad_dir, exist_ok=True)
    ckpt_dir = settings.distillationconfig.get("distillation_checkpoint_dir")
    os.makedirs(ckpt_dir, exist_ok=True)
    log_loss, least_loss, loss_value = [], math.inf, math.inf
    for iteration in range(total):
        if iteration in eval_iteration_pool:
            d.clamp()
            settings.log_string("----- Evaluation -----\nModel_train = Convnet2D\nIteration: " + str(iteration))
            for it_eval in range(num_eval):
                classifier_network = ConvNet(1, 10, 128, 3, "relu", "instancenorm", "avgpooling")
                evaluation.evaluate_synset(it_eval, classifier_network, copy.deepcopy(d.image_syn.detach()), copy.deepcopy(d.label_syn),
                                           testloader, iteration, Epoch=eval_epochs, export_cad_dir=export_cad_dir)
        loss = d.outer_step()
        log_loss.append(loss.item() / 10)
        if iteration % 100 == 0:
            loss_value = sum(log_loss) / len(log_loss)
            settings.log_string("Iteration: %d Loss matched is %s" % (iteration, loss_value))
            log_loss = []
        if iteration % settings.distillationconfig.getint("save_cad_every") == 0:
            if loss_value <= least_loss:
                least_loss = loss_value
                savepath = os.path.join(ckpt_dir, settings.exp_file_name + ".pth")
                settings.log_string("Saving at %s" % savepath)
                export.save_checkpoint(savepath, iteration, least_loss, network, d.optimizer_img, d.image_syn, d.label_syn)
            settings.log_string("Exporting Point Cloud")
            settings.save_cad(d.image_syn, export_cad_dir, iteration, normalize=False)
    return d


if __name__ == "__main__":
    main()
