"""CPU restatement of the reference's own Python on the hot path, built on oracle/minkowski_ref.py.
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Each function cites the reference lines it follows (paths relative to /root/reference).  These
restatements are pinned by tests/golden/*.npz, which were produced in the build container by running
the reference's files themselves (tests/golden/make_golden.py); the MinkowskiEngine layer underneath
is parity-unpinned (oracle/minkowski_ref.py header).
"""
from __future__ import annotations

from typing import Dict, List, Sequence

import torch
import torch.nn as nn

from . import minkowski_ref as ME

# ----------------------------------------------------------------------------- data plumbing


def real_item(image: torch.Tensor, label: torch.Tensor, num_points: int = 250) -> Dict[str, torch.Tensor]:
    """utils/Mnist2D.py:50-59 (pixel_val=False, fixed_pt=False): on-pixels of a [1,28,28] {0,1} image
    as float (0,y,x) rows, first ``num_points`` in row-major order; features == coordinates."""
    occ = torch.argwhere(image == 1)[:num_points].to(torch.float32)
    return {"coordinates": occ, "features": occ, "label": label.to(torch.int64)}


def syn_item(cloud: torch.Tensor, label: torch.Tensor) -> Dict[str, torch.Tensor]:
    """utils/Mnist2D.py:84-92: coordinates and features are the SAME (differentiable) tensor."""
    c = cloud.to(torch.float32)
    return {"coordinates": c, "features": c, "label": label.to(torch.int64)}


def collate(items: Sequence[Dict[str, torch.Tensor]]) -> Dict[str, torch.Tensor]:
    """utils/MinkowskiCollate.py:4-15."""
    c, f, l = ME.sparse_collate([d["coordinates"] for d in items], [d["features"] for d in items],
                                [d["label"].reshape(1) for d in items], dtype=torch.float32)
    return {"coordinates": c, "features": f, "labels": l}


def create_input_batch(batch, is_minknet=True, device="cpu", quantization_size=0.05, coord_dtype=None):
    """models/MEConv.py:260-269 (in-place division of the spatial columns, then a TensorField).
    coord_dtype (optional, for an fp64 run of this restatement): the dtype the division is carried out in.  The reference divides fp32
    coordinates in fp32; a point within rounding of a voxel face can land in the neighbouring voxel when the same quotient is taken in
    fp64 -- with torch.float32 here the integer voxel coordinates are the reference's while everything after them stays fp64."""
    assert is_minknet
    c = batch["coordinates"][:, 1:]
    if coord_dtype is not None and coord_dtype != c.dtype:
        batch["coordinates"][:, 1:] = (c.to(coord_dtype) / quantization_size).to(c.dtype)
    else:
        batch["coordinates"][:, 1:] = c / quantization_size
    return ME.TensorField(coordinates=batch["coordinates"], features=batch["features"], device=device)


# ----------------------------------------------------------------------------- models


class MEConvExp(ME.MinkowskiNetwork):
    """models/MEConv.py:208-258: (Conv k3 -> InstanceNorm(128) -> ReLU -> MaxPool k2 s2) x2, dense, Linear(100352,10)."""

    def __init__(self, in_channel, out_channel, embedding_channel=1024, dimension=3, full_minkowski=False, *, num_classes=10,
                 dense_extent=(1, 28, 28)):
        """``num_classes`` / ``dense_extent``: keyword-only openings of the reference's hard-coded MNIST geometry (Linear(100352,10),
        dense shape [B,128,1,28,28]) for the ModelNet/ShapeNet-shaped configs; the defaults are the reference's values."""
        super().__init__(dimension)
        self.full_minkowski = full_minkowski
        self.dense_extent = tuple(int(e) for e in dense_extent)
        layers: List[nn.Module] = []
        cin = in_channel
        for _ in range(2):
            layers += [ME.MinkowskiConvolution(cin, 128, kernel_size=3, dimension=dimension),
                       ME.MinkowskiInstanceNorm(128), ME.MinkowskiReLU(inplace=True),
                       ME.MinkowskiMaxPooling(kernel_size=2, stride=2, dimension=dimension)]
            cin = 128
        self.features = nn.Sequential(*layers)
        for m in self.modules():  # :232-242
            if isinstance(m, ME.MinkowskiConvolution):
                ME.kaiming_normal_(m.kernel, mode="fan_out", nonlinearity="relu")
        self.n_dense = 128 * self.dense_extent[0] * self.dense_extent[1] * self.dense_extent[2]
        self.classifier = nn.Linear(self.n_dense, num_classes)

    def forward(self, x: ME.TensorField):
        out = self.features(x.sparse())
        B = int(torch.max(x.C, dim=0).values[0].item()) + 1  # :250
        min_coord = out.C.min(0, keepdim=True)[0][:, 1:].cpu()  # :251-253, min over the WHOLE batch
        dense = out.dense(shape=torch.Size([B, 128, *self.dense_extent]), min_coordinate=min_coord)[0]
        return self.classifier(dense.view(-1, self.n_dense))


class MEPytorch(ME.MinkowskiNetwork):
    """models/MEConv.py:166-207 (net_depth 3): sparse() -> dense [B,3,1,28,28] -> [Conv3d(1,3,3),pad 1 -> ReLU ->
    MaxPool3d(2)] x3 -> Linear(1152,10)."""

    def __init__(self, in_channel, out_channel, dimension=3):
        super().__init__(dimension)
        layers: List[nn.Module] = []
        cin = in_channel
        for _ in range(3):
            layers += [nn.Conv3d(cin, 128, kernel_size=(1, 3, 3), padding=1), nn.ReLU(inplace=True),
                       nn.MaxPool3d(kernel_size=2, stride=2)]
            cin = 128
        self.pyt = nn.Sequential(*layers)
        self.classifier = nn.Linear(1152, 10)

    def forward(self, x: ME.TensorField):
        out = x.sparse()
        B = int(torch.max(x.C, dim=0).values[0].item()) + 1
        min_coord = out.C.min(0, keepdim=True)[0][:, 1:].cpu()
        dense = out.dense(shape=torch.Size([B, 3, 1, 28, 28]), min_coordinate=min_coord)[0]
        out = self.pyt(dense)
        return self.classifier(out.view(out.size(0), -1))


# ----------------------------------------------------------------------------- gradient distance


def _as_rows(t: torch.Tensor, pyt: bool) -> torch.Tensor:
    """Row grouping shared by distance_wb / distance_mse (distillation_loss.py:27-47 / :58-77)."""
    if t.ndim in (4, 5):
        return t.reshape(t.shape[0], -1)
    if t.ndim == 3:
        if not pyt:
            t = t.permute(2, 1, 0)
        return t.reshape(t.shape[0], -1)
    if t.ndim == 1:
        return t.reshape(1, -1)
    return t


def distance_wb(gwr, gws, pyt):
    """distillation_loss.py:26-55."""
    r, s = _as_rows(gwr, pyt), _as_rows(gws, pyt)
    return torch.sum(1 - torch.sum(r * s, dim=-1) / (torch.norm(r, dim=-1) * torch.norm(s, dim=-1) + 0.000001))


def distance_mse(gwr, gws, pyt):
    """distillation_loss.py:57-81."""
    r, s = _as_rows(gwr, pyt), _as_rows(gws, pyt)
    return torch.sum((r - s) ** 2)


def match_loss(gw_syn, gw_real, dis_metric, device="cpu", pyt=False):
    """distillation_loss.py:83-130 (``pyt`` stands for the global settings.Pyt read at :35,:66)."""
    if dis_metric == "ours":
        dis = torch.tensor(0.0, dtype=gw_real[0].dtype)
        for r, s in zip(gw_real, gw_syn):
            dis = dis + distance_wb(r, s, pyt)
    elif dis_metric == "modmse":
        dis = torch.tensor(0.0, dtype=gw_real[0].dtype)
        for r, s in zip(gw_real, gw_syn):
            dis = dis + distance_mse(r, s, pyt)
    elif dis_metric == "mse":
        r = torch.cat([g.reshape(-1) for g in gw_real])
        s = torch.cat([g.reshape(-1) for g in gw_syn])
        dis = torch.sum((s - r) ** 2)
    elif dis_metric == "cos":
        r = torch.cat([g.reshape(-1) for g in gw_real])
        s = torch.cat([g.reshape(-1) for g in gw_syn])
        dis = 1 - torch.sum(r * s, dim=-1) / (torch.norm(r, dim=-1) * torch.norm(s, dim=-1) + 0.000001)
    else:
        raise SystemExit("unknown distance function: %s" % dis_metric)
    return dis


# ----------------------------------------------------------------------------- one outer matching step


def clamp_synthetic_(image_syn: torch.Tensor):
    """mnist_distill.py:311-317."""
    with torch.no_grad():
        image_syn[image_syn < 0] = 0
        image_syn[image_syn > 27] = 27
        image_syn[:, :, 0] = 0


def matching_step(network, image_syn, real_images_per_class, ipc, dist_opt="modmse", pyt=False,
                  optimizer=None, num_points=250, quantization_size=1, class_ids=None, coord_dtype=None):
    """mnist_distill.py:333-369 for one iteration.

    real_images_per_class[c]: {0,1} float tensor [n_c,1,28,28] (the already-sampled real batch of class c), or -- for the
    point-cloud configs -- a list of [n_i,3] float clouds (coordinates = features = the points, as Mnist2Dreal does with pixels).
    image_syn: leaf tensor [num_classes*ipc, num_points, 3] with requires_grad=True.
    coord_dtype: see create_input_batch.
    class_ids (optional): the label of every entry of real_images_per_class (default: its position) -- a subset of the classes on its own.
    Returns dict(loss, per_class, gw_real, gw_syn (detached), grad) and applies optimizer.step() if given."""
    criterion = nn.CrossEntropyLoss()
    params = list(network.parameters())
    loss = torch.tensor(0.0, dtype=image_syn.dtype)
    per_class, gws_all, gwr_all, logits_all = [], [], [], []
    for c, imgs in enumerate(real_images_per_class):
        cid = c if class_ids is None else int(class_ids[c])
        labels = torch.full((len(imgs),), cid, dtype=torch.long)
        if torch.is_tensor(imgs):
            batch = collate([real_item(imgs[i], labels[i], num_points) for i in range(imgs.shape[0])])
        else:
            batch = collate([syn_item(imgs[i][:num_points], labels[i]) for i in range(len(imgs))])
        batch["coordinates"] = batch["coordinates"].to(image_syn.dtype)
        batch["features"] = batch["features"].to(image_syn.dtype)
        out = network(create_input_batch(batch, True, device="cpu", quantization_size=quantization_size, coord_dtype=coord_dtype))
        loss_real = criterion(out, batch["labels"])
        gw_real = [g.detach().clone() for g in torch.autograd.grad(loss_real, params)]
        syn = image_syn[c * ipc:(c + 1) * ipc]
        lab = torch.full((ipc,), cid, dtype=torch.long)
        items = [{"coordinates": syn[i], "features": syn[i], "label": lab[i]} for i in range(ipc)]
        c_, f_, l_ = ME.sparse_collate([d["coordinates"] for d in items], [d["features"] for d in items],
                                       [d["label"].reshape(1) for d in items], dtype=image_syn.dtype)
        sbatch = {"coordinates": c_, "features": f_, "labels": l_}
        out_s = network(create_input_batch(sbatch, True, device="cpu", quantization_size=quantization_size, coord_dtype=coord_dtype))
        loss_syn = criterion(out_s, sbatch["labels"])
        gw_syn = torch.autograd.grad(loss_syn, params, create_graph=True)
        m = match_loss(gw_syn, gw_real, dist_opt, "cpu", pyt)
        loss = loss + m
        per_class.append(m.detach())
        gws_all.append([g.detach() for g in gw_syn])
        gwr_all.append(gw_real)
        logits_all.append((out.detach(), out_s.detach()))
    if optimizer is not None:
        optimizer.zero_grad()
    grad = None
    if loss.requires_grad:
        (grad,) = torch.autograd.grad(loss, [image_syn], allow_unused=True)
    if optimizer is not None and grad is not None:
        image_syn.grad = grad.clone()
        optimizer.step()
    return {"loss": loss.detach(), "per_class": torch.stack(per_class), "gw_real": gwr_all, "gw_syn": gws_all,
            "grad": grad, "logits": logits_all}


# ------------------------------------------------------------------------------------------------------------------
# Evaluation block
# ------------------------------------------------------------------------------------------------------------------
def populate_img(occ_grid: torch.Tensor, img_array: torch.Tensor) -> torch.Tensor:
    """mnist_distill.py:163-168: every point switches on pixel (int(p[1]), int(p[2])) of its cloud's image; the conversion truncates
    toward zero and the assignment follows Python indexing (negative = from the end, out of range = IndexError).  Pinned by
    tests/golden/g6_eval.npz (the reference's own function on seeded clouds)."""
    h, w = img_array.shape[2], img_array.shape[3]
    for idx, cloud in enumerate(occ_grid.detach().cpu().numpy()):
        for p in cloud:
            y, x = int(p[1]), int(p[2])
            if not (-h <= y < h and -w <= x < w):
                raise IndexError("index out of range")
            img_array[idx, 0, y, x] = 1
    return img_array
