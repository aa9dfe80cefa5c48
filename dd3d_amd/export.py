"""Checkpoint and point-cloud export of the reference loop.

* ``save_checkpoint`` writes the ``torch.save`` dictionary of ``mnist_distill.py:390-398`` -- keys ``epoch``, ``least_loss``,
  ``model_state_dict`` (MinkowskiEngine naming: ``features.N.kernel`` in ``[27,Cin,Cout]`` offset order, see models/MEConv.py),
  ``optimizer_state_dict``, ``synthetic_xyz``, ``synthetic_labels`` -- so a checkpoint of either code base loads in the other.
* ``save_cad`` / ``export_pcd`` / ``export_mnist`` write the ``.ply`` files of ``configs/settings.py:204-222`` and
  ``mnist_distill.py:26-63`` with the reference's file names.  open3d is not installable offline; the writer emits what
  ``o3d.io.write_point_cloud`` emits by default for a point cloud without colours or normals (binary little-endian, three ``double``
  properties x y z, comment "Created by Open3D").  **Parity unpinned** against open3d itself (absent here); the tests pin the header
  text and a write/read round trip.
* ``save_image`` / ``make_grid``: the image grid ``torchvision.utils.save_image(mnist_imgs, name, nrow=cad_per_class)`` writes at
  ``mnist_distill.py:181`` (torchvision is absent offline; PIL is used directly).
"""
from __future__ import annotations

import os

import numpy as np
import torch

from .configs import settings

_PLY_HEADER = ("ply\nformat binary_little_endian 1.0\ncomment Created by Open3D\nelement vertex %d\n"
               "property double x\nproperty double y\nproperty double z\nend_header\n")


def write_ply(path: str, points) -> None:
    pts = np.asarray(points, dtype="<f8")
    if pts.ndim != 2 or pts.shape[1] != 3:
        raise ValueError("write_ply expects [n,3] points, got %r" % (pts.shape,))
    pts = np.ascontiguousarray(pts)
    with open(path, "wb") as f:
        f.write((_PLY_HEADER % pts.shape[0]).encode("ascii"))
        f.write(pts.tobytes())


def read_ply(path: str) -> np.ndarray:
    """Reads the files ``write_ply`` (or open3d, for xyz-only clouds in either float width) writes."""
    with open(path, "rb") as f:
        raw = f.read()
    end = raw.index(b"end_header\n") + len(b"end_header\n")
    header = raw[:end].decode("ascii").splitlines()
    if header[0] != "ply" or not header[1].startswith("format binary_little_endian"):
        raise ValueError("unsupported ply flavour: " + " / ".join(header[:2]))
    n = next(int(h.split()[2]) for h in header if h.startswith("element vertex"))
    props = [h.split()[1] for h in header if h.startswith("property")]
    if len(props) != 3 or len(set(props)) != 1 or props[0] not in ("double", "float"):
        raise ValueError("expected three float/double vertex properties, got %r" % (props,))
    return np.frombuffer(raw, dtype="<f8" if props[0] == "double" else "<f4", count=3 * n, offset=end).reshape(n, 3).astype(np.float64)


def save_cad(cad_list: torch.Tensor, directory: str, iteration: int, normalize: bool = True) -> None:
    """One ``<directory>/<i>_<iteration>.ply`` per synthetic cloud (configs/settings.py:204-222)."""
    clouds = cad_list.clone().detach().cpu().numpy()
    for i, cad in enumerate(clouds):
        if normalize:
            cad = np.clip(cad, -1, 1)
        write_ply(os.path.join(directory, "%d_%d.ply" % (i, iteration)), cad)


def export_pcd(arr: torch.Tensor, c, export_cad_dir: str, iteration: int, custom_name: str = "orig") -> None:
    """The first cloud of ``arr`` as ``original-<name>-<c>_<iteration>.ply`` (mnist_distill.py:46-63)."""
    write_ply(os.path.join(export_cad_dir, "original-%s-%s_%d.ply" % (custom_name, c, iteration)), np.asarray(arr.detach().cpu()[0]))


def export_mnist(arr: torch.Tensor, c, export_cad_dir: str, iteration: int, custom_name: str = "orig") -> None:
    """The on-pixels (== 1) of one image as integer points (mnist_distill.py:26-44)."""
    pts = np.argwhere(arr.clone().detach().cpu().numpy() == 1)
    write_ply(os.path.join(export_cad_dir, "original-%s-%s_%d.ply" % (custom_name, c, iteration)), pts)


def save_checkpoint(savepath: str, iteration: int, least_loss: float, network, optimizer_img, image_syn: torch.Tensor, label_syn: torch.Tensor):
    state = {
        "epoch": iteration + 1,
        "least_loss": least_loss,
        "model_state_dict": network.state_dict(),
        "optimizer_state_dict": optimizer_img.state_dict(),
        "synthetic_xyz": image_syn.clone(),
        "synthetic_labels": label_syn.clone(),
    }
    torch.save(state, savepath)
    return state


def load_checkpoint(path: str, network=None, optimizer_img=None, map_location=None):
    """Loads a checkpoint of this package or of the reference (tensors only: ``weights_only=True``)."""
    state = torch.load(path, map_location=map_location, weights_only=True)
    if network is not None:
        network.load_state_dict(state["model_state_dict"])
    if optimizer_img is not None:
        optimizer_img.load_state_dict(state["optimizer_state_dict"])
    return state


def make_grid(images: torch.Tensor, nrow: int = 8, padding: int = 2, pad_value: float = 0.0) -> torch.Tensor:
    """[N,C,H,W] -> [3,gh,gw]: ``nrow`` images per row, ``padding`` pixels between and around them; one channel becomes three."""
    if images.shape[1] == 1:
        images = images.expand(-1, 3, -1, -1)
    n, c, h, w = images.shape
    per_row = min(nrow, n)
    rows = (n + per_row - 1) // per_row
    grid = images.new_full((c, rows * (h + padding) + padding, per_row * (w + padding) + padding), pad_value)
    for i in range(n):
        y, x = (i // per_row) * (h + padding) + padding, (i % per_row) * (w + padding) + padding
        grid[:, y:y + h, x:x + w] = images[i]
    return grid


def save_image(images: torch.Tensor, path: str, nrow: int = 8) -> None:
    from PIL import Image
    grid = make_grid(images.detach().float().cpu(), nrow=nrow)
    arr = grid.mul(255).add_(0.5).clamp_(0, 255).permute(1, 2, 0).to(torch.uint8).numpy()
    Image.fromarray(arr).save(path)
