"""Drop-in for the hot-path part of the reference's ``utils/Mnist2D.py``: Mnist2Dreal (:43-76), Mnist2Dsyn (:78-102),
get_mnist_dataloader (:105-107).  ``get_dataset`` (torchvision MNIST download) is dataset preparation, outside the hot path; the
benchmark feeds MNIST-shaped synthetic images (dd3d_amd/synthetic.py)."""
import torch
from torch.utils.data import DataLoader, Dataset

from ..configs import settings
from .MinkowskiCollate import minkowski_collate_fn


class Mnist2Dreal(Dataset):
    def __init__(self, images_tensor, labels_tensor, pixel_val=False, fixed_pt=False):
        self.images, self.labels = images_tensor, labels_tensor
        self.pixel_val, self.fixed_pt = pixel_val, fixed_pt

    def __getitem__(self, index):
        image, label = self.images[index], self.labels[index]
        if self.pixel_val:
            return {"coordinates": torch.argwhere(image).to(torch.float32),
                    "features": torch.flatten(image).unsqueeze(1).to(torch.float32), "label": label.to(torch.int64)}
        pts = image if self.fixed_pt else torch.argwhere(image == 1)[:settings.num_points]
        pts = pts.to(torch.float32)
        return {"coordinates": pts, "features": pts, "label": label.to(torch.int64)}

    def __len__(self):
        return self.images.shape[0]


class Mnist2Dsyn(Dataset):
    def __init__(self, images_tensor, labels_tensor, pixel_val=False):
        self.images, self.labels, self.pixel_val = images_tensor, labels_tensor, pixel_val

    def __getitem__(self, index):
        image, label = self.images[index], self.labels[index]
        if self.pixel_val:
            return {"coordinates": torch.argwhere(image).to(torch.float32), "features": image.reshape(-1, 1).to(torch.float32),
                    "label": label.to(torch.int64)}
        pts = image.to(torch.float32)  # coordinates and features are the SAME differentiable tensor
        return {"coordinates": pts, "features": pts, "label": label.to(torch.int64)}

    def __len__(self):
        return self.images.shape[0]


def get_mnist_dataloader(dataset, batch_size):
    return DataLoader(dataset=dataset, batch_size=batch_size, shuffle=False, collate_fn=minkowski_collate_fn)


class TensorDataset(Dataset):
    def __init__(self, images, labels):
        self.images, self.labels = images.detach().float(), labels.detach()

    def __getitem__(self, index):
        return self.images[index], self.labels[index]

    def __len__(self):
        return self.images.shape[0]


def batch_from_images(images, labels, num_points=None):
    """The batch dict that ``get_mnist_dataloader(Mnist2Dreal(images, labels), len(images))`` yields (utils/Mnist2D.py:50-59 +
    utils/MinkowskiCollate.py:4-15), built on the device with one ``nonzero`` instead of one ``argwhere`` per image:
    rows (b, 0, y, x) in image-major, row-major order, at most ``num_points`` per image; features = the spatial columns."""
    if num_points is None:
        num_points = settings.num_points
    B = images.shape[0]
    nz = torch.nonzero(images == 1)                      # [n, 4] = (b, channel, y, x), lexicographic
    b = nz[:, 0]
    counts = torch.bincount(b, minlength=B)
    start = torch.cumsum(counts, 0) - counts
    rank = torch.arange(nz.shape[0], device=nz.device) - start[b]
    nz = nz[rank < num_points]
    coords = nz.to(torch.float32)
    kept = counts.clamp(max=num_points)
    pt_off = torch.cat([kept.new_zeros(1), torch.cumsum(kept, 0)]).to(torch.int32)   # first point of every cloud (+ total)
    return {"coordinates": coords, "features": coords[:, 1:].clone(), "labels": labels.to(torch.int64).reshape(-1),
            "num_clouds": B, "pt_off": pt_off}
