"""Seeded synthetic MNIST-3D-shaped inputs: there is no network for datasets, so the
benchmark and the tests use stroke digits of the same shape as thresholded MNIST (mnist_distill.py:281-299,
utils/Mnist2D.py:21 ThresholdTransform)."""
from __future__ import annotations

import numpy as np
import torch

_STEPS8 = np.array([(-1, -1), (-1, 0), (-1, 1), (0, -1), (0, 1), (1, -1), (1, 0), (1, 1)], dtype=np.int64)


def stroke_digits(n: int, seed: int = 0, size: int = 28, steps: int = 90) -> torch.Tensor:
    """n binary images [n,1,size,size] in {0,1}: an 8-connected random walk of ``steps`` steps drawn with a 2x2
    brush (about 82 on-pixels each, always fewer than the 250-point cap of settings.num_points)."""
    rng = np.random.default_rng(seed)
    img = np.zeros((n, 1, size, size), dtype=np.float32)
    for i in range(n):
        y, x = rng.integers(4, size - 5, size=2)
        moves = _STEPS8[rng.integers(0, 8, size=steps)]
        for dy, dx in moves:
            y = int(min(max(y + dy, 0), size - 2))
            x = int(min(max(x + dx, 0), size - 2))
            img[i, 0, y:y + 2, x:x + 2] = 1.0
    return torch.from_numpy(img)


def mnist3d_like(num_classes: int = 10, per_class: int = 256, seed: int = 0):
    """images_all [num_classes*per_class,1,28,28] and indices_class (mnist_distill.py:280-286 shapes)."""
    images = stroke_digits(num_classes * per_class, seed)
    indices_class = [list(range(c * per_class, (c + 1) * per_class)) for c in range(num_classes)]
    labels = torch.arange(num_classes).repeat_interleave(per_class)
    return images, labels, indices_class


def init_synthetic(num_classes: int, ipc: int, num_points: int = 250, seed: int = 0, extent: float = 28.0):
    """image_syn = 28 * rand([num_classes*ipc, num_points, 3]) (mnist_distill.py:295-299), seeded."""
    g = torch.Generator().manual_seed(seed)
    return extent * torch.rand((num_classes * ipc, num_points, 3), generator=g, dtype=torch.float32)


def shell_clouds(n: int, pts: int, seed: int = 0) -> torch.Tensor:
    """n clouds of ``pts`` points on random ellipsoid shells in [-1,1]^3 (ModelNet40 / ShapeNet-shaped configs)."""
    rng = np.random.default_rng(seed)
    v = rng.standard_normal((n, pts, 3))
    v /= np.linalg.norm(v, axis=2, keepdims=True) + 1e-12
    radii = rng.uniform(0.3, 0.95, size=(n, 1, 3))
    return torch.from_numpy((v * radii).astype(np.float32))
