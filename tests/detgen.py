"""Deterministic, seed-addressed tensors shared by the golden-vector generator and the tests, so fixtures do not
have to carry multi-megabyte weight blobs."""
import zlib

import numpy as np
import torch


def det_tensor(shape, tag: str, scale: float = 1.0, dtype=torch.float32) -> torch.Tensor:
    rng = np.random.default_rng(zlib.crc32(tag.encode()))
    a = rng.standard_normal(int(np.prod(shape)) if len(shape) else 1).astype(np.float64) * scale
    return torch.from_numpy(a.reshape(tuple(shape))).to(dtype)


def det_state_dict(module: torch.nn.Module, tag: str):
    """A state dict with the module's own names/shapes; per-parameter scale ~ 1/sqrt(fan) so activations stay O(1)."""
    sd = {}
    for name, p in module.state_dict().items():
        shape = tuple(p.shape)
        if name.endswith("kernel"):            # [K, Cin, Cout]
            scale = (2.0 / (shape[2] * shape[0])) ** 0.5
        elif p.ndim >= 2 and "classifier" in name:
            scale = 1.0 / shape[1] ** 0.5
        elif p.ndim == 5:                        # Conv3d [Cout, Cin, 1, 3, 3]
            scale = 1.0 / (shape[1] * 9) ** 0.5
        elif name.endswith("weight") and p.ndim == 2 and shape[0] == 1:   # instance-norm weight [1,C]
            sd[name] = (1.0 + det_tensor(shape, tag + name, 0.1)).to(p.dtype)
            continue
        else:
            scale = 0.1
        sd[name] = det_tensor(shape, tag + name, scale).to(p.dtype)
    return sd


def sample_idx(numel: int, tag: str, k: int = 64) -> np.ndarray:
    rng = np.random.default_rng(zlib.crc32(("idx" + tag).encode()))
    return rng.integers(0, numel, size=min(k, numel))


def eval_inputs():
    """Seeded inputs of the evaluation-block fixture g6_eval.npz: 6 clouds of 40 points in [0, 28), 12 binary 28x28 images, labels."""
    clouds = det_tensor((6, 40, 3), "g6.clouds").abs() * 9.0
    clouds = clouds.clamp(0, 27.99)
    clouds[:, :, 0] = 0
    imgs = (det_tensor((12, 1, 28, 28), "g6.imgs") > 0.8).float()
    labels = torch.arange(12) % 10
    return clouds, imgs, labels
