"""ctypes front of oracle/cpu/libdd3d_cpu.so, the C++/OpenMP restatement of the matching step of MEConvExp.
TEST INFRASTRUCTURE AND CPU BASELINE ONLY (see oracle/__init__.py): tests/ use it as a second checker that shares no code with
oracle/minkowski_ref.py (which pins it: tests/test_cpu_leg.py), bench.py's ``cpu_baseline`` leg times it on the host cores.

``matching_step`` mirrors ``oracle.ref_model.matching_step`` (mnist_distill.py:333-365 for one iteration, no optimizer step)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cpu")
_LIB = os.path.join(_DIR, "libdd3d_cpu.so")
METRICS = {"ours": 0, "modmse": 1, "mse": 2, "cos": 3}


class _Net(C.Structure):
    _fields_ = [("J", C.c_int), ("E0", C.c_int), ("E1", C.c_int), ("E2", C.c_int)] + [(n, C.c_void_p) for n in ("W0", "w1", "b1", "W1", "w5", "b5", "Wc", "bc")]


def build():
    """Compile the library in-tree (g++ -O3 -fopenmp; seconds)."""
    subprocess.run(["make", "-C", _DIR], check=True, stdout=subprocess.DEVNULL)


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            build()
        _lib = C.CDLL(_LIB)
        _lib.dd3d_cpu_num_params.restype = C.c_int64
        _lib.dd3d_cpu_num_params.argtypes = [C.POINTER(_Net)]
        _lib.dd3d_cpu_set_threads.argtypes = [C.c_int]
        _lib.dd3d_cpu_max_threads.restype = C.c_int
        for name in ("dd3d_cpu_class_step_f64", "dd3d_cpu_class_step_f32"):
            f = getattr(_lib, name)
            f.restype = C.c_int
            f.argtypes = [C.POINTER(_Net)] + [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int] * 2 + [C.c_int, C.c_int] + [C.c_void_p] * 4
    return _lib


class Network:
    """The eight parameters of a MEConvExp (any object with the reference's ``parameters()`` order and a ``dense_extent``) as contiguous
    host arrays of one dtype."""

    def __init__(self, network, dtype=np.float64):
        self.dtype = np.dtype(dtype)
        ps = [p.detach().cpu().double().numpy() for p in network.parameters()]
        assert len(ps) == 8 and ps[0].shape == (27, 3, 128) and ps[3].shape == (27, 128, 128), "MEConvExp parameters expected (models/MEConv.py:221-230)"
        self.arrays = [np.ascontiguousarray(p.reshape(-1), dtype=self.dtype) for p in ps]
        self.shapes = [p.shape for p in ps]
        E = tuple(getattr(network, "dense_extent", (1, 28, 28)))
        self.J, self.S = int(ps[7].shape[0]), int(E[0] * E[1] * E[2])
        assert ps[6].shape == (self.J, 128 * self.S)
        self.c = _Net(self.J, int(E[0]), int(E[1]), int(E[2]), *[a.ctypes.data for a in self.arrays])
        self.n_params = int(lib().dd3d_cpu_num_params(C.byref(self.c)))

    def split(self, flat):
        out, o = [], 0
        for s in self.shapes:
            n = int(np.prod(s))
            out.append(flat[o:o + n].reshape(s))
            o += n
        return out


def _cloud_arrays(clouds, quantization_size, dtype, coord_dtype):
    """clouds: list of [n_i, 3] arrays -> (coords / q as float64 [N,3], feats [N,3] in ``dtype``, off int32 [B+1]).  The quotient is taken in
    ``coord_dtype`` (the reference divides fp32 coordinates in fp32, models/MEConv.py:262) and then held exactly in a double."""
    off = np.zeros(len(clouds) + 1, np.int32)
    off[1:] = np.cumsum([c.shape[0] for c in clouds])
    pts = np.concatenate(clouds, 0)
    q = (pts.astype(coord_dtype) / np.asarray(quantization_size, coord_dtype)).astype(np.float64)
    return np.ascontiguousarray(q), np.ascontiguousarray(pts, dtype=dtype), off


def image_clouds(images, num_points=250):
    """utils/Mnist2D.py:50-59: the on-pixels of [n,1,28,28] {0,1} images as (0, y, x) rows, first ``num_points`` in row-major order."""
    out = []
    for img in np.asarray(images):
        yx = np.argwhere(img[0] == 1)[:num_points]
        out.append(np.concatenate([np.zeros((yx.shape[0], 1)), yx], 1).astype(np.float64))
    return out


def class_step(net: Network, real_clouds, syn_clouds, label, dist_opt="modmse", quantization_size=1, coord_dtype=np.float32, want_lists=False):
    """One class of one matching step: (loss, d loss / d synthetic points [sum n_i, 3], gw_real, gw_syn) -- the lists only when asked for."""
    dt = net.dtype
    rc, rf, ro = _cloud_arrays(real_clouds, quantization_size, dt, coord_dtype)
    sc, sf, so = _cloud_arrays(syn_clouds, quantization_size, dt, coord_dtype)
    loss = np.zeros(1, dt)
    grad = np.zeros_like(sf)
    gwr = np.zeros(net.n_params, dt) if want_lists else None
    gws = np.zeros(net.n_params, dt) if want_lists else None
    fn = lib().dd3d_cpu_class_step_f64 if dt == np.float64 else lib().dd3d_cpu_class_step_f32
    p = lambda a: None if a is None else a.ctypes.data
    rc_ = fn(C.byref(net.c), p(rc), p(rf), p(ro), len(real_clouds), p(sc), p(sf), p(so), len(syn_clouds), int(label), METRICS[dist_opt],
             p(loss), p(grad), p(gwr), p(gws))
    if rc_ == -2:
        raise RuntimeError("a stride-4 voxel lies outside the dense extent (MinkowskiEngine's dense() raises here)")
    if rc_ == -3:
        raise ValueError("a coordinate does not fit this leg's 16-bit voxel keys (|floor(c)| >= 32000, or more than 65535 clouds)")
    if rc_ != 0:
        raise ValueError("dd3d_cpu_class_step: bad arguments")
    return float(loss[0]), grad, (net.split(gwr) if want_lists else None), (net.split(gws) if want_lists else None)


def matching_step(net: Network, image_syn, real_images_per_class, ipc, dist_opt="modmse", num_points=250, quantization_size=1, class_ids=None,
                  coord_dtype=np.float32, want_lists=False):
    """mnist_distill.py:333-365 for one iteration (same arguments as oracle.ref_model.matching_step; image_syn: [classes * ipc, points, 3]).
    Returns dict(loss, per_class, grad [like image_syn], gw_real, gw_syn)."""
    syn = np.asarray(image_syn, dtype=np.float64)
    grad = np.zeros_like(syn)
    per_class, gwr_all, gws_all = [], [], []
    for c, imgs in enumerate(real_images_per_class):
        cid = c if class_ids is None else int(class_ids[c])
        reals = image_clouds(imgs, num_points) if getattr(imgs, "ndim", 0) == 4 else [np.asarray(x, dtype=np.float64)[:num_points] for x in imgs]
        syns = [syn[c * ipc + i] for i in range(ipc)]
        loss, g, gwr, gws = class_step(net, reals, syns, cid, dist_opt, quantization_size, coord_dtype, want_lists)
        grad[c * ipc:(c + 1) * ipc] = g.reshape(ipc, syn.shape[1], 3)
        per_class.append(loss)
        gwr_all.append(gwr)
        gws_all.append(gws)
    return {"loss": float(np.sum(per_class)), "per_class": np.asarray(per_class), "grad": grad, "gw_real": gwr_all, "gw_syn": gws_all}
