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
        P, I, L = C.c_void_p, C.c_int, C.c_int64
        sig = {"dd3d_cpu_quantize_unique": [P, P, I, P, P, P, P], "dd3d_cpu_stride_coords": [P, L, I, I, P, P, P, P], "dd3d_cpu_kernel_map": [P, L, I, P],
               "dd3d_cpu_pool_map": [P, L, P, L, I, P], "dd3d_cpu_dense_sites": [P, L, I, I, I, I, P]}
        for sfx in ("f64", "f32"):
            sig.update({f"dd3d_cpu_voxel_mean_fwd_{sfx}": [P, L, I, P, L, P], f"dd3d_cpu_voxel_mean_bwd_{sfx}": [P, L, I, P, L, P],
                        f"dd3d_cpu_conv_fwd_{sfx}": [P, I, P, I, P, L, L, P], f"dd3d_cpu_conv_bwd_data_{sfx}": [P, I, P, I, P, L, L, P],
                        f"dd3d_cpu_conv_bwd_weight_{sfx}": [P, I, P, I, P, L, L, P], f"dd3d_cpu_inorm_fwd_{sfx}": [P, P, I, I, P, P],
                        f"dd3d_cpu_inorm_bwd_{sfx}": [P, P, P, P, I, I, P], f"dd3d_cpu_affine_relu_pool_fwd_{sfx}": [P, P, P, I, P, L, L, P, P],
                        f"dd3d_cpu_affine_relu_pool_bwd_{sfx}": [P, P, P, P, P, P, I, L, P, P],
                        f"dd3d_cpu_sparse_linear_fwd_{sfx}": [P, P, L, P, I, I, I, P, P, I, P], f"dd3d_cpu_sparse_linear_bwd_x_{sfx}": [P, P, L, P, I, I, I, P, I, P],
                        f"dd3d_cpu_sparse_linear_bwd_w_{sfx}": [P, P, P, L, P, I, I, I, I, P], f"dd3d_cpu_softmax_ce_{sfx}": [P, P, I, I, P, P],
                        f"dd3d_cpu_match_loss_{sfx}": [P, P, I, I, I, P, P]})
        for name, args in sig.items():
            f = getattr(_lib, name)
            f.restype, f.argtypes = C.c_int, args
    return _lib


OP_NAMES = ("quantize_unique", "stride_coords", "kernel_map", "pool_map", "dense_sites")
FEATURE_OP_NAMES = ("voxel_mean_fwd", "voxel_mean_bwd", "conv_fwd", "conv_bwd_data", "conv_bwd_weight", "inorm_fwd", "inorm_bwd", "affine_relu_pool_fwd",
                    "affine_relu_pool_bwd", "sparse_linear_fwd", "sparse_linear_bwd_x", "sparse_linear_bwd_w", "softmax_ce", "match_loss")


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


# ---------------------------------------------------------------------------------------------------------------- op by op
# numpy fronts of the op-level entry points (oracle/cpu/dd3d_cpu.h): the array layouts of include/dd3d.h, so a test hands the arrays of a HIP
# call over (``.cpu().numpy()``) and compares -- maps bit for bit, features to rounding.

def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _f(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


def _p(a):
    return None if a is None else a.ctypes.data


def _check(rc, name):
    if rc == -2:
        raise RuntimeError(f"{name}: a voxel lies outside the dense extent")
    if rc == -3:
        raise ValueError(f"{name}: a coordinate does not fit this leg's 16-bit voxel keys")
    if rc != 0:
        raise ValueError(f"{name}: bad arguments")


def _op(name, dt, *args):
    full = f"dd3d_cpu_{name}" + ("" if dt is None else ("_f64" if np.dtype(dt) == np.float64 else "_f32"))
    _check(getattr(lib(), full)(*args), full)


def quantize_unique(coords, pt_off):
    """coords [N,3] (already divided by the voxel size), pt_off [B+1] -> (C [M,4], inv [N], seg_off [B+1]): TensorField.sparse()'s voxel rows in
    first-occurrence order."""
    coords, pt_off = _f(coords, np.float64), _i32(pt_off)
    N, B = coords.shape[0], pt_off.shape[0] - 1
    Cq, inv, seg, M = np.zeros((max(N, 1), 4), np.int32), np.zeros(max(N, 1), np.int32), np.zeros(B + 1, np.int32), np.zeros(1, np.int64)
    _op("quantize_unique", None, _p(coords), _p(pt_off), B, _p(Cq), _p(inv), _p(seg), _p(M))
    return Cq[:int(M[0])].copy(), inv[:N], seg


def stride_coords(Cq, B, stride):
    """-> (C_out [M_out,4], parent [M_in], seg_off [B+1]): floor(c / stride) * stride, unique in first-occurrence order."""
    Cq = _i32(Cq)
    M = Cq.shape[0]
    out, par, seg, Mo = np.zeros((max(M, 1), 4), np.int32), np.zeros(max(M, 1), np.int32), np.zeros(B + 1, np.int32), np.zeros(1, np.int64)
    _op("stride_coords", None, _p(Cq), M, B, stride, _p(out), _p(par), _p(seg), _p(Mo))
    return out[:int(Mo[0])].copy(), par[:M], seg


def kernel_map(Cq, tensor_stride):
    Cq = _i32(Cq)
    nbr = np.zeros((27, Cq.shape[0]), np.int32)
    _op("kernel_map", None, _p(Cq), Cq.shape[0], tensor_stride, _p(nbr))
    return nbr


def pool_map(C_in, C_out, tensor_stride_in):
    C_in, C_out = _i32(C_in), _i32(C_out)
    kids = np.zeros((8, C_out.shape[0]), np.int32)
    _op("pool_map", None, _p(C_in), C_in.shape[0], _p(C_out), C_out.shape[0], tensor_stride_in, _p(kids))
    return kids


def dense_sites(Cq, tensor_stride, extent):
    Cq = _i32(Cq)
    site = np.zeros(Cq.shape[0], np.int32)
    _op("dense_sites", None, _p(Cq), Cq.shape[0], tensor_stride, int(extent[0]), int(extent[1]), int(extent[2]), _p(site))
    return site


def voxel_mean_fwd(feats, inv, M, dt=np.float64):
    feats, inv = _f(feats, dt), _i32(inv)
    F0 = np.zeros((M, feats.shape[1]), dt)
    _op("voxel_mean_fwd", dt, _p(feats), feats.shape[0], feats.shape[1], _p(inv), M, _p(F0))
    return F0


def voxel_mean_bwd(dF0, inv, dt=np.float64):
    dF0, inv = _f(dF0, dt), _i32(inv)
    out = np.zeros((inv.shape[0], dF0.shape[1]), dt)
    _op("voxel_mean_bwd", dt, _p(dF0), dF0.shape[0], dF0.shape[1], _p(inv), inv.shape[0], _p(out))
    return out


def conv_fwd(X, W, nbr, M=None, dt=np.float64):
    """X [M,Cin], W [27,Cin,Cout], nbr [27,ld] -> Y [M,Cout] (stride-1 convolution: input and output rows are the same voxels)."""
    X, W, nbr = _f(X, dt), _f(W, dt), _i32(nbr)
    M = X.shape[0] if M is None else M
    Y = np.zeros((M, W.shape[2]), dt)
    _op("conv_fwd", dt, _p(X), W.shape[1], _p(W), W.shape[2], _p(nbr), nbr.shape[1], M, _p(Y))
    return Y


def conv_bwd_data(dY, W, nbr, M=None, dt=np.float64):
    dY, W, nbr = _f(dY, dt), _f(W, dt), _i32(nbr)
    M = dY.shape[0] if M is None else M
    dX = np.zeros((M, W.shape[1]), dt)
    _op("conv_bwd_data", dt, _p(dY), W.shape[2], _p(W), W.shape[1], _p(nbr), nbr.shape[1], M, _p(dX))
    return dX


def conv_bwd_weight(X, dY, nbr, M=None, dt=np.float64):
    X, dY, nbr = _f(X, dt), _f(dY, dt), _i32(nbr)
    M = dY.shape[0] if M is None else M
    dW = np.zeros((27, X.shape[1], dY.shape[1]), dt)
    _op("conv_bwd_weight", dt, _p(X), X.shape[1], _p(dY), dY.shape[1], _p(nbr), nbr.shape[1], M, _p(dW))
    return dW


def inorm_fwd(Y, seg_off, dt=np.float64):
    Y, seg = _f(Y, dt), _i32(seg_off)
    B = seg.shape[0] - 1
    xhat, inv_std = np.zeros_like(Y), np.zeros((B, Y.shape[1]), dt)
    _op("inorm_fwd", dt, _p(Y), _p(seg), B, Y.shape[1], _p(xhat), _p(inv_std))
    return xhat, inv_std


def inorm_bwd(g, xhat, inv_std, seg_off, dt=np.float64):
    g, xhat, inv_std, seg = _f(g, dt), _f(xhat, dt), _f(inv_std, dt), _i32(seg_off)
    dY = np.zeros_like(g)
    _op("inorm_bwd", dt, _p(g), _p(xhat), _p(inv_std), _p(seg), seg.shape[0] - 1, g.shape[1], _p(dY))
    return dY


def affine_relu_pool_fwd(xhat, w, b, kids, M_out=None, dt=np.float64):
    """max over the existing children of relu(xhat * w + b) -> (pooled [M_out,C], arg [M_out,C] = winning child row)."""
    xhat, w, b, kids = _f(xhat, dt), _f(w, dt).reshape(-1), _f(b, dt).reshape(-1), _i32(kids)
    M_out = kids.shape[1] if M_out is None else M_out
    out, arg = np.zeros((M_out, xhat.shape[1]), dt), np.zeros((M_out, xhat.shape[1]), np.int32)
    _op("affine_relu_pool_fwd", dt, _p(xhat), _p(w), _p(b), xhat.shape[1], _p(kids), kids.shape[1], M_out, _p(out), _p(arg))
    return out, arg


def affine_relu_pool_bwd(g_pool, arg, parent, xhat, w, b, dt=np.float64):
    """-> (ga [M_in,C] = the ReLU-masked pooled gradient at every input row, g_xhat = ga * w)."""
    g_pool, arg, parent, xhat = _f(g_pool, dt), _i32(arg), _i32(parent), _f(xhat, dt)
    w, b = _f(w, dt).reshape(-1), _f(b, dt).reshape(-1)
    ga, gx = np.zeros_like(xhat), np.zeros_like(xhat)
    _op("affine_relu_pool_bwd", dt, _p(g_pool), _p(arg), _p(parent), _p(xhat), _p(w), _p(b), xhat.shape[1], xhat.shape[0], _p(ga), _p(gx))
    return ga, gx


def sparse_linear_fwd(F, site, seg_off, S, Wc, bias, dt=np.float64):
    """dense(F).view(B, -1) @ Wc.T + bias without the dense tensor (models/MEConv.py:250-256): Wc [J, C*S], column c*S + site."""
    F, site, seg, Wc = _f(F, dt), _i32(site), _i32(seg_off), _f(Wc, dt)
    bias = None if bias is None else _f(bias, dt)
    B, J = seg.shape[0] - 1, Wc.shape[0]
    out = np.zeros((B, J), dt)
    _op("sparse_linear_fwd", dt, _p(F), _p(site), F.shape[0], _p(seg), B, F.shape[1], S, _p(Wc), _p(bias), J, _p(out))
    return out


def sparse_linear_bwd_x(dl, site, seg_off, S, Wc, C_=128, dt=np.float64):
    dl, site, seg, Wc = _f(dl, dt), _i32(site), _i32(seg_off), _f(Wc, dt)
    dF = np.zeros((site.shape[0], C_), dt)
    _op("sparse_linear_bwd_x", dt, _p(dl), _p(site), site.shape[0], _p(seg), seg.shape[0] - 1, C_, S, _p(Wc), Wc.shape[0], _p(dF))
    return dF


def sparse_linear_bwd_w(dl, F, site, seg_off, S, dt=np.float64):
    dl, F, site, seg = _f(dl, dt), _f(F, dt), _i32(site), _i32(seg_off)
    J = dl.shape[1]
    gW = np.zeros((J, F.shape[1] * S), dt)
    _op("sparse_linear_bwd_w", dt, _p(dl), _p(F), _p(site), F.shape[0], _p(seg), seg.shape[0] - 1, F.shape[1], S, J, _p(gW))
    return gW


def softmax_ce(logits, labels, dt=np.float64):
    """mean cross entropy and its gradient with respect to the logits."""
    logits, labels = _f(logits, dt), _i32(labels)
    loss, dl = np.zeros(1, dt), np.zeros_like(logits)
    _op("softmax_ce", dt, _p(logits), _p(labels), logits.shape[0], logits.shape[1], _p(loss), _p(dl))
    return float(loss[0]), dl


def match_loss(gw_syn, gw_real, dist_opt, dt=np.float64):
    """distillation_loss.match_loss over the eight MEConvExp-shaped tensors (lists in parameters() order) -> (loss, [d loss / d gw_syn[i]])."""
    shapes = [np.shape(t) for t in gw_syn]
    J, S = int(shapes[7][0]), int(shapes[6][1]) // 128
    fs, fr = (np.concatenate([_f(t, dt).reshape(-1) for t in l]) for l in (gw_syn, gw_real))
    loss, d = np.zeros(1, dt), np.zeros_like(fs)
    _op("match_loss", dt, _p(fs), _p(fr), J, S, METRICS[dist_opt], _p(loss), _p(d))
    out, o = [], 0
    for sh in shapes:
        n = int(np.prod(sh))
        out.append(d[o:o + n].reshape(sh))
        o += n
    return float(loss[0]), out
