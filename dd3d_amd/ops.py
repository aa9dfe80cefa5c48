"""torch.autograd wrappers around libdd3d_hip.so -- one Function per MinkowskiEngine op family on the hot path.

Second-order semantics.  The reference differentiates ``torch.autograd.grad(loss_syn, params,
create_graph=True)`` again (mnist_distill.py:359,365).  MinkowskiEngine's Functions have native backward passes, so what comes out
of them under ``create_graph=True`` is a constant; only the torch-native pieces (instance-norm affine, ReLU, dense()/Linear/CE) are
differentiated twice.  The Functions below reproduce exactly that: the backward of every sparse op calls a HIP kernel on raw
buffers (its results carry no graph), while

  * ``NormActPoolFn`` keeps the one second-order path that exists inside a block,
    d(sum_rows ga * xhat)/dxhat, by routing its backward through ``_NormActPoolBwdFn`` (a Function with its own backward);
  * the classifier (``SparseLinearFn`` and friends) is closed under differentiation, like index_put + addmm in the reference.
"""
from __future__ import annotations

from typing import Optional

import torch
from torch.autograd import Function

from . import _lib as L
from .sparse import CoordinateManager

_F32 = torch.float32

# ----------------------------------------------------------------------------------------------- second-order mode
# "me" (default): MinkowskiEngine's semantics, above.  "true" (an explicitly separate research mode): every
# op between the loss and the weights is differentiated twice -- the convolutions through SparseConvTrueFn (its backward is a
# Function with a backward of its own, all three on the HIP kernels), instance norm / ReLU / pooling through differentiable torch ops
# on the device (the pooling winners still come from the HIP kernel).  Only the op-by-op path of the models honours it; the fused step
# (fused_step.FusedDistiller) is the reference's semantics and nothing else.
_TRUE_SECOND_ORDER = False


class true_second_order:
    """``with ops.true_second_order(): ...`` -- networks run inside differentiate the sparse ops twice (see above)."""

    def __init__(self, enabled: bool = True):
        self.enabled = bool(enabled)

    def __enter__(self):
        global _TRUE_SECOND_ORDER
        self.prev, _TRUE_SECOND_ORDER = _TRUE_SECOND_ORDER, self.enabled
        return self

    def __exit__(self, *exc):
        global _TRUE_SECOND_ORDER
        _TRUE_SECOND_ORDER = self.prev


def true_second_order_enabled() -> bool:
    return _TRUE_SECOND_ORDER


def _c(t: torch.Tensor) -> torch.Tensor:
    t = t.detach()
    if t.dtype != _F32:
        raise RuntimeError("dd3d_amd kernels compute in fp32")
    return t if t.is_contiguous() else t.contiguous()


# ----------------------------------------------------------------------------------------------- voxel mean
class VoxelMeanFn(Function):
    """F0[v] = mean of the features of voxel v's points (MinkowskiSPMMAverageFunction inside TensorField.sparse())."""

    @staticmethod
    def forward(ctx, feats, cm: CoordinateManager):
        M = cm.num_rows(1)
        f = _c(feats)
        N, Cf = f.shape
        out = torch.empty((M, Cf), dtype=_F32, device=f.device)
        L.call("dd3d_voxel_mean_fwd", L.ptr(f), N, Cf, L.ptr(cm.vox_off), L.ptr(cm.vox_pts), M, None, L.ptr(out), L.stream())
        ctx.cm, ctx.N, ctx.Cf = cm, N, Cf
        return out

    @staticmethod
    def backward(ctx, g):
        cm = ctx.cm
        g = _c(g)
        d = torch.empty((ctx.N, ctx.Cf), dtype=_F32, device=g.device)
        L.call("dd3d_voxel_mean_bwd", L.ptr(g), g.shape[0], ctx.Cf, L.ptr(cm.levels[1].inv), L.ptr(cm.vox_off), ctx.N, L.ptr(d), L.stream())
        return d, None


def voxel_mean(feats, cm):
    return VoxelMeanFn.apply(feats, cm)


# ----------------------------------------------------------------------------------------------- convolution
def transposed_kernel(W: torch.Tensor) -> torch.Tensor:
    """Wt[k][co][ci] for the backward-data pass; cached ON the weight tensor and invalidated by its version counter."""
    hit = getattr(W, "_dd3d_wt", None)
    if hit is not None and hit[0] == W._version and hit[1].device == W.device:
        return hit[1]
    K, Cin, Cout = W.shape
    Wt = torch.empty((K, Cout, Cin), dtype=_F32, device=W.device)
    L.call("dd3d_transpose_kernel", L.ptr(_c(W)), K, Cin, Cout, L.ptr(Wt), L.stream())
    try:
        W._dd3d_wt = (W._version, Wt)
    except Exception:
        pass
    return Wt


_group_cache = {}


def _single_group(M: int, device) -> torch.Tensor:
    """group_row_off of a single group [0, M) (cached: building a device tensor from host values is a blocking copy)."""
    key = (int(M), str(device))
    t = _group_cache.get(key)
    if t is None:
        if len(_group_cache) > 4096:
            _group_cache.clear()
        t = _group_cache[key] = torch.tensor([0, M], dtype=torch.int32, device=device)
    return t


def conv_bwd_weight(X, dY, nbr, ld, M_out, group_row_off, G):
    K, Cin, Cout = nbr.shape[0], X.shape[1], dY.shape[1]
    dW = torch.empty((G, K, Cin, Cout), dtype=_F32, device=X.device)
    ws = L.workspace(int(L.lib.dd3d_conv_bwd_weight_ws_bytes(M_out, G, K, Cin, Cout)), X.device)
    L.call("dd3d_conv_bwd_weight", L.ptr(X), Cin, L.ptr(dY), Cout, L.ptr(nbr), ld, K, M_out, L.ptr(group_row_off), G, L.ptr(dW),
           L.ptr(ws), ws.numel(), L.stream())
    return dW


def _hot_shape(K, Cin, Cout) -> bool:
    """The shapes of MEConvExp (models/MEConv.py:224): their own kernels (conv.hip, conv_bww_sk.hip); everything else: conv_any.hip."""
    return K == 27 and Cout == 128 and Cin in (3, 128)


def conv_fwd_raw(x, w, nbr, ld, M_out):
    """Y[o] = sum_k x[nbr[k][o]] @ w[k] on raw (detached, contiguous fp32) buffers; nbr None: kernel_size 1 (every row its own neighbour)."""
    K, Cin, Cout = w.shape
    Y = torch.empty((M_out, Cout), dtype=_F32, device=x.device)
    if _hot_shape(K, Cin, Cout):
        L.call("dd3d_conv_fwd", L.ptr(x), x.shape[0], Cin, L.ptr(w), Cout, L.ptr(nbr), ld, K, M_out, None, 0, L.ptr(Y), L.stream())
    else:
        L.call("dd3d_conv_any_fwd", L.ptr(x), x.shape[0], Cin, L.ptr(w), Cout, L.ptr(nbr), ld, K, M_out, None, L.ptr(Y), L.stream())
    return Y


def conv_bwd_data_raw(g, w, nbr, ld, W_cache=None):
    """dX[i] = sum over the pairs (k, o) with nbr[k][o] == i of g[o] @ w[k]^T (stride-1 convolution: input rows == output rows).
    ``W_cache``: the tensor to hang the transposed kernel on (a parameter keeps it until its version counter moves)."""
    K, Cin, Cout = w.shape
    M = g.shape[0]
    dX = torch.empty((M, Cin), dtype=_F32, device=g.device)
    if _hot_shape(K, Cin, Cout):
        wk = transposed_kernel(W_cache if W_cache is not None else w) if Cin == 128 else w
        L.call("dd3d_conv_bwd_data", L.ptr(g), M, None, Cout, L.ptr(wk), Cin, L.ptr(nbr), ld, K, L.ptr(dX), L.stream())
    else:
        L.call("dd3d_conv_any_bwd_data", L.ptr(g), M, None, Cout, L.ptr(w), Cin, L.ptr(nbr), ld, K, L.ptr(dX), L.stream())
    return dX


def conv_bwd_weight_raw(x, g, nbr, ld, M_out, K):
    """dW[k] = sum_o x[nbr[k][o]]^T g[o], [K, Cin, Cout]."""
    Cin, Cout = x.shape[1], g.shape[1]
    if _hot_shape(K, Cin, Cout):
        return conv_bwd_weight(x, g, nbr, ld, M_out, _single_group(M_out, x.device), 1)[0]
    dW = torch.empty((K, Cin, Cout), dtype=_F32, device=x.device)
    ws = L.workspace(int(L.lib.dd3d_conv_any_bwd_weight_ws_bytes(K, Cin, Cout)), x.device)
    L.call("dd3d_conv_any_bwd_weight", L.ptr(x), Cin, L.ptr(g), Cout, L.ptr(nbr), ld, K, M_out, None, L.ptr(dW), L.ptr(ws), ws.numel(), L.stream())
    return dW


class SparseConvFn(Function):
    """ME.MinkowskiConvolution(stride=1, bias=False): Y[o] = sum_k X[nbr[k][o]] @ W[k], W [K, Cin, Cout] (``nbr`` None: kernel_size 1).
    Any channel widths up to 128.  The backward calls kernels on raw buffers: a constant under ``create_graph=True``, like MinkowskiEngine's."""

    @staticmethod
    def forward(ctx, X, W, nbr, ld, M_out):
        Y = conv_fwd_raw(_c(X), _c(W), nbr, ld, M_out)
        ctx.save_for_backward(X, W)
        ctx.nbr, ctx.ld, ctx.M_out = nbr, ld, M_out
        return Y

    @staticmethod
    def backward(ctx, g):
        X, W = ctx.saved_tensors
        x, w, g = _c(X), _c(W), _c(g)
        dX = dW = None
        if ctx.needs_input_grad[0]:
            dX = conv_bwd_data_raw(g, w, ctx.nbr, ctx.ld, W)
        if ctx.needs_input_grad[1]:
            dW = conv_bwd_weight_raw(x, g, ctx.nbr, ctx.ld, ctx.M_out, w.shape[0])
        return dX, dW, None, None, None


class SparseConvTrueFn(Function):
    """SparseConvFn for the true-second-order research mode: the same forward; the backward is ``_ConvGradFn``, a Function with a backward
    of its own, so gradients of gradients flow through the convolution (and, since that backward is again made of these two Functions,
    to any order)."""

    @staticmethod
    def forward(ctx, X, W, nbr, ld, M_out):
        Y = conv_fwd_raw(_c(X), _c(W), nbr, ld, M_out)
        ctx.save_for_backward(X, W)
        ctx.nbr, ctx.ld, ctx.M_out = nbr, ld, M_out
        return Y

    @staticmethod
    def backward(ctx, g):
        X, W = ctx.saved_tensors
        dX, dW = _ConvGradFn.apply(g, X, W, ctx.nbr, ctx.ld, ctx.M_out, ctx.needs_input_grad[0], ctx.needs_input_grad[1])
        return dX, dW, None, None, None


class _ConvGradFn(Function):
    """(g, X, W) -> (dX, dW) of a stride-1 sparse convolution: dX = sum_k scatter(g W[k]^T), dW[k] = X_k^T g (X_k = the rows gathered by
    offset k).  Both are bilinear, so the backward is made of the same three kernels:
      cotangent ggX of dX:  d g = conv(ggX, W),  d W[k] = ggX_k^T g;      cotangent ggW of dW:  d g += conv(X, ggW),  d X = bwd_data(g, ggW)."""

    @staticmethod
    def forward(ctx, g, X, W, nbr, ld, M_out, want_dx, want_dw):
        gq, x, w = _c(g), _c(X), _c(W)
        dX = conv_bwd_data_raw(gq, w, nbr, ld, W) if want_dx else None
        dW = conv_bwd_weight_raw(x, gq, nbr, ld, M_out, w.shape[0]) if want_dw else None
        ctx.save_for_backward(g, X, W)
        ctx.meta = (nbr, ld, M_out)
        return dX, dW

    @staticmethod
    def backward(ctx, ggX, ggW):
        g, X, W = ctx.saved_tensors
        meta = ctx.meta
        need_g, need_x, need_w = ctx.needs_input_grad[:3]
        d_g = d_x = d_w = None
        if ggX is not None:
            if need_g:
                d_g = SparseConvTrueFn.apply(ggX, W, *meta)
            if need_w:
                d_w = _ConvGradFn.apply(g, ggX, W, *meta, False, True)[1]
        if ggW is not None:
            if need_g:
                t = SparseConvTrueFn.apply(X, ggW, *meta)
                d_g = t if d_g is None else d_g + t
            if need_x:
                d_x = _ConvGradFn.apply(g, X, ggW, *meta, True, False)[0]
        return d_g, d_x, d_w, None, None, None, None, None


def max_pool_true(X, kids, ld, M_out):
    """MinkowskiMaxPooling(2, 2) for the true-second-order mode: the winners come from the HIP kernel, the values from a differentiable gather."""
    x = _c(X)
    out = torch.empty((M_out, x.shape[1]), dtype=_F32, device=x.device)
    arg = torch.empty((M_out, x.shape[1]), dtype=torch.int32, device=x.device)
    L.call("dd3d_affine_relu_pool_fwd", L.ptr(x), None, None, 0, x.shape[1], L.ptr(kids), ld, kids.shape[0], M_out, None, L.ptr(out), L.ptr(arg), L.stream())
    return X.gather(0, arg.long())


def instance_norm_torch(F, b, B):
    """MinkowskiInstanceNorm's normalisation (biased variance, eps = 1e-8 inside the sqrt) as differentiable torch ops on the device:
    the channel counts no HIP kernel takes (64, 32, 16: the reference's unused MEConv stack) and the true-second-order mode."""
    cnt = torch.bincount(b, minlength=B).to(F.dtype).clamp_(min=1)[:, None]
    seg = lambda t: torch.zeros((B, t.shape[1]), dtype=t.dtype, device=t.device).index_add(0, b, t)
    cen = F - (seg(F) / cnt)[b]
    return cen * (1.0 / torch.sqrt(seg(cen * cen) / cnt + 1e-8))[b]


# ----------------------------------------------------------------------------------------------- instance norm (stand-alone)
class InstanceNormFn(Function):
    """MinkowskiInstanceNormFunction: per cloud and channel, biased variance, eps = 1e-8 inside the sqrt."""

    @staticmethod
    def forward(ctx, Y, seg_off, B):
        y = _c(Y)
        xhat = torch.empty_like(y)
        inv_std = torch.empty((B, y.shape[1]), dtype=_F32, device=y.device)
        L.call("dd3d_inorm_fwd", L.ptr(y), L.ptr(seg_off), B, y.shape[1], L.ptr(xhat), L.ptr(inv_std), L.stream())
        ctx.saved = (xhat, inv_std, seg_off, B)
        return xhat

    @staticmethod
    def backward(ctx, g):
        xhat, inv_std, seg_off, B = ctx.saved
        g = _c(g)
        dY = torch.empty_like(g)
        L.call("dd3d_inorm_bwd", L.ptr(g), L.ptr(xhat), L.ptr(inv_std), L.ptr(seg_off), B, g.shape[1], L.ptr(dY), L.stream())
        return dY, None, None


# ----------------------------------------------------------------------------------------------- max pool (stand-alone)
def _arp_bwd(g_pool, arg, parent, xhat, w, b, use_ar, group_row_off, G, want_gw, want_ga, ga_prev=None, gw_extra=None):
    M_in, C = xhat.shape
    dev = xhat.device
    g_xhat = torch.empty_like(xhat)
    ga = torch.empty_like(xhat) if want_ga else None
    gw_w = torch.empty((G, C), dtype=_F32, device=dev) if want_gw else None
    gw_b = torch.empty((G, C), dtype=_F32, device=dev) if want_gw else None
    ws = L.workspace(int(L.lib.dd3d_arp_bwd_ws_bytes(M_in, G)), dev)
    L.call("dd3d_affine_relu_pool_bwd", L.ptr(g_pool), L.ptr(arg), L.ptr(parent), L.ptr(xhat), L.ptr(w), L.ptr(b), int(use_ar), C, M_in,
           L.ptr(group_row_off), G, L.ptr(g_xhat), L.ptr(ga), L.ptr(ga_prev), L.ptr(gw_extra), L.ptr(gw_w), L.ptr(gw_b), L.ptr(ws), ws.numel(),
           L.stream())
    return g_xhat, ga, gw_w, gw_b


class MaxPoolFn(Function):
    """ME.MinkowskiMaxPooling(kernel_size=2, stride=2): channel-wise max over the existing children of every output voxel."""

    @staticmethod
    def forward(ctx, X, kids, ld, parent, M_out):
        x = _c(X)
        out = torch.empty((M_out, x.shape[1]), dtype=_F32, device=x.device)
        arg = torch.empty((M_out, x.shape[1]), dtype=torch.int32, device=x.device)
        L.call("dd3d_affine_relu_pool_fwd", L.ptr(x), None, None, 0, x.shape[1], L.ptr(kids), ld, kids.shape[0], M_out, None, L.ptr(out), L.ptr(arg), L.stream())
        ctx.saved = (arg, parent, x)
        return out

    @staticmethod
    def backward(ctx, g):
        arg, parent, x = ctx.saved
        g_x, _, _, _ = _arp_bwd(_c(g), arg, parent, x, None, None, 0, _single_group(x.shape[0], x.device), 1, False, False)
        return g_x, None, None, None, None


# ----------------------------------------------------------------------------------------------- the reference's unused variants (f4)
def sparse_conv_any(X, W, nbr, ld, M_out):
    """MinkowskiConvolution for any channel widths and kernel sizes (the reference's UNUSED models, models/MEConv.py:25-39,102-130: 1 -> 128,
    128 -> 64 -> 32 -> 16, kernel-size-1 convolutions): hand-written HIP kernels (conv_any.hip) behind the same Function as the hot shapes."""
    fn = SparseConvTrueFn if _TRUE_SECOND_ORDER else SparseConvFn
    return fn.apply(X, W, nbr, ld, M_out)


def avg_pool_torch(X, kids, M_out):
    """MinkowskiAvgPooling(2, 2) as differentiable torch ops (true-second-order mode only; otherwise AvgPoolFn's HIP kernels)."""
    s = torch.zeros((M_out, X.shape[1]), dtype=X.dtype, device=X.device)
    n = torch.zeros((M_out, 1), dtype=X.dtype, device=X.device)
    for k in range(kids.shape[0]):
        kk = kids[k, :M_out].long()
        have = (kk >= 0)[:, None]
        s = s + torch.where(have, X[kk.clamp(min=0)], torch.zeros((), dtype=X.dtype, device=X.device))
        n = n + have.to(X.dtype)
    return s / n.clamp(min=1)


class AvgPoolFn(Function):
    """ME.MinkowskiAvgPooling(kernel_size=2, stride=2) (models/MEConv.py:28,35,105,112,126): channel-wise mean over the existing
    children of every output voxel.  First-order backward only, like MinkowskiEngine's native pooling function."""

    @staticmethod
    def forward(ctx, X, kids, ld, parent, M_out):
        x = _c(X)
        C = x.shape[1]
        out = torch.empty((M_out, C), dtype=_F32, device=x.device)
        cnt = torch.empty((max(M_out, 1),), dtype=torch.int32, device=x.device)
        L.call("dd3d_avg_pool_fwd", L.ptr(x), x.shape[0], C, L.ptr(kids), ld, kids.shape[0], M_out, None, L.ptr(out), L.ptr(cnt), L.stream())
        ctx.saved = (cnt, parent, x.shape[0], M_out, C)
        return out

    @staticmethod
    def backward(ctx, g):
        cnt, parent, M_in, M_out, C = ctx.saved
        dX = torch.zeros((M_in, C), dtype=_F32, device=g.device)
        L.call("dd3d_avg_pool_bwd", L.ptr(_c(g)), L.ptr(cnt), L.ptr(parent), M_in, M_out, C, L.ptr(dX), L.stream())
        return dX, None, None, None, None


class GlobalAvgPoolFn(Function):
    """ME.MinkowskiGlobalAvgPooling (models/MEConv.py:20,97): one row per cloud, the mean of the cloud's voxel features."""

    @staticmethod
    def forward(ctx, X, seg_off, B):
        x = _c(X)
        M, C = x.shape
        out = torch.empty((B, C), dtype=_F32, device=x.device)
        L.call("dd3d_global_avg_pool_fwd", L.ptr(x), M, L.ptr(seg_off), B, C, L.ptr(out), L.stream())
        ctx.saved = (seg_off, B, M, C)
        return out

    @staticmethod
    def backward(ctx, g):
        seg_off, B, M, C = ctx.saved
        dX = torch.zeros((M, C), dtype=_F32, device=g.device)
        L.call("dd3d_global_avg_pool_bwd", L.ptr(_c(g)), M, L.ptr(seg_off), B, C, L.ptr(dX), L.stream())
        return dX, None, None


# ----------------------------------------------------------------------------------------------- fused norm + affine + ReLU + pool
class _NormActPoolBwdFn(Function):
    """(g_pool, Y, w, b) -> (dY, dw, db).  Its own backward implements the only second-order term of the block:
    dw = sum_rows ga * xhat depends on Y through xhat (ga is a constant)."""

    @staticmethod
    def forward(ctx, g_pool, Y, w, b, st):
        xhat, inv_std, arg, parent, seg_off, B, gro, G = st
        wq, bq = _c(w).reshape(-1), _c(b).reshape(-1)
        g_xhat, ga, gw_w, gw_b = _arp_bwd(_c(g_pool), arg, parent, xhat, wq, bq, 1, gro, G, True, True)
        dY = torch.empty_like(xhat)
        L.call("dd3d_inorm_bwd", L.ptr(g_xhat), L.ptr(xhat), L.ptr(inv_std), L.ptr(seg_off), B, xhat.shape[1], L.ptr(dY), L.stream())
        ctx.saved = (ga, xhat, inv_std, seg_off, B, gro, G)
        ctx.wshape = w.shape
        ctx.mark_non_differentiable(dY)
        return dY, gw_w.reshape(w.shape) if G == 1 else gw_w, gw_b.reshape(b.shape) if G == 1 else gw_b

    @staticmethod
    def backward(ctx, _ggY, ggw, _ggb):
        ga, xhat, inv_std, seg_off, B, gro, G = ctx.saved
        if ggw is None:
            return None, None, None, None, None
        ggw = _c(ggw).reshape(G, -1)
        gx = torch.empty_like(xhat)
        L.call("dd3d_scale_rows_by_group", L.ptr(ga), L.ptr(ggw), L.ptr(gro), G, xhat.shape[0], xhat.shape[1], L.ptr(gx), L.stream())
        dY = torch.empty_like(xhat)
        L.call("dd3d_inorm_bwd", L.ptr(gx), L.ptr(xhat), L.ptr(inv_std), L.ptr(seg_off), B, xhat.shape[1], L.ptr(dY), L.stream())
        return None, dY, None, None, None


class NormActPoolFn(Function):
    """MinkowskiInstanceNorm(128) -> MinkowskiReLU -> MinkowskiMaxPooling(2,2) (models/MEConv.py:225-227) as two kernels."""

    @staticmethod
    def forward(ctx, Y, w, b, seg_off, B, kids, ld, parent, M_out, gro, G):
        y = _c(Y)
        C = y.shape[1]
        xhat = torch.empty_like(y)
        inv_std = torch.empty((B, C), dtype=_F32, device=y.device)
        L.call("dd3d_inorm_fwd", L.ptr(y), L.ptr(seg_off), B, C, L.ptr(xhat), L.ptr(inv_std), L.stream())
        out = torch.empty((M_out, C), dtype=_F32, device=y.device)
        arg = torch.empty((M_out, C), dtype=torch.int32, device=y.device)
        L.call("dd3d_affine_relu_pool_fwd", L.ptr(xhat), L.ptr(_c(w).reshape(-1)), L.ptr(_c(b).reshape(-1)), 1, C, L.ptr(kids), ld, kids.shape[0],
               M_out, None, L.ptr(out), L.ptr(arg), L.stream())
        ctx.save_for_backward(Y, w, b)
        ctx.st = (xhat, inv_std, arg, parent, seg_off, B, gro, G)
        return out

    @staticmethod
    def backward(ctx, g):
        Y, w, b = ctx.saved_tensors
        dY, dw, db = _NormActPoolBwdFn.apply(g, Y, w, b, ctx.st)
        return dY, dw, db, None, None, None, None, None, None, None, None


# ----------------------------------------------------------------------------------------------- classifier over the dense placement
class DensePlacement:
    """site / vox_at tables of SparseTensor.dense(shape=[B,C,E0,E1,E2], min_coordinate=batch min) for the coarsest level."""

    def __init__(self, cm: CoordinateManager, stride: int, extent, cloud_group=None, G: int = 1, group_cloud_off=None):
        lv = cm.ensure_level(stride)
        dev = cm.device
        self.B, self.G = cm.B, G
        self.E = tuple(int(e) for e in extent)
        self.S = self.E[0] * self.E[1] * self.E[2]
        self.cloud_group = cloud_group
        self.group_cloud_off = group_cloud_off if group_cloud_off is not None else torch.tensor([0, self.B], dtype=torch.int32, device=dev)
        self.seg_off = cm.seg_off(stride)
        self.C, self.M_bound = lv.C, max(cm._rows_bound(lv), 1)   # voxel rows (column 0 = cloud) and their bound
        pre = getattr(cm, "dense_tables", None)
        if pre is not None and stride == 4 and pre[0] == self.E and pre[1] is cloud_group:
            _, _, self.gmin, self.site, self.vox_at = pre     # the cloud-local map builder already placed the voxels
            return
        bound = cm._rows_bound(lv)
        self.site = torch.empty(max(bound, 1), dtype=torch.int32, device=dev)
        self.vox_at = torch.empty((self.S, max(self.B, 1)), dtype=torch.int32, device=dev)
        self.gmin = torch.empty((G, 3), dtype=torch.int32, device=dev)
        L.call("dd3d_dense_sites", L.ptr(lv.C), cm._count_ptr(lv), bound, L.ptr(cloud_group), G, self.B, stride, self.E[0], self.E[1], self.E[2],
               L.ptr(self.gmin), L.ptr(self.site), L.ptr(self.vox_at), L.ptr(cm.err), L.stream())
        cm._synced = False


def sl_fwd_call(F, dp, cloud_group, Wl, strides, J, bias, out, B=None, M_bound=None):
    """dd3d_sparse_linear_fwd on raw buffers: out[B,J] = sum over each cloud's voxels of F[v] . Wl[group][.,.,site(v)] (+ bias).
    ``B`` / ``M_bound``: restrict the call to the first B clouds (their rows are a prefix of the row range, bounded by M_bound)."""
    B = dp.B if B is None else B
    M_bound = dp.M_bound if M_bound is None else M_bound
    ws = L.workspace(int(L.lib.dd3d_sparse_linear_ws_bytes(M_bound, J)), F.device)
    L.call("dd3d_sparse_linear_fwd", L.ptr(F), L.ptr(dp.site), L.ptr(dp.C), M_bound, L.ptr(dp.seg_off), B, 128, L.ptr(cloud_group), L.ptr(Wl),
           strides, J, L.ptr(bias), L.ptr(out), L.ptr(ws), ws.numel(), L.stream())


def sl_bwd_x_call(dl, dp, cloud_group, Wl, strides, J, accumulate, dF, B=None, M_bound=None):
    """dd3d_sparse_linear_bwd_x on raw buffers: dF[v] (+)= sum_j dl[cloud(v)][j] * Wl[group][j][., site(v)] (same B / M_bound)."""
    B = dp.B if B is None else B
    M_bound = dp.M_bound if M_bound is None else M_bound
    L.call("dd3d_sparse_linear_bwd_x", L.ptr(dl), L.ptr(dp.site), L.ptr(dp.C), M_bound, L.ptr(dp.seg_off), B, 128, L.ptr(cloud_group), L.ptr(Wl),
           strides, J, int(accumulate), L.ptr(dF), L.stream())


def _param_strides(J, C, S):
    import ctypes
    return (ctypes.c_int64 * 4)(J * C * S, C * S, S, 1)


def _site_strides(J, C, S):
    import ctypes
    return (ctypes.c_int64 * 4)(J * C * S, C, 1, J * C)


def site_major(Wl: torch.Tensor, C: int, S: int) -> torch.Tensor:
    """Site-major copy [.., S, J, C] of a weight-like operand in the parameter layout [.., J, C*S]: every voxel then reads its C
    weights of one output as one contiguous 512-B row.  Cached on the tensor object (version-checked), so the frozen classifier
    weight is converted once."""
    hit = getattr(Wl, "_dd3d_sm", None)
    if hit is not None and hit[0] == Wl._version and hit[1].device == Wl.device:
        return hit[1]
    w = _c(Wl)
    G = w.shape[0] if w.dim() == 3 else 1
    J = w.shape[-2]
    out = torch.empty_like(w)
    L.call("dd3d_classifier_layout", L.ptr(w), G, J, C, S, 0, L.ptr(out), L.stream())
    try:
        Wl._dd3d_sm = (Wl._version, out)
    except Exception:
        pass
    return out


class _SLFwd(Function):
    @staticmethod
    def forward(ctx, F, Wl, dp: DensePlacement):
        f = _c(F)
        w = site_major(Wl, f.shape[1], dp.S)
        J = w.shape[-2]
        out = torch.empty((dp.B, J), dtype=_F32, device=f.device)
        sl_fwd_call(f, dp, dp.cloud_group if w.dim() == 3 else None, w, _site_strides(J, f.shape[1], dp.S), J, None, out)
        ctx.save_for_backward(F, Wl)
        ctx.dp = dp
        return out

    @staticmethod
    def backward(ctx, g):
        F, Wl = ctx.saved_tensors
        dF = _SLBwdX.apply(g, Wl, ctx.dp) if ctx.needs_input_grad[0] else None
        dW = _SLBwdW.apply(g, F, ctx.dp, Wl.dim() == 3) if ctx.needs_input_grad[1] else None
        return dF, dW, None


class _SLBwdX(Function):
    @staticmethod
    def forward(ctx, dl, Wl, dp: DensePlacement):
        d = _c(dl)
        w = site_major(Wl, 128, dp.S)
        J = d.shape[1]
        M = int(dp.seg_off_rows)
        dF = torch.empty((M, 128), dtype=_F32, device=d.device)
        sl_bwd_x_call(d, dp, dp.cloud_group if w.dim() == 3 else None, w, _site_strides(J, 128, dp.S), J, 0, dF)
        ctx.save_for_backward(dl, Wl)
        ctx.dp = dp
        return dF

    @staticmethod
    def backward(ctx, g):
        dl, Wl = ctx.saved_tensors
        d_dl = _SLFwd.apply(g, Wl, ctx.dp) if ctx.needs_input_grad[0] else None
        d_W = _SLBwdW.apply(dl, g, ctx.dp, Wl.dim() == 3) if ctx.needs_input_grad[1] else None
        return d_dl, d_W, None


class _SLBwdW(Function):
    @staticmethod
    def forward(ctx, dl, F, dp: DensePlacement, grouped: bool):
        d, f = _c(dl), _c(F)
        J, C = d.shape[1], f.shape[1]
        G = dp.G if grouped else 1
        shape = (G, J, C * dp.S) if grouped else (J, C * dp.S)
        dW = torch.empty(shape, dtype=_F32, device=d.device)
        gco = dp.group_cloud_off if grouped else dp.whole_batch_off
        L.call("dd3d_sparse_linear_bwd_w", L.ptr(d), L.ptr(f), L.ptr(dp.vox_at), dp.S, dp.B, L.ptr(gco), G, C, J, L.ptr(dW), _param_strides(J, C, dp.S),
               dW.numel(), L.stream())
        ctx.save_for_backward(dl, F)
        ctx.dp, ctx.grouped = dp, grouped
        return dW

    @staticmethod
    def backward(ctx, g):
        dl, F = ctx.saved_tensors
        d_dl = _SLFwd.apply(F, g, ctx.dp) if ctx.needs_input_grad[0] else None
        d_F = _SLBwdX.apply(dl, g, ctx.dp) if ctx.needs_input_grad[1] else None
        return d_dl, d_F, None, None


def sparse_linear(F: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], dp: DensePlacement) -> torch.Tensor:
    """logits[B,J] = dense(F).view(B,-1) @ weight.T + bias without materialising the dense tensor (models/MEConv.py:254-256)."""
    dp.seg_off_rows = F.shape[0]
    if not hasattr(dp, "whole_batch_off"):
        dp.whole_batch_off = torch.tensor([0, dp.B], dtype=torch.int32, device=F.device)
    out = _SLFwd.apply(F, weight, dp)
    return out + bias if bias is not None else out
