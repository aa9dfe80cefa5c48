"""Pure-torch / numpy CPU restatement of the MinkowskiEngine 0.5.4 subset that the
dd3d hot path reaches.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

PARITY UNPINNED at this boundary: MinkowskiEngine==0.5.4 (pinned in
/root/reference/environment.yml:278) is a third-party native dependency whose
source is not part of the reference tree and which cannot be installed offline.
What follows restates its published behaviour (SURVEY.md Appendix A, items
A1-A9); every function names the reference call site that reaches it.

Design notes that matter for parity with the reference:

* Every ME op that is a ``torch.autograd.Function`` with a native backward is
  restated as a Function whose backward runs under ``torch.no_grad()``: under
  ``create_graph=True`` (mnist_distill.py:359) its outputs are therefore
  constants, exactly as with ME's opaque pybind backward.
* Everything the reference does with torch-native ops (instance-norm affine,
  ReLU, ``dense()`` index assignment, Linear, CrossEntropy) stays torch-native
  here, so torch autograd provides the same second-order terms.
* Canonical orders: voxel rows in first-occurrence order of the
  input rows; kernel offsets ``k = (d0+1) + 3(d1+1) + 9(d2+1)`` with d0 the first
  spatial column; pooling offsets ``k = e0 + 2 e1 + 4 e2``; pairs inside an
  offset sorted by output row.

All functions are dtype-agnostic (run them in float64 for a tight checker).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn as nn
from torch.autograd import Function

# --------------------------------------------------------------------------------------
# integer coordinate machinery (numpy, exact)
# --------------------------------------------------------------------------------------


class _KeySpace:
    """Mixed-radix packing of int (b, c1, c2, c3) rows into one int64 key (exact, no hashing)."""

    def __init__(self, C: np.ndarray, margin: int):
        C = np.asarray(C, dtype=np.int64).reshape(-1, 4)
        if C.shape[0] == 0:
            self.lo = np.zeros(4, np.int64)
            self.ext = np.ones(4, np.int64)
        else:
            m = np.array([0, margin, margin, margin], np.int64)
            self.lo = C.min(0) - m
            self.ext = C.max(0) + m - self.lo + 1
        assert float(np.prod(self.ext.astype(np.float64))) < 2.0 ** 62, "coordinate extent too large for the oracle"

    def pack(self, Q: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
        Q = np.asarray(Q, dtype=np.int64)
        rel = Q - self.lo
        valid = np.all((rel >= 0) & (rel < self.ext), axis=1)
        rel = np.where(valid[:, None], rel, 0)
        key = ((rel[:, 0] * self.ext[1] + rel[:, 1]) * self.ext[2] + rel[:, 2]) * self.ext[3] + rel[:, 3]
        return key, valid


def unique_rows_first_occurrence(C: np.ndarray) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """Unique rows of int ``C[N,4]`` in first-occurrence order.

    Returns (unique_index[M] ascending input rows, inverse[N] row->voxel, C_unique[M,4]).
    ME: CoordinateMapManager insert_and_map, CPU insertion order."""
    C = np.asarray(C, dtype=np.int64).reshape(-1, 4)
    if C.shape[0] == 0:
        return np.zeros(0, np.int64), np.zeros(0, np.int64), C.astype(np.int32)
    ks = _KeySpace(C, 0)
    key, _ = ks.pack(C)
    _, first, inverse = np.unique(key, return_index=True, return_inverse=True)
    order = np.argsort(first, kind="stable")  # sorted-key id -> rank in first-occurrence order
    rank = np.empty_like(order)
    rank[order] = np.arange(order.size)
    unique_index = first[order]
    inv = rank[inverse.reshape(-1)]
    return unique_index.astype(np.int64), inv.astype(np.int64), C[unique_index].astype(np.int32)


def quantize_unique(coords: torch.Tensor):
    """``TensorField.sparse()`` coordinate half (models/MEConv.py:247,199).

    coords: float [N,4] (b, c1, c2, c3).  floor -> int32 -> unique (first occurrence).
    Returns C0 int32 [M,4], inverse int64 [N], unique_index int64 [M]."""
    q = np.floor(coords.detach().cpu().double().numpy()).astype(np.int64)
    uidx, inv, C0 = unique_rows_first_occurrence(q)
    return torch.from_numpy(C0), torch.from_numpy(inv), torch.from_numpy(uidx)


def floor_div(a: np.ndarray, d: int) -> np.ndarray:
    return np.floor_divide(a, d)


def stride_map(C: torch.Tensor, new_stride: int):
    """Coordinates of a stride-2 pooling output (MinkowskiMaxPooling, models/MEConv.py:227).

    out = unique(floor(c / new_stride) * new_stride) over the spatial columns, first-occurrence order.
    Returns C_out int32 [Mo,4], parent int64 [M] (output row of each input row)."""
    c = C.cpu().numpy().astype(np.int64).copy()
    c[:, 1:] = floor_div(c[:, 1:], new_stride) * new_stride
    _, inv, Cout = unique_rows_first_occurrence(c)
    return torch.from_numpy(Cout), torch.from_numpy(inv)


def region_offsets(kernel_size: int, tensor_stride: int) -> np.ndarray:
    """Hyper-cube region offsets, first spatial axis fastest.
    odd kernel -> centred {-1,0,1}*ts ; even kernel -> {0,1}*ts."""
    if kernel_size % 2 == 1:
        r = np.arange(kernel_size) - kernel_size // 2
    else:
        r = np.arange(kernel_size)
    offs = []
    for d2 in r:
        for d1 in r:
            for d0 in r:
                offs.append((d0 * tensor_stride, d1 * tensor_stride, d2 * tensor_stride))
    return np.asarray(offs, dtype=np.int64)


def kernel_map(C_in: torch.Tensor, C_out: torch.Tensor, kernel_size: int, tensor_stride: int) -> torch.Tensor:
    """Neighbour table nbr[K, M_out]: input row i with C_in[i] == C_out[o] + offset_k, else -1."""
    cin = C_in.cpu().numpy().astype(np.int64).reshape(-1, 4)
    cout = C_out.cpu().numpy().astype(np.int64).reshape(-1, 4)
    offs = region_offsets(kernel_size, tensor_stride)
    K = offs.shape[0]
    nbr = np.full((K, cout.shape[0]), -1, dtype=np.int64)
    if cin.shape[0] == 0 or cout.shape[0] == 0:
        return torch.from_numpy(nbr)
    ks = _KeySpace(cin, 0)
    key_in, _ = ks.pack(cin)
    order = np.argsort(key_in, kind="stable")
    sorted_keys = key_in[order]
    for k in range(K):
        q = cout.copy()
        q[:, 1:] += offs[k]
        key_q, valid = ks.pack(q)
        pos = np.searchsorted(sorted_keys, key_q)
        pos_c = np.minimum(pos, sorted_keys.size - 1)
        hit = valid & (sorted_keys[pos_c] == key_q)
        nbr[k, hit] = order[pos_c[hit]]
    return torch.from_numpy(nbr)


def kernel_map_bruteforce(C_in, C_out, kernel_size: int, tensor_stride: int) -> torch.Tensor:
    """O(M*K) python-dict builder, for small known-answer tests only."""
    cin = [tuple(int(v) for v in r) for r in C_in.tolist()]
    table = {c: i for i, c in enumerate(cin)}
    offs = region_offsets(kernel_size, tensor_stride)
    nbr = np.full((offs.shape[0], len(C_out)), -1, dtype=np.int64)
    for o, c in enumerate(C_out.tolist()):
        for k, (d0, d1, d2) in enumerate(offs.tolist()):
            nbr[k, o] = table.get((int(c[0]), int(c[1]) + d0, int(c[2]) + d1, int(c[3]) + d2), -1)
    return torch.from_numpy(nbr)


def pairs_from_nbr(nbr: torch.Tensor):
    """Canonical pair lists: grouped by offset k, sorted by out_row inside each offset.
    Returns in_rows int64 [P], out_rows int64 [P], offs int64 [K+1]."""
    K, M = nbr.shape
    ins, outs, offs = [], [], [0]
    for k in range(K):
        o = torch.nonzero(nbr[k] >= 0).reshape(-1)
        ins.append(nbr[k][o])
        outs.append(o)
        offs.append(offs[-1] + o.numel())
    return torch.cat(ins), torch.cat(outs), torch.tensor(offs, dtype=torch.int64)


def voxel_csr(inv: torch.Tensor, M: int):
    """Per-voxel point lists, points ascending: (offsets int64 [M+1], points int64 [N])."""
    inv_np = inv.cpu().numpy()
    order = np.argsort(inv_np, kind="stable")
    cnt = np.bincount(inv_np, minlength=M)
    off = np.zeros(M + 1, np.int64)
    np.cumsum(cnt, out=off[1:])
    return torch.from_numpy(off), torch.from_numpy(order.astype(np.int64))


# --------------------------------------------------------------------------------------
# feature ops as autograd Functions with OPAQUE backward (no_grad), like ME's native ones
# --------------------------------------------------------------------------------------

# The checker of the product's true-second-order research mode (dd3d_amd.ops.true_second_order; SURVEY.md section 8f row 4 -- NOT the
# reference's semantics): inside ``true_second_order()`` the modules below compute with plain differentiable torch ops instead of the
# opaque Functions, so torch autograd differentiates every sparse op twice.  Same forward values.
_TRUE_SECOND_ORDER = False


class true_second_order:
    def __enter__(self):
        global _TRUE_SECOND_ORDER
        self.prev, _TRUE_SECOND_ORDER = _TRUE_SECOND_ORDER, True
        return self

    def __exit__(self, *exc):
        global _TRUE_SECOND_ORDER
        _TRUE_SECOND_ORDER = self.prev


def _conv_plain(X, W, nbr):
    Y = torch.zeros(nbr.shape[1], W.shape[2], dtype=X.dtype)
    for k in range(nbr.shape[0]):
        o = torch.nonzero(nbr[k] >= 0).reshape(-1)
        if o.numel():
            Y = Y.index_add(0, o, X[nbr[k][o]] @ W[k])
    return Y


def _inorm_plain(X, bidx, B):
    cnt = torch.bincount(bidx, minlength=B).to(X.dtype).clamp_(min=1)[:, None]
    seg = lambda t: torch.zeros((B, t.shape[1]), dtype=t.dtype).index_add(0, bidx, t)
    cen = X - (seg(X) / cnt)[bidx]
    return cen * (1.0 / torch.sqrt(seg(cen * cen) / cnt + 1e-8))[bidx]



def _seg_sum(x: torch.Tensor, idx: torch.Tensor, n: int) -> torch.Tensor:
    out = torch.zeros((n,) + tuple(x.shape[1:]), dtype=x.dtype)
    return out.index_add_(0, idx, x)


class VoxelMeanFn(Function):
    """MinkowskiSPMMAverageFunction (UNWEIGHTED_AVERAGE quantisation)."""

    @staticmethod
    def forward(ctx, feats, inv, M):
        cnt = torch.bincount(inv, minlength=M).to(feats.dtype)
        ctx.save_for_backward(inv, cnt)
        return _seg_sum(feats, inv, M) / cnt[:, None]

    @staticmethod
    def backward(ctx, g):
        inv, cnt = ctx.saved_tensors
        with torch.no_grad():
            d = g[inv] / cnt[inv][:, None]
        return d, None, None


class SparseConvFn(Function):
    """MinkowskiConvolutionFunction: Y[o] += X[i] @ W[k] over pairs of offset k."""

    @staticmethod
    def forward(ctx, X, W, nbr):
        K, Mo = nbr.shape
        Y = torch.zeros(Mo, W.shape[2], dtype=X.dtype)
        for k in range(K):
            o = torch.nonzero(nbr[k] >= 0).reshape(-1)
            if o.numel():
                Y.index_add_(0, o, X[nbr[k][o]] @ W[k])
        ctx.save_for_backward(X, W, nbr)
        return Y

    @staticmethod
    def backward(ctx, g):
        X, W, nbr = ctx.saved_tensors
        with torch.no_grad():
            dX = torch.zeros_like(X)
            dW = torch.zeros_like(W)
            for k in range(nbr.shape[0]):
                o = torch.nonzero(nbr[k] >= 0).reshape(-1)
                if o.numel():
                    i = nbr[k][o]
                    go = g[o]
                    dX.index_add_(0, i, go @ W[k].t())
                    dW[k] = X[i].t() @ go
        return dX, dW, None


class InstanceNormFn(Function):
    """MinkowskiInstanceNormFunction: biased variance, eps=1e-8 inside the sqrt."""

    @staticmethod
    def forward(ctx, X, bidx, B):
        cnt = torch.bincount(bidx, minlength=B).to(X.dtype).clamp_(min=1)[:, None]
        mean = _seg_sum(X, bidx, B) / cnt
        cen = X - mean[bidx]
        var = _seg_sum(cen * cen, bidx, B) / cnt
        inv_std = 1.0 / torch.sqrt(var + 1e-8)
        norm = cen * inv_std[bidx]
        ctx.save_for_backward(inv_std, norm, bidx, cnt)
        ctx.B = B
        return norm

    @staticmethod
    def backward(ctx, g):
        inv_std, norm, bidx, cnt = ctx.saved_tensors
        with torch.no_grad():
            mean_dout = _seg_sum(g, bidx, ctx.B) / cnt
            mean_dout_feat = _seg_sum(g * norm, bidx, ctx.B) / cnt
            din = (g - norm * mean_dout_feat[bidx] - mean_dout[bidx]) * inv_std[bidx]
        return din, None, None


class MaxPoolFn(Function):
    """MinkowskiLocalPoolingFunction, MAX mode.  kids[Kp, M_out] = child rows or -1."""

    @staticmethod
    def forward(ctx, X, kids):
        Kp, Mo = kids.shape
        best = torch.full((Mo, X.shape[1]), -math.inf, dtype=X.dtype)
        arg = torch.full((Mo, X.shape[1]), -1, dtype=torch.int64)
        for k in range(Kp):
            have = kids[k] >= 0
            v = torch.where(have[:, None], X[kids[k].clamp(min=0)], torch.full_like(best, -math.inf))
            upd = v > best  # strict: the first maximum in offset order wins
            best = torch.where(upd, v, best)
            arg = torch.where(upd, kids[k][:, None].expand_as(arg), arg)
        ctx.save_for_backward(arg)
        ctx.Mi = X.shape[0]
        ctx.mark_non_differentiable(arg)
        return best, arg

    @staticmethod
    def backward(ctx, g, _garg):
        (arg,) = ctx.saved_tensors
        with torch.no_grad():
            dX = torch.zeros(ctx.Mi, g.shape[1], dtype=g.dtype)
            ok = arg >= 0
            dX.scatter_add_(0, arg.clamp(min=0), torch.where(ok, g, torch.zeros_like(g)))
        return dX, None


# --------------------------------------------------------------------------------------
# minimal object model with MinkowskiEngine's names (so the reference's models/MEConv.py can
# run on top of this module in the build container, and so the twin model below reads alike)
# --------------------------------------------------------------------------------------


class CoordinateManager:
    """Caches coordinate maps per tensor stride and kernel maps per (stride, kernel)."""

    def __init__(self):
        self.coords: Dict[int, torch.Tensor] = {}
        self.kmaps: Dict[Tuple[int, int, int], torch.Tensor] = {}
        self.parents: Dict[int, torch.Tensor] = {}

    def get_kernel_map(self, in_stride: int, out_stride: int, kernel_size: int) -> torch.Tensor:
        key = (in_stride, out_stride, kernel_size)
        if key not in self.kmaps:
            self.kmaps[key] = kernel_map(self.coords[in_stride], self.coords[out_stride], kernel_size, in_stride)
        return self.kmaps[key]

    def get_stride_map(self, in_stride: int, out_stride: int):
        if out_stride not in self.coords:
            Cout, parent = stride_map(self.coords[in_stride], out_stride)
            self.coords[out_stride] = Cout
            self.parents[out_stride] = parent
        return self.coords[out_stride]


class SparseTensor:
    def __init__(self, features, coordinates, tensor_stride=1, coordinate_manager=None):
        self.F = features
        self.C = coordinates  # int32 [M,4]
        self.tensor_stride = int(tensor_stride)
        self.coordinate_manager = coordinate_manager or CoordinateManager()
        self.coordinate_manager.coords.setdefault(self.tensor_stride, coordinates)

    @property
    def shape(self):
        return self.F.shape

    def dense(self, shape=None, min_coordinate=None):
        """SparseTensor.dense (models/MEConv.py:254,203)."""
        ts = self.tensor_stride
        assert min_coordinate is not None and shape is not None
        minc = min_coordinate.reshape(-1).to(torch.int64)
        assert bool(torch.all(minc % ts == 0)), "min_coordinate must be a multiple of tensor_stride"
        coords = (self.C[:, 1:].to(torch.int64) - minc[None, :]) // ts
        shape = list(shape)
        if shape[1] != self.F.shape[1]:
            shape[1] = self.F.shape[1]
        dense = torch.zeros(shape, dtype=self.F.dtype)
        b = self.C[:, 0].to(torch.int64)
        dense[b, :, coords[:, 0], coords[:, 1], coords[:, 2]] = self.F
        return dense, min_coordinate, torch.tensor([ts] * 3)


class TensorField:
    """ME.TensorField (models/MEConv.py:263-267): float coords with batch column."""

    def __init__(self, features=None, coordinates=None, device=None, **_):
        self.C = coordinates
        self.F = features
        self.device = device

    @property
    def shape(self):
        return self.F.shape

    def sparse(self):
        C0, inv, _ = quantize_unique(self.C)
        F0 = VoxelMeanFn.apply(self.F, inv, C0.shape[0])
        st = SparseTensor(F0, C0, 1, CoordinateManager())
        st.inverse_mapping = inv
        return st


class MinkowskiNetwork(nn.Module):
    def __init__(self, D):
        super().__init__()
        self.D = D


class MinkowskiConvolution(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size=-1, stride=1, dilation=1, bias=False, dimension=None):
        super().__init__()
        assert stride == 1 and dilation == 1 and not bias and dimension == 3
        self.kernel_size = kernel_size
        self.kernel_volume = kernel_size ** 3
        self.in_channels, self.out_channels = in_channels, out_channels
        # kernel volume 1 (kernel_size=1, only in the reference's unused models): a 2-D [Cin, Cout] parameter, as in MinkowskiEngine
        self.kernel = nn.Parameter(torch.empty(in_channels, out_channels) if self.kernel_volume == 1
                                   else torch.empty(self.kernel_volume, in_channels, out_channels))
        stdv = 1.0 / math.sqrt(in_channels * self.kernel_volume)
        with torch.no_grad():
            self.kernel.uniform_(-stdv, stdv)

    def forward(self, x: SparseTensor):
        ts = x.tensor_stride
        if self.kernel_volume == 1:
            nbr = torch.arange(x.F.shape[0], dtype=torch.int64)[None]
            conv = _conv_plain if _TRUE_SECOND_ORDER else SparseConvFn.apply
            return SparseTensor(conv(x.F, self.kernel[None], nbr), x.C, ts, x.coordinate_manager)
        nbr = x.coordinate_manager.get_kernel_map(ts, ts, self.kernel_size)
        Y = _conv_plain(x.F, self.kernel, nbr) if _TRUE_SECOND_ORDER else SparseConvFn.apply(x.F, self.kernel, nbr)
        return SparseTensor(Y, x.C, ts, x.coordinate_manager)

    def __repr__(self):
        return (f"MinkowskiConvolution(in={self.in_channels}, out={self.out_channels}, "
                f"kernel_size=[{self.kernel_size}]*3, stride=[1, 1, 1], dilation=[1, 1, 1])")


class MinkowskiInstanceNorm(nn.Module):
    def __init__(self, num_features):
        super().__init__()
        self.num_features = num_features
        self.weight = nn.Parameter(torch.ones(1, num_features))
        self.bias = nn.Parameter(torch.zeros(1, num_features))

    def forward(self, x: SparseTensor):
        b = x.C[:, 0].to(torch.int64)
        B = int(b.max().item()) + 1 if b.numel() else 0
        out = _inorm_plain(x.F, b, B) if _TRUE_SECOND_ORDER else InstanceNormFn.apply(x.F, b, B)
        out = out * self.weight + self.bias  # torch-native, like ME
        return SparseTensor(out, x.C, x.tensor_stride, x.coordinate_manager)

    def __repr__(self):
        return f"MinkowskiInstanceNorm(nchannels={self.num_features})"


# ---- test hook: ReLU / max-pool on branches somebody else took -------------------------------------------------------------
# ReLU followed by max-pool is piecewise linear: which child row wins and whether the winner is positive.  With 27 M ReLU inputs per
# cfg2 step, once in a few steps an fp32 pre-activation lies within rounding of zero (or two candidates tie) and an fp32 device
# takes the OTHER branch than this fp64 restatement; the gradients then differ by a jump, not by rounding.  To check such a step to
# the same tolerance as every other, the test replays this restatement ON THE DEVICE'S BRANCHES: inside ``forced_branches`` every
# MinkowskiReLU passes its input through and the i-th MinkowskiMaxPooling call computes out[o][c] = x[arg_i[o][c]][c] if mask_i[o][c]
# else 0 (backward: the gradient goes to that row, nowhere else) -- the same opaque Function structure as MaxPoolFn.
_FORCED: Optional[list] = None


class forced_branches:
    def __init__(self, branches):
        self.branches = list(branches)   # [(arg [M_out, C] int64 rows of the pooled level's input, mask [M_out, C] bool), ...] in call order

    def __enter__(self):
        global _FORCED
        assert _FORCED is None
        _FORCED = list(self.branches)
        return self

    def __exit__(self, *exc):
        global _FORCED
        left, _FORCED = _FORCED, None
        if exc[0] is None:
            assert not left, "fewer pooling calls than forced branch sets"


class ForcedReluPoolFn(Function):
    @staticmethod
    def forward(ctx, X, arg, mask):
        ctx.save_for_backward(arg, mask)
        ctx.Mi = X.shape[0]
        return torch.where(mask, X.gather(0, arg), torch.zeros((), dtype=X.dtype))

    @staticmethod
    def backward(ctx, g):
        arg, mask = ctx.saved_tensors
        with torch.no_grad():
            dX = torch.zeros(ctx.Mi, g.shape[1], dtype=g.dtype)
            dX.scatter_add_(0, arg, torch.where(mask, g, torch.zeros_like(g)))
        return dX, None, None


class MinkowskiReLU(nn.Module):
    def __init__(self, inplace=False):
        super().__init__()
        self.relu = nn.ReLU(inplace=inplace)

    def forward(self, x: SparseTensor):
        if _FORCED is not None:   # the sign is applied by the pooling call that follows (forced_branches)
            return SparseTensor(x.F, x.C, x.tensor_stride, x.coordinate_manager)
        return SparseTensor(self.relu(x.F), x.C, x.tensor_stride, x.coordinate_manager)

    def __repr__(self):
        return "MinkowskiReLU()"


class MinkowskiMaxPooling(nn.Module):
    def __init__(self, kernel_size, stride=1, dilation=1, dimension=None):
        super().__init__()
        assert kernel_size == 2 and stride == 2 and dilation == 1 and dimension == 3
        self.kernel_size, self.stride = kernel_size, stride

    def forward(self, x: SparseTensor):
        ts = x.tensor_stride
        cm = x.coordinate_manager
        Cout = cm.get_stride_map(ts, ts * self.stride)
        kids = cm.get_kernel_map(ts, ts * self.stride, self.kernel_size)
        if _FORCED is not None:
            arg, mask = _FORCED.pop(0)
            assert arg.shape == (Cout.shape[0], x.F.shape[1]) and mask.shape == arg.shape
            # the forced row must be a child of the pooled voxel: a branch the pooling window really has
            assert bool(((kids.t()[:, :, None] == arg[:, None, :]).any(1)).all()), "forced arg-max row is not a child of its pooled voxel"
            return SparseTensor(ForcedReluPoolFn.apply(x.F, arg, mask), Cout, ts * self.stride, cm)
        if _TRUE_SECOND_ORDER:   # the winners as MaxPoolFn finds them, the values through a differentiable gather
            with torch.no_grad():
                _, arg = MaxPoolFn.apply(x.F.detach(), kids)
            return SparseTensor(x.F.gather(0, arg), Cout, ts * self.stride, cm)
        out, _ = MaxPoolFn.apply(x.F, kids)
        return SparseTensor(out, Cout, ts * self.stride, cm)

    def __repr__(self):
        return "MinkowskiMaxPooling(kernel_size=[2, 2, 2], stride=[2, 2, 2], dilation=[1, 1, 1])"


class AvgPoolFn(Function):
    """MinkowskiLocalPoolingFunction, AVERAGE mode: mean over the children that exist (the divisor is the number of inputs mapped to
    the output).  kids[Kp, M_out] = child rows or -1.  Opaque (first-order) backward like the other native functions.
    Parity unpinned: MinkowskiEngine is absent; only the reference's UNUSED models name this layer (models/MEConv.py:28,35,105,112,126)."""

    @staticmethod
    def forward(ctx, X, kids):
        Kp, Mo = kids.shape
        s = torch.zeros((Mo, X.shape[1]), dtype=X.dtype)
        n = torch.zeros((Mo, 1), dtype=X.dtype)
        for k in range(Kp):
            have = (kids[k] >= 0)[:, None]
            s = s + torch.where(have, X[kids[k].clamp(min=0)], torch.zeros_like(s))
            n = n + have.to(X.dtype)
        ctx.save_for_backward(kids, n)
        ctx.Mi = X.shape[0]
        return s / n.clamp(min=1)

    @staticmethod
    def backward(ctx, g):
        kids, n = ctx.saved_tensors
        with torch.no_grad():
            dX = torch.zeros(ctx.Mi, g.shape[1], dtype=g.dtype)
            gn = g / n.clamp(min=1)
            for k in range(kids.shape[0]):
                o = torch.nonzero(kids[k] >= 0).reshape(-1)
                dX.index_add_(0, kids[k][o], gn[o])
        return dX, None


class GlobalAvgPoolFn(Function):
    """MinkowskiGlobalPoolingFunction, AVERAGE mode: one row per batch index, the mean of its voxels' features (opaque backward)."""

    @staticmethod
    def forward(ctx, X, bidx, B):
        cnt = torch.bincount(bidx, minlength=B).to(X.dtype).clamp_(min=1)[:, None]
        ctx.save_for_backward(bidx, cnt)
        return _seg_sum(X, bidx, B) / cnt

    @staticmethod
    def backward(ctx, g):
        bidx, cnt = ctx.saved_tensors
        with torch.no_grad():
            dX = (g / cnt)[bidx]
        return dX, None, None


class MinkowskiAvgPooling(nn.Module):
    def __init__(self, kernel_size, stride=1, dilation=1, dimension=None):
        super().__init__()
        assert kernel_size == 2 and stride == 2 and dilation == 1 and dimension == 3
        self.kernel_size, self.stride = kernel_size, stride

    def forward(self, x: SparseTensor):
        ts = x.tensor_stride
        cm = x.coordinate_manager
        Cout = cm.get_stride_map(ts, ts * self.stride)
        kids = cm.get_kernel_map(ts, ts * self.stride, self.kernel_size)
        return SparseTensor(AvgPoolFn.apply(x.F, kids), Cout, ts * self.stride, cm)


class _Pooled:
    def __init__(self, F):
        self.F = F


class MinkowskiGlobalAvgPooling(nn.Module):
    def forward(self, x: SparseTensor):
        b = x.C[:, 0].to(torch.int64)
        B = int(b.max().item()) + 1 if b.numel() else 0
        return _Pooled(GlobalAvgPoolFn.apply(x.F, b, B))


class MinkowskiLinear(nn.Module):
    def __init__(self, in_features, out_features, bias=True):
        super().__init__()
        self.linear = nn.Linear(in_features, out_features, bias=bias)

    def forward(self, x):
        return _Pooled(self.linear(x.F))


def _unavailable(name):
    class _U(nn.Module):
        def __init__(self, *a, **k):
            raise NotImplementedError(f"{name} is outside the dd3d hot path (SURVEY.md section 8f row 4)")
    _U.__name__ = name
    return _U


MinkowskiBatchNorm = _unavailable("MinkowskiBatchNorm")


def sparse_collate(coords: Sequence[torch.Tensor], feats: Sequence[torch.Tensor], labels=None, dtype=torch.int32,
                   device=None):
    """ME.utils.sparse_collate (utils/MinkowskiCollate.py:5-10)."""
    N = sum(len(c) for c in coords)
    assert N == sum(len(f) for f in feats)
    D = coords[0].shape[1]
    bcoords = torch.zeros((N, D + 1), dtype=dtype, device=device or coords[0].device)
    s = 0
    for b, coord in enumerate(coords):
        if dtype == torch.int32 and coord.dtype in (torch.float32, torch.float64):
            coord = coord.floor()
        n = coord.shape[0]
        bcoords[s:s + n, 1:] = coord
        bcoords[s:s + n, 0] = b
        s += n
    feats_batch = torch.cat(list(feats), 0)
    if labels is None:
        return bcoords, feats_batch
    return bcoords, feats_batch, torch.cat(list(labels), 0)


def kaiming_normal_(tensor, a=0, mode="fan_in", nonlinearity="leaky_relu"):
    """ME.utils.kaiming_normal_ for a [K, in, out] kernel (models/MEConv.py:237-238)."""
    K, cin, cout = tensor.shape if tensor.dim() == 3 else (1, *tensor.shape)
    fan = cin * K if mode == "fan_in" else cout * K
    gain = nn.init.calculate_gain(nonlinearity, a)
    std = gain / math.sqrt(fan)
    with torch.no_grad():
        return tensor.normal_(0, std)
