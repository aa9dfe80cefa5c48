"""Layer-wise gradient distance -- drop-in for the reference's ``distillation_loss.py`` (match_loss / distance_wb /
distance_mse, distillation_loss.py:26-130), computed by ONE segmented-reduction HIP kernel over all layers plus its backward.

``match_loss(gw_syn, gw_real, dis_metric, device)`` keeps the reference's four-argument signature and its behaviour: the result
is a 0-d tensor differentiable w.r.t. ``gw_syn``; an unknown metric ends the process with ``exit(...)`` (distillation_loss.py:128);
3-D tensors are MinkowskiEngine kernels [K,Cin,Cout] grouped per output channel unless ``settings.Pyt`` is set (:35-40).
"""
from __future__ import annotations

import ctypes

import torch
from torch.autograd import Function

from . import _lib as L
from .configs import settings

_METRICS = {"ours": 0, "modmse": 1, "mse": 2, "cos": 3}


def _rows_cols(shape, pyt: bool):
    """(rows, cols, colmajor) after the reshape of distance_wb / distance_mse (distillation_loss.py:27-47)."""
    n = len(shape)
    numel = 1
    for d in shape:
        numel *= d
    if n in (4, 5):
        return shape[0], numel // shape[0], 0
    if n == 3:
        if not pyt:
            return shape[2], shape[0] * shape[1], 1  # permute(2,1,0): one row per output channel
        return shape[0], shape[1] * shape[2], 0
    if n == 2:
        return shape[0], shape[1], 0
    if n == 1:
        return 1, shape[0], 0
    if n == 0:
        return 1, 1, 0
    raise ValueError("unsupported gradient rank %d" % n)


def _segments(syn, real, dsyn, layout):
    arr = (L.GSeg * len(syn))()
    for i, (s, r) in enumerate(zip(syn, real)):
        rows, cols, cm = layout[i]
        arr[i].syn, arr[i].real = s.data_ptr(), r.data_ptr()
        arr[i].dsyn = dsyn[i].data_ptr() if dsyn is not None and dsyn[i] is not None else None
        arr[i].rows, arr[i].cols, arr[i].colmajor = rows, cols, cm
    return arr


def _prep(t):
    t = t.detach()
    if not t.is_cuda:
        raise RuntimeError("match_loss: gradients must live on the GPU (dd3d_amd has no CPU path)")
    return t.to(torch.float32).contiguous()


class _MatchLossChunk(Function):
    """Up to DD3D_MAX_SEG layers in one launch."""

    @staticmethod
    def forward(ctx, metric, layout, n, *tensors):
        syn = [_prep(t) for t in tensors[:n]]
        real = [_prep(t) for t in tensors[n:]]
        dev = syn[0].device
        segs = _segments(syn, real, None, layout)
        nstats = int(L.lib.dd3d_match_loss_stats_floats(segs, n, 1))
        if nstats < 0:
            raise RuntimeError("match_loss: bad segment description")
        stats = torch.empty(nstats, dtype=torch.float32, device=dev)
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        ws = L.workspace(int(L.lib.dd3d_match_loss_ws_bytes(segs, n, 1)), dev)
        L.call("dd3d_match_loss_fwd", segs, n, 1, metric, L.ptr(loss), L.ptr(stats), L.ptr(ws), ws.numel(), L.stream())
        ctx.saved = (syn, real, stats, layout, metric, n, [t.shape for t in tensors[:n]])
        return loss.reshape(())

    @staticmethod
    def backward(ctx, gl):
        syn, real, stats, layout, metric, n, shapes = ctx.saved
        need = ctx.needs_input_grad[3:3 + n]
        dsyn = [torch.empty_like(s) if need[i] else None for i, s in enumerate(syn)]
        if any(need):
            segs = _segments(syn, real, dsyn, layout)
            glc = gl.detach().to(torch.float32).reshape(1).contiguous()
            ws = L.workspace(256, glc.device)
            L.call("dd3d_match_loss_bwd", segs, n, 1, metric, L.ptr(stats), L.ptr(glc), L.ptr(ws), ws.numel(), L.stream())
        grads = [d.reshape(shapes[i]) if d is not None else None for i, d in enumerate(dsyn)]
        return (None, None, None, *grads, *([None] * n))


def match_loss(gw_syn, gw_real, dis_metric, device):
    """distillation_loss.py:83-130.  ``device`` is accepted for signature compatibility (the result lives where the gradients do)."""
    if dis_metric not in _METRICS:
        exit("unknown distance function: %s" % dis_metric)
    metric = _METRICS[dis_metric]
    pyt = bool(getattr(settings, "Pyt", False))
    gw_syn, gw_real = list(gw_syn), list(gw_real)
    n = len(gw_real)
    layout = [_rows_cols(tuple(g.shape), pyt) for g in gw_real]
    if metric in (2, 3):  # 'mse' / 'cos' flatten everything into one vector: one global row per layer chunk
        layout = [(1, r * c, 0) for (r, c, _) in layout]
    maxseg = 16
    if n <= maxseg:
        return _MatchLossChunk.apply(metric, layout, n, *gw_syn[:n], *gw_real)
    if metric == 3:
        raise NotImplementedError("'cos' over more than %d layers" % maxseg)
    total = None
    for i in range(0, n, maxseg):
        j = min(i + maxseg, n)
        part = _MatchLossChunk.apply(metric, layout[i:j], j - i, *gw_syn[i:j], *gw_real[i:j])
        total = part if total is None else total + part
    return total


def distance_wb(gwr, gws):
    """distillation_loss.py:26-55 for one layer."""
    pyt = bool(getattr(settings, "Pyt", False))
    return _MatchLossChunk.apply(0, [_rows_cols(tuple(gwr.shape), pyt)], 1, gws, gwr)


def distance_mse(gwr, gws):
    """distillation_loss.py:57-81 for one layer."""
    pyt = bool(getattr(settings, "Pyt", False))
    return _MatchLossChunk.apply(1, [_rows_cols(tuple(gwr.shape), pyt)], 1, gws, gwr)
