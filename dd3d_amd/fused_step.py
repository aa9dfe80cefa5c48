"""Class-batched gradient-matching step: the same arithmetic as ``Distiller.outer_step`` (mnist_distill.py:309-374 of the reference),
but ALL classes owned by this process go through every kernel together and nothing runs through torch autograd.

Why it exists (SURVEY.md section 7 "tiny problem, big GPU"): one class is ~7k voxels at the 128-channel layer -- about a hundred
64-row tiles for 256 CUs -- and the reference loop pays hundreds of launches plus three host syncs per class.  Classes are
independent inside a step (mnist_distill.py:335), so a (class, cloud) batch index turns 10 small problems into one large one:
the class of a cloud is its *group*; per-group reductions (kernel gradients, affine gradients, classifier gradients, the matching
loss) come out of the same launches.  The synthetic and the real batch of every class share those launches too (groups g and
G + g of one merged batch, synthetic clouds first), so a step is ONE chain of ~55 launches; in graph mode nothing is read back.

The derivative structure is the reference's: first-order sparse kernels everywhere; the second-order terms
are exactly those that torch autograd produces around MinkowskiEngine's opaque backward passes --
  (i)  d gw[IN weight] / d xhat  = ga (x) G            (affine product, torch-native in the reference),
  (ii) the classifier: gw[W] = dlogits^T dense, gw[b] = sum dlogits, with dlogits = (softmax - onehot)/B differentiable through the
       logits and ``dense`` differentiable through the whole forward graph.
``tests/test_gpu_fused.py`` checks this path against the autograd drop-in path and against the CPU oracle.
"""
from __future__ import annotations

import ctypes
import os
import warnings
import weakref
from typing import List, Optional

import numpy as np
import torch

from . import _lib as L
from .distillation_loss import _METRICS, _rows_cols
from .ops import DensePlacement, _site_strides, site_major, sl_bwd_x_call, sl_fwd_call, transposed_kernel
from .sparse import CloudCoordinateManager, CoordinateManager
from .utils.Mnist2D import batch_from_images

# row bound (= synthetic points) up to which the second-order 128-channel convolution runs offsets-split (dd3d_conv128_split); beyond 512
# tiles the split count is 1 anyway and the plain launch saves the idle summing node
SPLIT_MAX_ROWS = int(os.environ.get("DD3D_CONV_SPLIT_MAX", "16384"))

_F32, _I32 = torch.float32, torch.int32


def _empty(shape, dev, dtype=_F32):
    return torch.empty(shape, dtype=dtype, device=dev)


def _same(a, b):
    return a.cuda_stream == b.cuda_stream


def _wait(a, b):
    """Stream ``a`` waits for everything enqueued on ``b`` so far.  Nothing to do on one and the same stream -- a pass may run with its
    side streams set to its own stream (the pipelined step's synthetic pass), and a stream waiting for its own event inside a capture
    crashed hipStreamEndCapture on ROCm 7.2."""
    if not _same(a, b):
        a.wait_stream(b)


def _wait_event(a, ev, recorded_on):
    if not _same(a, recorded_on):
        a.wait_event(ev)


def conv_kernel_gradient(X, Cin, dY, nbr, ld, M, gro, G, dev, seg=None, B=0, err=None, out=None):
    """dW [G, 27, Cin, 128] of a stride-1 convolution on the current stream (one kernel gradient per group of rows); ``out``: a caller's buffer."""
    dW = out if out is not None else _empty((G, 27, Cin, 128), dev)
    if Cin == 128 and seg is not None and os.environ.get("DD3D_BWW_CLOUDS", "0") != "0":
        # small clouds: whole-cloud windows of both operands in LDS (conv_bww.hip).  Off by default: alone it takes what the row-chunk kernel
        # and its slab sum take (243 + 10 us against 223 + 30 us at cfg2, with a quarter of the memory traffic), but its workgroups fill a
        # CU's registers, so next to the main stream's kernels it starves them (measured: step 1.497 against 1.455 ms)
        ws = L.workspace(int(L.lib.dd3d_conv128_bwd_weight_clouds_ws_bytes(M, G)), dev)
        L.call("dd3d_conv128_bwd_weight_clouds", L.ptr(X), L.ptr(dY), L.ptr(nbr), ld, 27, M, L.ptr(seg), B, L.ptr(gro), G, L.ptr(dW),
               L.ptr(err), L.ptr(ws), ws.numel(), L.stream())
        return dW
    ws = L.workspace(int(L.lib.dd3d_conv_bwd_weight_ws_bytes(M, G, 27, Cin, 128)), dev)
    L.call("dd3d_conv_bwd_weight", L.ptr(X), Cin, L.ptr(dY), 128, L.ptr(nbr), ld, 27, M, L.ptr(gro), G, L.ptr(dW), L.ptr(ws), ws.numel(),
           L.stream())
    return dW


def _cw_site_major():
    """Layout of the per-class classifier-weight gradients inside the step: [G, J, S, 128] (sites outside the channels; default) or the
    parameter's own [G, J, 128, S] (DD3D_CW_LAYOUT=cs).  With the channels innermost ``k_sl_bwd_w`` writes whole 512-byte rows instead of
    4-byte elements S floats apart (cfg4: 562 MB of gradients, 18 ms -> see DESIGN.md), and the second-order pass reads the loss
    gradient through the same strides without a layout launch.  The matching loss only needs the J rows to be contiguous, which both are."""
    return os.environ.get("DD3D_CW_LAYOUT", "sc") != "cs"


def _cw_strides(J, C, S):
    """(group, output, channel, site) strides of that layout, in floats."""
    return (ctypes.c_int64 * 4)(J * C * S, C * S, 1, C) if _cw_site_major() else (ctypes.c_int64 * 4)(J * C * S, C * S, S, 1)


def _group_counters(dev, G):
    """G zero ints for the kernels that elect a group's last workgroup (they leave the counters at zero).  One buffer per OWNER (a
    FusedDistiller keeps its own beside its captured graphs, which hold the address): never shared between distillers, never regrown."""
    return torch.zeros(max(int(G), 1), dtype=_I32, device=dev)


class _Pass:
    """Forward + first-order backward of MEConvExp over ONE class-batched set of clouds -- the synthetic clouds first, then the real
    ones -- and the second-order backward over the synthetic prefix.

    Both batches go through every kernel together: group g < Gs is the synthetic batch of class g, group Gs + g the real batch of the
    same class, so the per-group reductions deliver gw_syn = gw[:Gs] and gw_real = gw[Gs:] from the same launches.  Because the
    synthetic clouds come first, their points, voxel rows (at every stride) and clouds are PREFIXES of the arrays: the second-order
    pass is the same kernels called on the first ``B_syn`` clouds with the prefix's row counts (``seg_off[B_syn]``, on the device)."""

    def __init__(self, net, coords, B, G, cloud_group, group_cloud_off, B_syn, N_syn, side, n_dev=None, static=False, pt_off=None, max_points=0,
                 extent=None, side2=None, stash=None, Gs=None, dl_scale=None, gmin_floor=None, max_per_group=0, flags=None, feats3=None, counters=None):
        """Phase 1 (no host sync): every coordinate map, enqueued on the current stream.
        ``static``: never read a size back -- every launch covers its bound and takes the true row count from the device
        (``n_dev`` = number of valid rows of ``coords``), so the pass can be captured into a HIP graph."""
        dev = coords.device
        self.dev, self.B, self.G, self.net = dev, B, G, net
        # Gs synthetic groups first, then the real ones (G - Gs of them: the same classes unless the real batches of further classes are
        # shared with other ranks, FusedDistiller's second sharding level); dl_scale [B - B_syn, 1]: weight of every real cloud's logit
        # gradient (a rank holding n_part of a class's n clouds contributes sum(CE) / n, not its own mean)
        self.B_syn, self.N_syn, self.Gs, self.dl_scale = B_syn, N_syn, (G // 2 if Gs is None else Gs), dl_scale
        self.max_per_group = max_per_group if max_per_group > 0 else B   # clouds of the largest group (a bound)
        # election counters of dd3d_classifier_head: the owner's buffer (zero between launches), or a fresh one for a stand-alone pass
        self.counters = counters if counters is not None and counters.numel() >= G else _group_counters(dev, G)
        self.cloud_group, self.gco = cloud_group, group_cloud_off
        self.static, self.side, self.side2 = static, side, side2 if side2 is not None else side
        # dense() placement over `extent` sites per axis: the network's dense extent, or the (smaller) box the voxels can actually reach
        self.extent = extent = tuple(net.dense_extent) if extent is None else tuple(extent)
        if pt_off is not None and 0 < max_points <= int(L.lib.dd3d_cloud_maps_max_points()):
            # clouds small enough for LDS: the whole pyramid (+ dense placement) in three launches
            cm = CloudCoordinateManager(coords, pt_off, B, max_points, extent, cloud_group, G, group_cloud_off,
                                        out=None if stash is None else {k: stash[k] for k in ("nbr1", "nbr2", "gro")}, gmin_floor=gmin_floor,
                                        flags=flags, feats3=feats3, F0_out=None if stash is None else stash["F0"])
        else:
            if stash is not None or gmin_floor is not None:
                raise ValueError("the deferred-loss mode and split real batches need the cloud-local map builder (clouds of at most 1024 points)")
            cm = CoordinateManager(coords, B, n_dev=n_dev)
        self.cm, self.stash = cm, stash
        self.N = cm.N
        # every cloud has at most max_points voxel rows at any stride: small enough for the cloud-window kernel gradient?
        self.small_clouds = pt_off is not None and 0 < max_points <= int(L.lib.dd3d_conv128_bwd_weight_clouds_max_rows())
        k0, p0 = cm.kernel_map(1, 1, 3), cm.kernel_map(1, 2, 2)
        k1, p1 = cm.kernel_map(2, 2, 3), cm.kernel_map(2, 4, 2)
        seg1, seg2 = cm.seg_off(1), cm.seg_off(2)
        self.dp = DensePlacement(cm, 4, extent, cloud_group, G, group_cloud_off)
        if getattr(cm, "group_row_off", None) is not None:
            gro1, gro2 = cm.group_row_off[0], cm.group_row_off[1]
        else:
            gidx = group_cloud_off.long()
            gro1, gro2 = seg1[gidx].contiguous(), seg2[gidx].contiguous()   # rows of every group at tensor strides 1 and 2
        self._pre = (k0, p0, k1, p1, seg1, seg2, gro1, gro2)

    def run(self, feats, labels, defer=False, early=None, gw_out=None):
        """Phase 2: (eager mode: ONE device->host read of the row counts, then) forward + first-order backward on the current stream.
        ``early(pass, ev_l2)`` (optional): called once the input gradient of the 128-channel convolution is in the main stream's queue;
        by then the side stream holds the classifier's gradients and the main stream has produced the second level's (event ``ev_l2``),
        i.e. three of the four differentiable entries of gw exist.  FusedDistiller uses it to start the matching loss over them and the
        second-order pass on the side stream, next to the rest of the first-order backward.
        ``gw_out`` (optional): eight caller-owned tensors [G, *parameter shape] (order of network.parameters(), the classifier weight in the
        step's layout) that receive the gradient lists -- buffers that outlive the pass (the pipelined step hands a real pass's lists to
        the NEXT step's synthetic pass).
        ``defer`` (needs a stash): the two kernel gradients of the convolutions, which feed nothing but the loss VALUE, are not launched here;
        their operands are written into the stash, and FusedDistiller runs them next to the beginning of the following step."""
        stash = self.stash
        assert not defer or stash is not None
        net, dev, B, G, cm, dp = self.net, self.dev, self.B, self.G, self.cm, self.dp
        cloud_group, group_cloud_off = self.cloud_group, self.gco
        (k0, ld0), (p0, pl0), (k1, ld1), (p1, pl1), seg1, seg2, gro1, gro2 = self._pre
        st = L.stream()
        if self.static:
            M0, M1, M2 = cm.rows_bound(1), cm.rows_bound(2), cm.rows_bound(4)
            c0, c1, c2 = cm.count_ptr(1), cm.count_ptr(2), cm.count_ptr(4)
        else:
            cm.sync()
            M0, M1, M2 = cm.num_rows(1), cm.num_rows(2), cm.num_rows(4)
            c0 = c1 = c2 = None
        self.M, self.cnt = (M0, M1, M2), (c0, c1, c2)
        self.maps = (k0, ld0, p0, pl0, k1, ld1, p1, pl1, seg1, seg2, gro1, gro2)
        conv0, in0, conv1, in1 = net.features[0], net.features[1], net.features[4], net.features[5]
        W0, W1 = conv0.kernel.detach(), conv1.kernel.detach()
        w1, b1 = in0.weight.detach().reshape(-1), in0.bias.detach().reshape(-1)
        w5, b5 = in1.weight.detach().reshape(-1), in1.bias.detach().reshape(-1)
        cls_w, cls_b = net.classifier.weight, net.classifier.bias.detach()
        J = cls_w.shape[0]
        S = dp.S
        self.J, self.S = J, S
        Wsm = self.Wsm = compact_site_major(cls_w, 128, net.dense_extent, self.extent)
        sst = _site_strides(J, 128, S)
        W1t = self.W1t = transposed_kernel(conv1.kernel)
        self.params = (W0, W1, w1, b1, w5, b5)

        # Launch schedule of the two large 128-channel convolution launches (forward, backward-data): pattern-sorted rows + balanced
        # tile placement (dd3d_conv_schedule; results are bit-identical).  It needs the maps only, so it runs on the side stream next to
        # the first layer; its nodes are ENQUEUED after the first layer's (a HIP graph dispatches its nodes in capture order, and a
        # branch captured first holds back the one captured after it).
        use_sched = M1 >= int(os.environ.get("DD3D_CONV_SCHED_MIN", "8192")) and os.environ.get("DD3D_CONV_SCHED", "1") != "0"
        self.sched = None
        if use_sched:
            ev_maps = torch.cuda.Event()
            ev_maps.record(torch.cuda.current_stream(dev))

        # ---------------- forward ----------------
        feats = feats.detach().contiguous()
        if getattr(cm, "F0", None) is not None:   # the map builder averaged the features per voxel on its way (dd3d_cloud_maps_f0)
            F0 = self.F0 = cm.F0[:M0]
        else:
            F0 = self.F0 = stash["F0"][:M0] if stash is not None else _empty((M0, 3), dev)
            L.call("dd3d_voxel_mean_fwd", L.ptr(feats), self.N, 3, L.ptr(cm.vox_off), L.ptr(cm.vox_pts), M0, c0, L.ptr(F0), st)
        Y0 = _empty((M0, 128), dev)
        L.call("dd3d_conv_fwd", L.ptr(F0), M0, 3, L.ptr(W0), 128, L.ptr(k0), ld0, 27, M0, c0, 0, L.ptr(Y0), st)
        # The normalised rows are never written: the level keeps Y, the per-cloud mean and inv_std, and every later pass normalises a row
        # as it reads it (same bits; one [rows, 128] write less per level).  DD3D_XHAT=1: the stored-xhat launches (measurements).
        self.lean = lean = os.environ.get("DD3D_XHAT", "0") != "1"
        self.Y0, self.mu0, is0 = Y0, (_empty((B, 128), dev) if lean else None), _empty((B, 128), dev)
        self.is0 = is0
        self._xh0 = None if lean else _empty((M0, 128), dev)
        P0, arg0 = self.P0, self.arg0 = (stash["P0"][:M1] if stash is not None else _empty((M1, 128), dev)), _empty((M1, 128), dev, _I32)
        # instance norm + affine + ReLU + max-pool of a level in one launch per cloud (the pooled rows re-read their children from cache)
        L.call("dd3d_inorm_arp_fwd_y" if lean else "dd3d_inorm_arp_fwd", L.ptr(Y0), L.ptr(seg1), B, 128, L.ptr(self.mu0 if lean else self._xh0), L.ptr(is0),
               L.ptr(w1), L.ptr(b1), L.ptr(p0), pl0, 8, L.ptr(seg2), L.ptr(P0), L.ptr(arg0), st)
        Y1 = _empty((M1, 128), dev)
        if use_sched:
            # the 128-row kernel also takes the table in pattern order and the offsets of every tile (the 32-row one gathers through perm)
            tr = int(L.lib.dd3d_conv_schedule_tile_rows(M1))
            _wait_event(self.side, ev_maps, torch.cuda.current_stream(dev))
            with torch.cuda.stream(self.side):
                # The schedule's outputs are ALLOCATED on the side stream too: its kernels wait for the maps only and run beside the first
                # layer, so a block that the main stream's allocator pool had just taken back from a tensor of THIS step could still be
                # in use by a queued main-stream kernel when the side stream writes the schedule into it (measured: intermittent wrong
                # values and, with the 14 MB pattern-ordered table, illegal addresses in the convolution that gathers through it)
                perm1 = _empty((M1,), dev, _I32)
                tmap1 = _empty((int(L.lib.dd3d_conv_schedule_blocks(M1)),), dev, _I32)
                k1s = _empty((27, ld1), dev, _I32) if tr == 128 else None
                tmask1 = _empty(((M1 + tr - 1) // tr,), dev, _I32) if tr == 128 else None
                ws = L.workspace(int(L.lib.dd3d_conv_schedule_ws_bytes(M1)), dev)
                L.call("dd3d_conv_schedule", L.ptr(k1), ld1, 27, M1, c1, L.ptr(perm1), L.ptr(tmap1), L.ptr(k1s), L.ptr(tmask1), L.ptr(ws), ws.numel(),
                       L.stream())
            self.sched = (perm1, tmap1, k1s, tmask1)
            _wait(torch.cuda.current_stream(dev), self.side)
            if not self.static:
                for t in self.sched:   # allocated on the side stream, read by both convolution launches on the main one
                    if t is not None:
                        t.record_stream(torch.cuda.current_stream(dev))
            L.call("dd3d_conv128_ordered", L.ptr(P0), M1, L.ptr(W1), L.ptr(k1), ld1, 27, M1, c1, 0, L.ptr(perm1), L.ptr(tmap1), L.ptr(k1s), L.ptr(tmask1),
                   L.ptr(Y1), st)
        else:
            L.call("dd3d_conv_fwd", L.ptr(P0), M1, 128, L.ptr(W1), 128, L.ptr(k1), ld1, 27, M1, c1, 0, L.ptr(Y1), st)
        self.Y1, self.mu1, is1 = Y1, (_empty((B, 128), dev) if lean else None), _empty((B, 128), dev)
        self.is1 = is1
        self._xh1 = None if lean else _empty((M1, 128), dev)
        P1, arg1 = self.P1, self.arg1 = _empty((M2, 128), dev), _empty((M2, 128), dev, _I32)
        L.call("dd3d_inorm_arp_fwd_y" if lean else "dd3d_inorm_arp_fwd", L.ptr(Y1), L.ptr(seg2), B, 128, L.ptr(self.mu1 if lean else self._xh1), L.ptr(is1),
               L.ptr(w5), L.ptr(b5), L.ptr(p1), pl1, 8, L.ptr(dp.seg_off), L.ptr(P1), L.ptr(arg1), st)
        logits = self.logits = _empty((B, J), dev)
        prob, dl = self.prob, self.dl = _empty((B, J), dev), _empty((B, J), dev)
        closs, self.ce = _empty((B,), dev), _empty((G,), dev)
        # classifier, cross entropy, its logit gradient and the per-group bias gradient: two launches (dd3d_classifier_head)
        go = gw_out if gw_out is not None else [None] * 8
        g_cb = go[7] if go[7] is not None else _empty((G, J), dev)
        ws = L.workspace(int(L.lib.dd3d_sparse_linear_ws_bytes(dp.M_bound, J)), dev)
        L.call("dd3d_classifier_head", L.ptr(P1), L.ptr(dp.site), L.ptr(dp.C), dp.M_bound, L.ptr(dp.seg_off), B, 128, None, L.ptr(Wsm), sst, J,
               L.ptr(cls_b), L.ptr(labels), L.ptr(group_cloud_off), G, int(self.max_per_group), L.ptr(logits), L.ptr(prob), L.ptr(dl), L.ptr(closs),
               L.ptr(self.ce), None if self.dl_scale is not None else L.ptr(g_cb), L.ptr(self.counters), L.ptr(ws), ws.numel(), st)
        if self.dl_scale is not None:   # every first-order gradient below is linear in dl
            dl[self.B_syn:].mul_(self.dl_scale)

        # ---------------- first-order backward: one gradient list per group ----------------
        # The parameter gradients of the classifier and of the 128-channel convolution feed nothing but the matching loss, so they run
        # on a side stream next to the data-gradient chain (sparse linear -> pool/norm -> convolution bwd-data -> pool/norm).
        main, side = torch.cuda.current_stream(dev), self.side

        # With an early second-order start the classifier's gradients and the hook's chain take the THIRD stream (ahead of the first layer's
        # kernel gradient, which waits for the end of the first-order backward anyway), so that the 128-channel kernel gradient on the
        # second stream does not queue behind that chain of small launches
        cls_stream = self.cls_stream = self.side2 if (early is not None and os.environ.get("DD3D_SO_STREAM", "side2") == "side2") else side

        def classifier_gradients():
            side = cls_stream
            _wait(side, main)
            with torch.cuda.stream(side):
                if self.dl_scale is not None:   # (the bias gradient of the SCALED logit gradients)
                    L.call("dd3d_group_sum_rows", L.ptr(dl), L.ptr(group_cloud_off), G, J, L.ptr(g_cb), L.stream())
                g_cw = go[6] if go[6] is not None else _empty((G, J, 128 * S), dev)
                pst = _cw_strides(J, 128, S)
                L.call("dd3d_sparse_linear_bwd_w", L.ptr(dl), L.ptr(P1), L.ptr(dp.vox_at), S, B, L.ptr(group_cloud_off), G, 128, J, L.ptr(g_cw), pst,
                       g_cw.numel(), L.stream())
                self.ev_cls = torch.cuda.Event()
                self.ev_cls.record(side)
            return g_cw

        # The classifier's kernel gradient is bandwidth-bound and nothing needs it before the matching loss: started here it shares the
        # memory system with the norm / pool backward below (both take twice their time); started behind that (DD3D_CLSW_LATE, default)
        # it runs next to the matrix-bound input gradient of the convolution instead.
        cls_late = os.environ.get("DD3D_CLSW_LATE", "1" if early is None else "0") != "0"
        if not cls_late:
            g_cw = classifier_gradients()
        gP1 = _empty((M2, 128), dev)
        sl_bwd_x_call(dl, dp, None, Wsm, sst, J, 0, gP1)
        # the ReLU-masked pooled gradients (ga) are kept for the synthetic clouds only: the second-order pass needs nothing else
        Ns = max(self.N_syn, 1)
        self.ga1, self.ga0 = (_empty((Ns, 128), dev), _empty((Ns, 128), dev)) if self.B_syn > 0 else (None, None)   # (a pass without synthetic clouds keeps none)
        gY1, g_w5, g_b5 = self._arp_inorm_bwd(gP1, arg1, cm.parent(4), self._rows(1), is1, w5, b5, seg2, M1, B, G, self.ga1, self.B_syn, True,
                                              dY_out=None if stash is None else stash["gY1"][:M1], gb_out=go[5] if stash is None else stash["g_b5"],
                                              gw_out=go[4])
        # Two launches want the matrix cores here: the input gradient of the 128-channel convolution (main stream, everything else waits
        # for it) and its kernel gradient (side stream, only the loss value waits for it).  Side by side they share the pipe and BOTH take
        # ~400 us; with the kernel gradient held back until the input gradient is done (DD3D_BWW_AFTER, default) the main stream gets the
        # pipe for ~200 us and the kernel gradient then overlaps the bandwidth-bound norm / pool backward and the second-order pass.
        if cls_late:
            g_cw = classifier_gradients()
        after = os.environ.get("DD3D_BWW_AFTER", "1" if early is None else "0") != "0"
        g_W1 = g_W0 = None
        ev_l2 = None
        if early is not None:   # the second level's gradients exist on the main stream from here on
            ev_l2 = torch.cuda.Event()
            ev_l2.record(main)
        if not after and not defer:
            _wait(side, main)
            with torch.cuda.stream(side):
                g_W1 = self._bww(P0, 128, gY1, k1, ld1, M1, gro2, seg2, out=go[3])
        gP0 = _empty((M1, 128), dev)
        if self.sched is not None:
            L.call("dd3d_conv128_ordered", L.ptr(gY1), M1, L.ptr(W1t), L.ptr(k1), ld1, 27, M1, c1, 1, L.ptr(self.sched[0]), L.ptr(self.sched[1]),
                   L.ptr(self.sched[2]), L.ptr(self.sched[3]), L.ptr(gP0), st)
        else:
            L.call("dd3d_conv_bwd_data", L.ptr(gY1), M1, c1, 128, L.ptr(W1t), 128, L.ptr(k1), ld1, 27, L.ptr(gP0), st)
        # The hook's work goes onto the SIDE stream: behind the classifier's kernel gradient, which that stream computed itself, and in
        # front of the 128-channel kernel gradient below; it runs next to the input gradient above (enqueued first: a HIP graph
        # dispatches in capture order).  Only main <-> side waits, the patterns the capture already has (a third stream waiting for an
        # event of the second crashed hipStreamEndCapture on ROCm 7.2).
        self.early_out = None
        if early is not None:
            self.g_head = (g_w5, g_cw, g_cb)
            self.early_out = early(self, ev_l2)
        late = os.environ.get("DD3D_BWW_LATE", "0") != "0"   # (measurement) the 128-channel kernel gradient behind the norm / pool backward of level 1
        if after and not defer and not late:
            _wait(side, main)
            with torch.cuda.stream(side):
                g_W1 = self._bww(P0, 128, gY1, k1, ld1, M1, gro2, seg2, out=go[3])
        gY0, g_w1, g_b1 = self._arp_inorm_bwd(gP0, arg0, cm.parent(2), self._rows(0), is0, w1, b1, seg1, M0, B, G, self.ga0, self.B_syn, True,
                                              dY_out=None if stash is None else stash["gY0"][:M0], gb_out=go[2] if stash is None else stash["g_b1"],
                                              gw_out=go[1])
        if after and not defer and late:
            _wait(side, main)
            with torch.cuda.stream(side):
                g_W1 = self._bww(P0, 128, gY1, k1, ld1, M1, gro2, seg2, out=go[3])
        # The first layer's kernel gradient feeds nothing but the loss VALUE either: it runs on a stream of its own next to the 128-channel
        # one (bandwidth-bound next to matrix-bound), and the main stream goes on to the matching loss of the differentiable entries and
        # the second-order pass (the caller joins the streams: FusedDistiller._compute)
        side2 = self.side2   # (queued behind the 128-channel kernel gradient on the same stream instead: 1.385 against 1.338 ms)
        if not defer:
            _wait(side2, main)
            with torch.cuda.stream(side2):
                g_W0 = self._bww(F0, 3, gY0, k0, ld0, M0, gro1, out=go[0])
        if not self.static:
            for t in (g_cb, g_cw, g_W1, g_W0):   # allocated on the side stream, consumed on the main one
                if t is not None:
                    t.record_stream(main)
        # order of network.parameters()
        self.gw = [g_W0, g_w1, g_b1, g_W1, g_w5, g_b5, g_cw, g_cb]
        # Operands that a SIDE stream is still reading when this returns (the two kernel gradients run next to the matching loss and the
        # second-order pass): they stay referenced by the pass, i.e. until the next step, so the allocator cannot hand their memory to
        # a main-stream tensor of this step (an intermittent wrong loss VALUE otherwise: 1 in ~60 first steps, measured)
        self._side_operands = (gY1, gY0, gP1, gP0)
        return self

    def _rows(self, level):
        """What the backward passes of a level read: (xhat, None), or (Y, mean) when the normalised rows are not kept."""
        if self.lean:
            return (self.Y0, self.mu0) if level == 0 else (self.Y1, self.mu1)
        return (self._xh0, None) if level == 0 else (self._xh1, None)

    def _xhat(self, level):
        """The normalised rows of a level as a tensor (tests and diagnostics; the step itself never materialises them)."""
        if not self.lean:
            return self._xh0 if level == 0 else self._xh1
        Y, mu, inv = (self.Y0, self.mu0, self.is0) if level == 0 else (self.Y1, self.mu1, self.is1)
        seg = self.maps[8 + level]
        cloud = torch.bucketize(torch.arange(Y.shape[0], device=Y.device, dtype=seg.dtype), seg[1:], right=True).clamp_(max=self.B - 1).long()
        return (Y - mu[cloud]) * inv[cloud]

    xh0 = property(lambda self: self._xhat(0))
    xh1 = property(lambda self: self._xhat(1))

    def _arp_inorm_bwd(self, g_pool, arg, parent, rows, inv_std, w, b, seg, M_in, B, G, ga_out, ga_clouds, want_gw, ga_prev=None, gw_extra=None,
                       dY_out=None, gb_out=None, gw_out=None):
        """Backward of max-pool -> ReLU -> affine -> instance norm in one pass per cloud, over the first B clouds:
        (dY, d weight [G,128], d bias [G,128]); ``ga_out`` receives the masked pooled gradient of the first ``ga_clouds`` clouds.
        ``rows`` = (xhat, None) or (Y, mean): see _rows."""
        dev = self.dev
        xhat, mean = rows
        dY = dY_out if dY_out is not None else _empty((M_in, 128), dev)
        gw_w = (gw_out if gw_out is not None else _empty((G, 128), dev)) if want_gw else None
        gw_b = (gb_out if gb_out is not None else _empty((G, 128), dev)) if want_gw else None
        ws = L.workspace(int(L.lib.dd3d_arp_inorm_bwd_ws_bytes(B)), dev)
        head = (L.ptr(g_pool), L.ptr(arg), L.ptr(parent), L.ptr(xhat)) + ((L.ptr(mean),) if mean is not None else ())
        L.call("dd3d_arp_inorm_bwd_y" if mean is not None else "dd3d_arp_inorm_bwd", *head, L.ptr(inv_std), L.ptr(w), L.ptr(b), 128, L.ptr(seg), B,
               L.ptr(self.cloud_group), L.ptr(self.gco), G, L.ptr(dY), L.ptr(ga_out), int(ga_clouds), L.ptr(ga_prev), L.ptr(gw_extra), L.ptr(gw_w),
               L.ptr(gw_b), L.ptr(ws), ws.numel(), L.stream())
        return dY, gw_w, gw_b

    def _bww(self, X, Cin, dY, nbr, ld, M, gro, seg=None, out=None):
        return conv_kernel_gradient(X, Cin, dY, nbr, ld, M, gro, self.G, self.dev, seg if self.small_clouds else None, self.B, self.cm.err, out=out)



    # ---------------- second-order backward into the synthetic points: the same kernels on the synthetic prefix ----------------
    def backward_to_points(self, G_w1, G_w5, G_cw, G_cb, per):
        """d(sum_g matching loss) / d feats [N_syn,3], given its gradients w.r.t. the four differentiable entries of gw_syn
        (each [Gs, ...]).  ``per`` = clouds per synthetic group."""
        return self.backward_from_conv(self.backward_to_conv(G_w5, G_cw, G_cb, per), G_w1)

    def backward_to_conv(self, G_w5, G_cw, G_cb, per):
        """First part of backward_to_points: classifier, second norm / pool level and the second convolution's input gradient -- everything
        that does not need the first level's gradient G_w1.  Returns hP0 [N_syn,128].  Runs on the current stream."""
        dev, Bs, Gs, J, S, Ns = self.dev, self.B_syn, self.Gs, self.J, self.S, max(self.N_syn, 1)
        k0, ld0, p0, pl0, k1, ld1, p1, pl1, seg1, seg2, gro1, gro2 = self.maps
        W0, W1, w1, b1, w5, b5 = self.params
        dp, cm, st = self.dp, self.cm, L.stream()
        m2s = seg2[Bs:].data_ptr()   # rows of the synthetic clouds at stride 2 (device-side count; its bound is the number of synthetic points)
        sst = _site_strides(J, 128, S)
        if _cw_site_major():   # the loss gradient already has the channels innermost: read through its strides, no layout launch
            Gsm, gst = G_cw, _cw_strides(J, 128, S)
        else:
            Gsm, gst = _empty((Gs, S, J, 128), dev), sst
            L.call("dd3d_classifier_layout", L.ptr(G_cw), Gs, J, 128, S, 0, L.ptr(Gsm), st)
        # u = dL/d(dlogits): through gw[W] = dlogits^T dense and gw[b] = sum dlogits (the latter is added inside softmax_ce_bwd2)
        u = _empty((Bs, J), dev)
        sl_fwd_call(self.P1, dp, self.cloud_group, Gsm, gst, J, None, u, Bs, Ns)
        dz = _empty((Bs, J), dev)
        L.call("dd3d_softmax_ce_bwd2", L.ptr(self.prob), L.ptr(u), L.ptr(G_cb), L.ptr(self.gco), Gs, J, per, L.ptr(dz), st)
        # H = dL/dP1: through dense in gw[W] (weights G_cw, coefficients dlogits) and through the logits (weights W, coefficients dz)
        H = _empty((Ns, 128), dev)
        sl_bwd_x_call(self.dl, dp, self.cloud_group, Gsm, gst, J, 0, H, Bs, Ns)
        sl_bwd_x_call(dz, dp, None, self.Wsm, sst, J, 1, H, Bs, Ns)
        hY1, _, _ = self._arp_inorm_bwd(H, self.arg1, cm.parent(4), self._rows(1), self.is1, w5, b5, seg2, Ns, Bs, Gs, None, 0, False, self.ga1, G_w5)
        hP0 = _empty((Ns, 128), dev)
        if Ns <= SPLIT_MAX_ROWS:  # few tiles: deal every tile's offsets over several CUs
            ws = L.workspace(int(L.lib.dd3d_conv128_split_ws_bytes(Ns)), dev)
            L.call("dd3d_conv128_split", L.ptr(hY1), Ns, L.ptr(self.W1t), L.ptr(k1), ld1, 27, Ns, m2s, 1, L.ptr(hP0), L.ptr(ws), ws.numel(), st)
        else:
            L.call("dd3d_conv_bwd_data", L.ptr(hY1), Ns, m2s, 128, L.ptr(self.W1t), 128, L.ptr(k1), ld1, 27, L.ptr(hP0), st)
        return hP0

    def backward_from_conv(self, hP0, G_w1):
        """Second part of backward_to_points: first norm / pool level (which takes G_w1), first convolution, voxel mean -> dfeats [N_syn,3]."""
        dev, Bs, Gs, Ns = self.dev, self.B_syn, self.Gs, max(self.N_syn, 1)
        k0, ld0, p0, pl0, k1, ld1, p1, pl1, seg1, seg2, gro1, gro2 = self.maps
        W0, W1, w1, b1, w5, b5 = self.params
        cm, st = self.cm, L.stream()
        m1s = seg1[Bs:].data_ptr()   # rows of the synthetic clouds at stride 1
        hY0, _, _ = self._arp_inorm_bwd(hP0, self.arg0, cm.parent(2), self._rows(0), self.is0, w1, b1, seg1, Ns, Bs, Gs, None, 0, False, self.ga0, G_w1)
        hF0 = _empty((Ns, 3), dev)
        L.call("dd3d_conv_bwd_data", L.ptr(hY0), Ns, m1s, 128, L.ptr(W0), 3, L.ptr(k0), ld0, 27, L.ptr(hF0), st)
        dfeats = _empty((Ns, 3), dev)
        L.call("dd3d_voxel_mean_bwd", L.ptr(hF0), Ns, 3, L.ptr(cm.levels[1].inv), L.ptr(cm.vox_off), self.N_syn, L.ptr(dfeats), st)
        return dfeats


def grouped_match_loss(gw_syn: List[torch.Tensor], gw_real: List[torch.Tensor], dis_metric: str, G: int, need_grad, pyt: bool = False,
                       loss_out: Optional[torch.Tensor] = None):
    """distillation_loss.match_loss for G classes at once: tensors are [G, *param_shape]; returns (loss[G], grads list).  With no gradient
    requested only the forward launches run."""
    if dis_metric not in _METRICS:
        exit("unknown distance function: %s" % dis_metric)
    metric = _METRICS[dis_metric]
    n = len(gw_syn)
    dev = gw_syn[0].device
    layout = [_rows_cols(tuple(t.shape[1:]), pyt) for t in gw_real]
    if metric in (2, 3):
        layout = [(1, r * c, 0) for (r, c, _) in layout]
    dsyn = [torch.empty_like(s) if need_grad[i] else None for i, s in enumerate(gw_syn)]
    segs = (L.GSeg * n)()
    for i in range(n):
        segs[i].syn, segs[i].real = gw_syn[i].data_ptr(), gw_real[i].data_ptr()
        segs[i].dsyn = dsyn[i].data_ptr() if dsyn[i] is not None else None
        segs[i].rows, segs[i].cols, segs[i].colmajor = layout[i]
    stats = _empty((int(L.lib.dd3d_match_loss_stats_floats(segs, n, G)),), dev)
    loss = loss_out if loss_out is not None else _empty((G,), dev)
    ws = L.workspace(int(L.lib.dd3d_match_loss_ws_bytes(segs, n, G)), dev)
    if any(need_grad):   # (one launch for small lists, the forward's two and the backward's one otherwise: the same bits)
        L.call("dd3d_match_loss_fwd_bwd", segs, n, G, metric, L.ptr(loss), L.ptr(stats), None, L.ptr(ws), ws.numel(), L.stream())
    else:
        L.call("dd3d_match_loss_fwd", segs, n, G, metric, L.ptr(loss), L.ptr(stats), L.ptr(ws), ws.numel(), L.stream())
    return loss, dsyn


def reachable_sites(full, compact):
    """Flat indices (in the network's dense extent ``full``) of the sites of the corner box ``compact``, in the box's own row-major order."""
    a, d, e = (torch.arange(n) for n in compact)
    return ((a[:, None, None] * full[1] + d[None, :, None]) * full[2] + e[None, None, :]).reshape(-1)


def compact_site_major(cls_w: torch.Tensor, C: int, full, compact) -> torch.Tensor:
    """Site-major classifier operand [Sc, J, C] over the reachable sites only.  ``dense()`` places a voxel at (coordinate - batch minimum)
    / 4 inside a [E0,E1,E2] box (MEConv.py:250-254); clouds clamped to [0, 27] reach 7 x 7 of the 28 x 28 sites, so 94 % of the
    100 352 classifier columns only ever meet zeros.  Cached on the (frozen) weight, per box."""
    full, compact = tuple(full), tuple(compact)
    if compact == full:
        return site_major(cls_w, C, full[0] * full[1] * full[2])
    hit = getattr(cls_w, "_dd3d_csm", None)
    if hit is not None and hit[0] == (cls_w._version, compact) and hit[1].device == cls_w.device:
        return hit[1]
    J = cls_w.shape[0]
    idx = reachable_sites(full, compact).to(cls_w.device)
    wc = cls_w.detach().reshape(J, C, -1)[:, :, idx].reshape(J, -1).contiguous()
    out = site_major(wc, C, idx.numel())
    try:
        cls_w._dd3d_csm = ((cls_w._version, compact), out)
    except Exception:
        pass
    return out


class _GradList(list):
    """Per-parameter gradient list [G, *shape] whose classifier-weight entry is held over the reachable sites only; indexing it
    returns the parameter-shaped tensor (zeros elsewhere), expanded on every access: in graph mode the list wraps fixed buffers that
    the next replay overwrites, so nothing derived from them may be cached.  The step itself never expands it."""

    def __init__(self, tensors, full, compact, C=128):
        super().__init__(tensors)
        self._full, self._compact, self._C, self._same = tuple(full), tuple(compact), C, tuple(full) == tuple(compact)
        self._site_major = _cw_site_major()   # the step holds the entry as [G, J, S, C] (see _cw_site_major)

    def compact(self, i):
        return list.__getitem__(self, i)

    def __getitem__(self, i):
        if i in (6, -2) and (not self._same or self._site_major):
            t = list.__getitem__(self, 6)
            G, J = t.shape[0], t.shape[1]
            S = self._full[0] * self._full[1] * self._full[2]
            t = t.reshape(G, J, -1, self._C).permute(0, 1, 3, 2) if self._site_major else t.reshape(G, J, self._C, -1)   # -> [G, J, C, sites held]
            if self._same:
                return t.reshape(G, J, self._C * S)
            out = torch.zeros((G, J, self._C, S), dtype=t.dtype, device=t.device)
            out[..., reachable_sites(self._full, self._compact).to(t.device)] = t
            return out.reshape(G, J, self._C * S)
        return list.__getitem__(self, i)

    def __iter__(self):
        return (self[i] for i in range(len(self)))


class _SgdState:
    """``optimizer_img`` of the reference loop (mnist_distill.py:303: SGD([image_syn], lr, momentum=0.5)) as the fused step keeps it -- a
    momentum buffer updated inside dd3d_sgd_scatter -- with torch.optim.SGD's ``state_dict`` layout, so checkpoints written from the fused
    path carry the reference's ``optimizer_state_dict`` (mnist_distill.py:394) and load into a torch SGD in either code base."""

    def __init__(self, owner):
        self._o = weakref.proxy(owner)   # no reference cycle: a dropped distiller is freed by its reference count, where it is dropped

    def state_dict(self):
        o = self._o
        ref = torch.optim.SGD([torch.zeros(1, requires_grad=True)], lr=o.lr, momentum=o.momentum).state_dict()   # this torch's group keys
        state = {0: {"momentum_buffer": o.momentum_buf.clone()}} if o.steps_done > 0 else {}
        return {"state": state, "param_groups": ref["param_groups"]}

    def load_state_dict(self, sd):
        o = self._o
        g = sd["param_groups"][0]
        o.lr, o.momentum = float(g["lr"]), float(g["momentum"])
        buf = sd.get("state", {}).get(0, {}).get("momentum_buffer")
        if buf is None:
            o.momentum_buf.zero_()
        else:
            o.momentum_buf.copy_(buf.to(o.momentum_buf.device))
        if o._graph is not None:
            raise RuntimeError("load the optimizer state before the first captured step (lr / momentum are baked into the graph)")


class FusedDistiller:
    """Same constructor and state as ``mnist_distill.Distiller`` (MEConvExp only); ``outer_step`` runs the class-batched path.

    ``graph=True`` (default): the whole step -- real-batch assembly from a point pool resident in HBM, both passes, the matching loss,
    the backward into the synthetic points and the SGD update -- is captured ONCE into a HIP graph and replayed; per step the host
    draws the sample indices (``np.random.permutation`` per class, as mnist_distill.py:200-203 does), uploads them and launches the
    graph.  Every launch inside covers a bound and reads the true row counts from device memory, so batches of different sizes
    replay through the same graph.  ``graph=False``: the same kernels enqueued eagerly with exact sizes (one size read-back per pass).
    Both modes produce bit-identical results (``tests/test_gpu_fused.py``)."""

    def __init__(self, images_all, indices_class, network, ipc, num_classes=10, num_points=250, batch_size=256, dist_opt="modmse",
                 dist_lr=0.1, device=None, seed=0, image_syn=None, classes=None, momentum=0.5, graph=True, real_clouds=None,
                 quantization_size=1.0, clamp_range=(0.0, 27.0), zero_first_axis=True, defer_loss=False, real_classes=None, real_part=(0, 1),
                 exchange=None, pipeline=None):
        """``real_classes`` / ``real_part`` / ``exchange``: the second sharding level.  This process computes the
        real-batch gradients of ``real_classes`` (default: its own ``classes``; must contain them) over part ``real_part = (p, P)`` of
        every sampled batch -- clouds p, p + P, ... of the SAME permutation every process of the group draws (same ``seed``) --
        weighted n_part / n so that the parts add up to the whole batch's gw_real (instance norm is per cloud, the cross entropy a mean
        over clouds, dense() is placed by the whole sample's minimum corner).  ``exchange(flat)``: sums the flat fp32 buffer of those
        partial gradients over the processes that share the classes, in place (``sharding.all_reduce_in(group)``); called between
        the first-order backward and the matching loss.  A split step runs with eager launches (``graph`` is ignored).  The synthetic
        clouds, the loss, the second-order pass and the update stay with ``classes``.
        ``real_clouds`` (optional, instead of ``images_all``): the real data as point clouds -- a float tensor [n, P, 3] or a list of
        [n_i, 3] tensors; coordinates = features = the points (what Mnist2Dreal does with pixels), coordinates divided by
        ``quantization_size`` (create_input_batch).  ``clamp_range`` / ``zero_first_axis``: the per-iteration clamp of the synthetic
        points (mnist_distill.py:311-317: [0, 27], depth := 0); ``clamp_range=None`` switches it off for 3-D clouds."""
        self.device = dev = torch.device(device or "cuda")
        self.images_all = images_all.to(dev) if images_all is not None else None
        self.quant, self.clamp_range, self.zero_first_axis = float(quantization_size), clamp_range, bool(zero_first_axis)
        self._pool = None
        if real_clouds is not None:   # point pool straight from the clouds (first num_points points of each)
            cl = [c[:num_points] for c in real_clouds]
            cnt = torch.tensor([0] + [int(c.shape[0]) for c in cl])
            self._pool = (torch.cat([c.reshape(-1, 3) for c in cl]).to(dev, _F32).contiguous(), torch.cumsum(cnt, 0).to(dev, _I32))
        self.indices_class = [np.asarray(ix) for ix in indices_class]
        self.network = network.to(dev).train()
        self.num_classes, self.ipc, self.num_points, self.batch_size = num_classes, ipc, num_points, batch_size
        self.dist_opt, self.rng = dist_opt, np.random.RandomState(seed)
        self.classes = list(range(num_classes)) if classes is None else list(classes)
        if image_syn is None:
            g = torch.Generator().manual_seed(seed)
            image_syn = 28 * torch.rand((num_classes * ipc, num_points, 3), generator=g, dtype=torch.float)
        self.image_syn = image_syn.to(dev).clone().contiguous()
        self.momentum_buf = torch.zeros_like(self.image_syn)
        self.label_syn = torch.arange(num_classes, device=dev).repeat_interleave(ipc)   # mnist_distill.py:297
        self.optimizer_img = _SgdState(self)
        self.lr, self.momentum, self.steps_done = dist_lr, momentum, 0
        G = self.G = len(self.classes)
        self.real_classes = list(self.classes) if real_classes is None else list(real_classes)
        self.real_part = (int(real_part[0]), int(real_part[1]))
        if not (0 <= self.real_part[0] < self.real_part[1]) or any(c not in self.real_classes for c in self.classes):
            raise ValueError("real_part = (p, P) with 0 <= p < P, and real_classes must contain classes")
        self.exchange = exchange
        self.split = self.real_part[1] > 1 or self.real_classes != self.classes
        if self.split and exchange is None:
            raise ValueError("a split real batch needs exchange(flat) to sum the partial gradient lists")
        self.Gr = len(self.real_classes)
        self.own_idx = torch.tensor([self.real_classes.index(c) for c in self.classes], device=dev, dtype=torch.long)
        self._sizes_full = [min(batch_size, len(self.indices_class[c])) for c in self.real_classes]
        self._sizes = [len(range(self.real_part[0], n, self.real_part[1])) for n in self._sizes_full]
        self.dl_scale = None
        if self.split:   # weight of every real cloud's logit gradient: n_part / n_whole of its class
            w = [a / b if b else 0.0 for a, b in zip(self._sizes, self._sizes_full)]
            self.dl_scale = torch.tensor(np.repeat(w, self._sizes), device=dev, dtype=_F32)[:, None].contiguous()
        self._corners = self._floor_np = None
        cls = torch.tensor(self.classes, device=dev)
        # rows of image_syn owned by this process, class-major
        self.syn_rows = (cls[:, None] * ipc + torch.arange(ipc, device=dev)[None, :]).reshape(-1)
        # position of every synthetic cloud among the owned ones (-1: another rank's), for the one-launch prologue / epilogue
        self.dest_cloud = torch.full((num_classes * ipc,), -1, dtype=_I32, device=dev)
        self.dest_cloud[self.syn_rows] = torch.arange(G * ipc, dtype=_I32, device=dev)
        self.syn_labels = cls.repeat_interleave(ipc).to(_I32)
        self.syn_group = torch.arange(G, device=dev, dtype=_I32).repeat_interleave(ipc)
        self.syn_gco = (torch.arange(G + 1, device=dev) * ipc).to(_I32)
        self.syn_pt_off = (torch.arange(G * ipc + 1, device=dev) * num_points).to(_I32)
        self.syn_batch_col = torch.arange(G * ipc, device=dev, dtype=_F32).repeat_interleave(num_points)[:, None]
        self.last = {}
        self.stream_side = torch.cuda.Stream(dev)   # parameter-gradient kernels that only the loss consumes
        # the first layer's kernel gradient, next to the 128-channel one.  Deferred-loss mode: all of the previous step's loss-value work
        # (a capture with a fourth stream crashed hipStreamEndCapture on ROCm 7.2, so it shares this one)
        self.stream_side2 = torch.cuda.Stream(dev)
        self.stream_defer = self.stream_side2
        # eager mode: next step's real batch is sampled on its own stream while this step's kernels run
        self.stream_data, self._prefetched = torch.cuda.Stream(dev), None
        # A split real batch runs with eager launches: the variant with two graph captures around the exchange (in one memory pool) passed
        # its tests alone but ended in illegal addresses inside long test sequences -- withdrawn until that is understood
        self.use_graph = bool(graph) and G > 0 and not self.split
        # Deferred loss value (graph mode, layer-additive metrics): the two kernel gradients of the convolutions and the loss terms over them
        # feed nothing but the VALUE of the matching loss -- not the update of the synthetic points.  With ``defer_loss`` a step leaves
        # their operands in a stash (two sets, alternating) and the NEXT step's graph computes them on side streams next to its own
        # map building and first layer, where the matrix cores are idle.  ``outer_step`` then returns the complete loss of the PREVIOUS
        # step (zero on the first one); ``flush()`` completes and returns the last step's.  Same kernels, same bits, one step later.
        self.defer = bool(defer_loss) and self.use_graph and dist_opt in ("ours", "modmse", "mse") and not self.split
        self._stash, self._pending, self._graphs = None, None, [None, None]
        self._graph = self._static = None
        self._graph_stamp, self._needs_eager, self._last_loss = None, False, None
        self._scope = "graph%x" % id(self)
        # Pipelined step (``pipeline=True`` / DD3D_PIPELINE=1; see _pipe_body): the real and the synthetic batch as TWO passes, the real pass
        # of step t+1 beside the synthetic pass of step t
        self.pipeline = ((os.environ.get("DD3D_PIPELINE", "0") != "0") if pipeline is None else bool(pipeline)) and G > 0 and not self.split and not self.defer
        self._pipe = None
        self.prefetch = True   # eager mode: sample the next step's real batch right behind this step (ChunkedDistiller: off, the chunks share one sample stream)
        self.so_early = os.environ.get("DD3D_SO_EARLY", "1") != "0"   # second-order pass started next to the first-order tail (_compute_gen)
        self._counters = _group_counters(dev, G + self.Gr)   # this distiller's own (its graphs keep the address)
        self._closed = False

    def _param_stamp(self):
        """(address, version) of every network parameter: what the captured graph and its cached weight layouts were built from."""
        return tuple((p.data_ptr(), p._version) for p in self.network.parameters())

    def invalidate(self):
        """Drop the captured graph and the cached weight layouts (transposed 128-channel kernel, site-major classifier weight).  Called
        automatically when a parameter's address or version counter changed since the capture; call it yourself after editing weights
        through ``.data`` (which bumps no version counter).  The next step runs eagerly and rebuilds the layouts, the one after re-captures."""
        had_graph = self._graph is not None or any(g is not None for g in self._graphs)
        self._graph, self._graphs = None, [None, None]
        self._needs_eager = True
        if had_graph:
            # the dropped graphs' workspaces go with them, once nothing on the device can still be running a replay: a block that a captured
            # graph was given is never regrown (_lib.workspace refuses), so the next capture must start from a clean scope
            torch.cuda.synchronize(self.device)
            L.drop_scope(self._scope)
        for p in self.network.parameters():
            for attr in ("_dd3d_csm", "_dd3d_wt", "_dd3d_sm"):
                if hasattr(p, attr):
                    try:
                        delattr(p, attr)
                    except Exception:
                        pass

    def _real_range(self):
        """(lo[3], hi[3]) of the real points before quantisation (one read-back, cached)."""
        if getattr(self, "_rrange", None) is None:
            if self.images_all is not None:
                H, W = int(self.images_all.shape[-2]), int(self.images_all.shape[-1])
                self._rrange = ([0.0, 0.0, 0.0], [0.0, H - 1.0, W - 1.0])   # Mnist2Dreal: points (0, y, x) of the on-pixels
            else:
                xyz = self._point_pool()[0]
                self._rrange = (xyz.min(0).values.tolist(), xyz.max(0).values.tolist())
        return self._rrange

    def site_extent(self, real_batch=None):
        """Sites per axis that a stride-4 voxel of this job can reach inside the network's dense extent: index = (coordinate - batch
        minimum) / 4 <= floor(hi / 4) - floor(lo / 4) with [lo, hi] the floored coordinate range of the clamped synthetic points and
        of the real data.  Without a clamp the full extent is kept."""
        E = tuple(self.network.dense_extent)
        if self.clamp_range is None:
            return E
        q = np.float32(self.quant)
        slo, shi = [self.clamp_range[0]] * 3, [self.clamp_range[1]] * 3
        if self.zero_first_axis:
            slo[0] = shi[0] = 0.0
        if real_batch is not None:   # an explicit batch brings its own (already quantised) coordinates
            c = real_batch["coordinates"][:, 1:]
            rlo, rhi = (np.floor(v) for v in (c.min(0).values.cpu().numpy(), c.max(0).values.cpu().numpy())) if c.shape[0] else ([0] * 3, [0] * 3)
        else:
            rlo, rhi = (np.floor(np.float32(v) / q) for v in (np.float32(self._real_range()[0]), np.float32(self._real_range()[1])))
        out = []
        for a in range(3):
            lo = min(int(np.floor(np.float32(slo[a]) / q)), int(rlo[a]))
            hi = max(int(np.floor(np.float32(shi[a]) / q)), int(rhi[a]))
            out.append(max(1, min(E[a], hi // 4 - lo // 4 + 1)))
        return tuple(out)

    def close(self):
        """Release everything the device may still be using, in a safe order: wait for the distiller's streams (a replay or an eager step
        may still be running on up to four of them), THEN drop the captured graphs and every buffer they hold addresses of, and last the
        workspace scopes.  Idempotent; called by ``__del__``.  Any HIP error pending before or raised by the teardown is reported as a
        warning that names it, so a failing teardown says what failed instead of aborting the process from a destructor."""
        if getattr(self, "_closed", True):   # (also: a constructor that raised before the streams existed)
            return
        self._closed = True
        before = L.hip_last_error()
        try:
            for s in (self.stream_side, self.stream_side2, self.stream_data):
                s.synchronize()
            torch.cuda.synchronize(self.device)
        except Exception as e:   # never raise out of a destructor
            warnings.warn("FusedDistiller.close(): synchronisation failed: %r" % (e,))
        self._graph, self._graphs, self._pending, self._stash, self._prefetched = None, [None, None], None, None, None
        self.last = {}
        self._static = self._pipe = None
        L.drop_scope(self._scope)
        after = L.hip_last_error()
        if before or after:
            warnings.warn("FusedDistiller.close(): HIP error pending before teardown: %s; after: %s" % (before or "none", after or "none"))

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def clamp(self):
        s = self.image_syn
        if self.clamp_range is not None:
            s.clamp_(self.clamp_range[0], self.clamp_range[1])
        if self.zero_first_axis:
            s[:, :, 0] = 0

    def _point_pool(self):
        """(pool_xyz [total,3], pool_off [n+1]): every real cloud's points (its first num_points), resident in HBM, built once."""
        if self._pool is None:
            dev, npts = self.device, self.num_points
            n_img = int(self.images_all.shape[0])
            off, xyz, total = [torch.zeros(1, dtype=torch.int64, device=dev)], [], 0
            for a in range(0, n_img, 8192):           # chunks keep the one-time nonzero() temporaries small
                imgs = self.images_all[a:a + 8192]
                b = batch_from_images(imgs, torch.zeros(imgs.shape[0], dtype=torch.int64, device=dev), npts)
                cnt = torch.bincount(b["coordinates"][:, 0].long(), minlength=imgs.shape[0])
                off.append(torch.cumsum(cnt, 0) + total)
                total += int(cnt.sum().item())
                xyz.append(b["features"])
            self._pool = (torch.cat(xyz).contiguous() if total else torch.zeros((1, 3), device=dev), torch.cat(off).to(_I32).contiguous())
        return self._pool

    # ------------------------------------------------------------------ eager-mode real batch (host-sized tensors)
    def _corner_table(self):
        """[n_pool, 3] int32 (host): minimum stride-4 voxel corner of every cloud of the pool, as dd3d_cloud_maps computes it from
        coordinates = points / quantization_size (fp32).  One pass over the pool, once."""
        if self._corners is None:
            xyz, off = self._point_pool()
            n, dev = int(off.numel()) - 1, self.device
            cnt = (off[1:] - off[:-1]).long()
            seg = torch.repeat_interleave(torch.arange(n, device=dev), cnt)
            vox = torch.floor(L.true_divide(xyz[:int(off[-1])], self.quant))
            mn = torch.full((n, 3), float("inf"), device=dev).scatter_reduce_(0, seg[:, None].expand(-1, 3), vox, "amin")
            c = torch.floor(mn / 4) * 4
            c[torch.isinf(c)] = float(0x7F7F7F7F)   # a cloud without points has no corner
            self._corners = c.to(torch.int64).to(torch.int32).cpu().numpy()
        return self._corners

    def _draw(self):
        """mnist_distill.py:200-203 for every class whose real batch this process works on: np.random.permutation(indices_class[c])
        [:batch_size], and of it the clouds p, p + P, ... (``real_part``; all of it by default).  With a split batch also the minimum
        corner of the WHOLE sample per class -> self._floor_np [G + Gr, 3] (no floor for the synthetic groups)."""
        p, P = self.real_part
        ids, floor = [], [np.full((self.G, 3), 0x7F7F7F7F, dtype=np.int32)]
        for c in self.real_classes:
            full = self.rng.permutation(self.indices_class[c])[:self.batch_size]
            ids.append(full[p::P])
            if self.split:
                floor.append(self._corner_table()[full].min(0, keepdims=True) if len(full) else np.full((1, 3), 0x7F7F7F7F, dtype=np.int32))
        self._floor_np = np.concatenate(floor).astype(np.int32) if self.split else None
        return np.concatenate(ids)

    def sample_real(self):
        """mnist_distill.py:200-208 for every owned class at once: one gather + one nonzero."""
        idx = self._draw()
        sizes = self._sizes
        if self.images_all is None:
            b = self.real_batch_from_pool(idx, sizes)
        else:
            b = self.real_batch_from_images(self.images_all[torch.as_tensor(idx, device=self.device)], sizes)
        if self.split:
            b["gmin_floor"] = torch.from_numpy(self._floor_np).to(self.device)
        return b

    def real_batch_from_pool(self, idx, sizes):
        """The collated batch of the sampled clouds ``idx`` (coordinates = points / quantization_size, features = points)."""
        dev = self.device
        xyz, off = self._point_pool()
        ids = torch.as_tensor(np.asarray(idx), device=dev).long()
        start, cnt = off[ids].long(), (off[ids + 1] - off[ids]).long()
        pt = torch.cat([cnt.new_zeros(1), torch.cumsum(cnt, 0)])
        b = torch.repeat_interleave(torch.arange(ids.numel(), device=dev), cnt)
        rows = start[b] + (torch.arange(int(pt[-1]), device=dev) - pt[b])
        p = xyz[rows]
        return {"coordinates": torch.cat([b[:, None].to(_F32), L.true_divide(p, self.quant)], 1), "features": p.clone(), "num_clouds": int(ids.numel()),
                "labels": torch.tensor(np.repeat(self.real_classes, sizes), device=dev), "pt_off": pt.to(_I32),
                "cloud_group": torch.tensor(np.repeat(np.arange(self.Gr), sizes), device=dev, dtype=_I32),
                "group_cloud_off": torch.tensor(np.concatenate([[0], np.cumsum(sizes)]), device=dev, dtype=_I32), "max_per": int(max(sizes))}

    def real_batch_from_images(self, images, sizes):
        dev = self.device
        labels = torch.tensor(np.repeat(self.real_classes, sizes), device=dev)
        b = batch_from_images(images, labels, self.num_points)
        if self.quant != 1.0:
            b["coordinates"][:, 1:] = L.true_divide(b["coordinates"][:, 1:], self.quant)
        b["cloud_group"] = torch.tensor(np.repeat(np.arange(self.Gr), sizes), device=dev, dtype=_I32)
        b["group_cloud_off"] = torch.tensor(np.concatenate([[0], np.cumsum(sizes)]), device=dev, dtype=_I32)
        b["max_per"] = int(max(sizes))
        return b

    # ------------------------------------------------------------------ static (graph) mode: point pool + fixed-address buffers
    def _build_static(self):
        """Point pool of the whole image set (CSR, resident in HBM) and the fixed-address buffers of the captured step."""
        dev, npts = self.device, self.num_points
        st = type("Static", (), {})()
        st.pool_xyz, st.pool_off = self._point_pool()
        st.n_pool = int(st.pool_off.numel()) - 1
        counts = (st.pool_off[1:] - st.pool_off[:-1]).cpu().numpy()
        sizes = self._sizes
        # tight bound on the points of one sampled batch: the `size` largest clouds of every class
        bound = sum(int(np.sort(counts[self.indices_class[c]])[::-1][:n].sum()) for c, n in zip(self.real_classes, sizes))
        st.bound = max(256, (bound + 255) // 256 * 256)
        st.sizes, st.B = sizes, int(sum(sizes))
        # merged batch: the synthetic clouds first (fixed sizes), then the sampled real clouds
        Bs, Ns, G = self.G * self.ipc, self.G * self.ipc * npts, self.G
        st.bound_total = Ns + st.bound
        st.ids = torch.zeros(max(st.B, 1), dtype=_I32, device=dev)
        st.coords = torch.zeros((st.bound_total, 4), dtype=_F32, device=dev)
        st.coords[:Ns, 0] = self.syn_batch_col[:, 0]
        st.feats = torch.zeros((st.bound_total, 3), dtype=_F32, device=dev)
        st.pt_off = torch.zeros(Bs + st.B + 1, dtype=_I32, device=dev)
        st.pt_off[:Bs + 1] = self.syn_pt_off
        st.n = torch.zeros(1, dtype=_I32, device=dev)
        st.err = torch.zeros(1, dtype=_I32, device=dev)
        st.flags = torch.zeros(CloudCoordinateManager.MAX_LEVELS + 1, dtype=_I32, device=dev)   # the map builder's count / error words (cleared by dd3d_assemble_batch)
        rl = torch.tensor(np.repeat(self.real_classes, sizes), device=dev, dtype=_I32)
        rg = torch.tensor(np.repeat(np.arange(self.Gr), sizes), device=dev, dtype=_I32)
        rgco = torch.tensor(np.cumsum(sizes), device=dev, dtype=_I32)
        st.labels = torch.cat([self.syn_labels, rl])
        st.cloud_group = torch.cat([self.syn_group, rg + G])
        st.gco = torch.cat([self.syn_gco, rgco + Bs])
        st.max_per = int(max(sizes))
        if self.defer:
            bt, G2 = st.bound_total, 2 * G
            self._stash = [{"F0": _empty((bt, 3), dev), "P0": _empty((bt, 128), dev), "gY1": _empty((bt, 128), dev), "gY0": _empty((bt, 128), dev),
                            "nbr1": _empty((27, bt), dev, _I32), "nbr2": _empty((27, bt), dev, _I32), "gro": _empty((3, G2 + 1), dev, _I32),
                            "g_b1": _empty((G2, 128), dev), "g_b5": _empty((G2, 128), dev), "loss_a": torch.zeros(G, device=dev),
                            "loss": torch.zeros(G, device=dev), "ld": bt, "parity": i} for i in range(2)]
        st.gmin_floor = torch.full((G + self.Gr, 3), 0x7F7F7F7F, dtype=_I32, device=dev) if self.split else None
        st.pin_floor = [torch.empty((G + self.Gr, 3), dtype=_I32).pin_memory() for _ in range(4)] if self.split else None
        st.pin = [torch.empty(max(st.B, 1), dtype=_I32).pin_memory() for _ in range(4)]
        st.pin_ev = [torch.cuda.Event() for _ in range(4)]
        st.pin_i = 0
        self._static = st

    def _upload_ids(self):
        """Host side of a graph step: draw the sample, stage it in pinned memory, enqueue the (tiny) upload."""
        st = self._static
        i = st.pin_i
        st.pin_i = (i + 1) % len(st.pin)
        st.pin_ev[i].synchronize()                # the upload that last used this staging buffer has finished
        st.pin[i].numpy()[:st.B] = self._draw()
        st.ids.copy_(st.pin[i], non_blocking=True)
        if self.split:
            st.pin_floor[i].numpy()[:] = self._floor_np
            st.gmin_floor.copy_(st.pin_floor[i], non_blocking=True)
        st.pin_ev[i].record()


    # ------------------------------------------------------------------ pipelined step: two passes, the real one a step ahead
    def _build_pipe(self):
        """Fixed-address buffers of the pipelined step: the real batch and the synthetic batch as separate batches, and TWO sets of the
        real batches' gradient lists (a real pass writes one while the synthetic pass of the step before reads the other)."""
        if self._static is None:
            self._build_static()
        st, dev, G, ipc, npts = self._static, self.device, self.G, self.ipc, self.num_points
        Bs, Ns = G * ipc, G * ipc * npts
        pp = type("Pipe", (), {})()
        i32 = lambda *shape: torch.zeros(shape, dtype=_I32, device=dev)
        pp.rcoords, pp.rfeats = torch.zeros((st.bound, 4), device=dev), torch.zeros((st.bound, 3), device=dev)
        pp.rpt_off, pp.rn, pp.rerr = i32(st.B + 1), i32(1), i32(1)
        pp.rlabels = torch.tensor(np.repeat(self.real_classes, st.sizes), device=dev, dtype=_I32)
        pp.rgroup = torch.tensor(np.repeat(np.arange(self.Gr), st.sizes), device=dev, dtype=_I32)
        pp.rgco = torch.tensor(np.concatenate([[0], np.cumsum(st.sizes)]), device=dev, dtype=_I32)
        pp.scoords, pp.sfeats = torch.zeros((Ns, 4), device=dev), torch.zeros((Ns, 3), device=dev)
        pp.scoords[:, 0] = self.syn_batch_col[:, 0]
        extent = self.site_extent()
        J, S = self.network.classifier.weight.shape[0], extent[0] * extent[1] * extent[2]
        shapes = [(27, 3, 128), (128,), (128,), (27, 128, 128), (128,), (128,), (J, 128 * S), (J,)]
        pp.gwr = [[torch.zeros((self.Gr,) + sh, device=dev) for sh in shapes] for _ in range(2)]
        pp.counters_r, pp.counters_s = _group_counters(dev, self.Gr), _group_counters(dev, G)
        pp.extent, pp.graphs, pp.steps = extent, [None, None], 0
        self._pipe = pp

    def _pipe_gather(self):
        st, pp = self._static, self._pipe
        L.call("dd3d_gather_clouds", L.ptr(st.pool_xyz), L.ptr(st.pool_off), st.n_pool, L.ptr(st.ids), st.B, self.num_points, st.bound, 0, 0, self.quant,
               L.ptr(pp.rcoords), L.ptr(pp.rfeats), L.ptr(pp.rpt_off), L.ptr(pp.rn), L.ptr(st.err), L.stream())

    def _pipe_real_pass(self, out_set):
        """The real batches that _pipe_gather assembled (the sample whose ids are in the static id buffer): maps, forward, first-order
        backward on the CURRENT stream (+ the side stream for the parameter gradients that only the loss reads); their eight gradient
        lists go into set ``out_set``."""
        st, pp, dev = self._static, self._pipe, self.device
        pr = _Pass(self.network, pp.rcoords, st.B, self.Gr, pp.rgroup, pp.rgco, 0, 0, self.stream_side, pp.rn, True, pp.rpt_off, self.num_points, pp.extent,
                   self.stream_side, None, Gs=0, max_per_group=st.max_per, counters=pp.counters_r, feats3=pp.rfeats)
        pr.run(pp.rfeats, pp.rlabels, gw_out=pp.gwr[out_set])
        torch.cuda.current_stream(dev).wait_stream(self.stream_side)
        return pr

    def _pipe_syn_pass(self, in_set):
        """Clamp, the synthetic batch, its forward and first-order backward, the matching loss against the real gradient lists of set
        ``in_set``, the second-order backward into the points and the update -- all on the CURRENT stream (one stream: this pass is a chain
        of small launches that runs beside the real pass)."""
        pp, dev, G, ipc, npts = self._pipe, self.device, self.G, self.ipc, self.num_points
        cur = torch.cuda.current_stream(dev)
        cr = self.clamp_range
        L.call("dd3d_syn_head", L.ptr(self.image_syn), self.image_syn.shape[0], npts, L.ptr(self.dest_cloud), int(cr is not None),
               float(cr[0]) if cr else 0.0, float(cr[1]) if cr else 0.0, int(self.zero_first_axis), self.quant, L.ptr(pp.scoords), L.ptr(pp.sfeats), L.stream())
        ps = _Pass(self.network, pp.scoords, G * ipc, G, self.syn_group, self.syn_gco, G * ipc, G * ipc * npts, cur, None, True, self.syn_pt_off, npts, pp.extent,
                   cur, None, Gs=G, max_per_group=ipc, counters=pp.counters_s, feats3=pp.sfeats)
        ps.run(pp.sfeats, self.syn_labels)
        need = [i in (1, 4, 6, 7) for i in range(8)]
        loss_g, d = grouped_match_loss(ps.gw, pp.gwr[in_set], self.dist_opt, G, need)
        dfeats = ps.backward_to_points(d[1], d[4], d[6], d[7], ipc)
        grad = torch.empty_like(self.image_syn)
        L.call("dd3d_sgd_scatter", L.ptr(self.image_syn), L.ptr(self.momentum_buf), L.ptr(dfeats), L.ptr(self.dest_cloud), self.image_syn.shape[0], npts,
               self.lr, self.momentum, L.ptr(grad), L.stream())
        return ps, loss_g, grad, loss_g.sum()

    def _pipe_body(self, parity):
        """One pipelined step: the synthetic pass of THIS step (third stream; it reads the real gradient lists of set ``parity``, written one
        step ago) beside the real pass of the NEXT step's sample (main + second stream; it writes set ``1 - parity``).  Exact: the network
        is frozen for the whole run (mnist_distill.py:262: the inner loop is declared and never run), so a real batch's gradient lists
        depend on its sample only.  Only main <-> side and main <-> third-stream waits (what the capture tolerates on ROCm 7.2)."""
        pp, dev = self._pipe, self.device
        main, syn_stream = torch.cuda.current_stream(dev), self.stream_side2
        with L.ws_scope(self._scope + "pipe"):
            self._pipe_gather()   # (the fork comes behind the main stream's first launch: a fork from a capture stream that has no node yet crashed the capture)
            syn_stream.wait_stream(main)
            # (a HIP graph dispatches in capture order and a branch captured first holds back the one captured after it: DD3D_PIPE_ORDER
            # chooses which pass is enqueued first)
            if os.environ.get("DD3D_PIPE_ORDER", "syn") == "syn":
                with torch.cuda.stream(syn_stream):
                    ps, loss_g, grad, total = self._pipe_syn_pass(parity)
                pr = self._pipe_real_pass(1 - parity)
            else:
                pr = self._pipe_real_pass(1 - parity)
                with torch.cuda.stream(syn_stream):
                    ps, loss_g, grad, total = self._pipe_syn_pass(parity)
            main.wait_stream(syn_stream)
        self.last = {"per_class": loss_g, "grad": grad, "gw_real": _GradList(pp.gwr[parity], self.network.dense_extent, pp.extent),
                     "gw_syn": _GradList([t for t in ps.gw], self.network.dense_extent, pp.extent), "logits": (None, ps.logits),
                     "conv1_maps": [(p_.maps[4], p_.M[1], p_.cnt[1]) for p_ in (ps, pr)], "passes": (ps, pr)}
        return total

    def _outer_step_pipelined(self):
        """Host side of the pipelined step.  Step 0 primes the pipe eagerly: real pass of sample 0, then a body (synthetic pass 0 beside the
        real pass of sample 1); from step 1 on the body of parity t % 2 is captured once and replayed.  The host draws sample t + 1 before
        step t -- the same stream of draws as the merged step's, one draw earlier."""
        if self._pipe is None:
            self._build_pipe()
        pp = self._pipe
        if pp.steps > 0 and self._param_stamp() != self._graph_stamp:
            raise RuntimeError("the pipelined step holds gradient lists of a real batch computed with the old weights; build a new distiller after a weight change")
        if pp.steps == 0:
            self._graph_stamp = self._param_stamp()
            self._upload_ids()
            with L.ws_scope(self._scope + "pipe"):
                self._pipe_gather()
                pr0 = self._pipe_real_pass(0)
            torch.cuda.synchronize(self.device)   # (once: the priming pass's buffers go back to the allocator only when its launches are done)
            del pr0
        self._upload_ids()   # the sample of the NEXT step
        parity = pp.steps % 2
        if pp.steps == 0 or not self.use_graph:
            loss = self._pipe_body(parity)
        elif pp.graphs[parity] is None:
            g = torch.cuda.CUDAGraph()
            L.note_capture()
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                loss = self._pipe_body(parity)
            pp.graphs[parity] = (g, loss, self.last)
            self._graph = g
            g.replay()
        else:
            g, loss, self.last = pp.graphs[parity]
            g.replay()
        pp.steps += 1
        self.steps_done += 1
        self._last_loss = loss
        return loss

    # ------------------------------------------------------------------ the step
    def _merge_eager(self, rb, xyz):
        """Eager mode: the merged batch (synthetic clouds, then the real batch ``rb``) from host-sized tensors."""
        G, Bs, Ns = self.G, self.G * self.ipc, xyz.shape[0]
        rc = rb["coordinates"].clone()
        rc[:, 0] += Bs
        pt = rb.get("pt_off")
        if pt is None:
            cnt = torch.bincount(rb["coordinates"][:, 0].long(), minlength=int(rb["num_clouds"]))
            pt = torch.cat([cnt.new_zeros(1), torch.cumsum(cnt, 0)]).to(_I32)
        return {"coordinates": torch.cat([torch.cat([self.syn_batch_col, L.true_divide(xyz, self.quant)], 1), rc]), "features": torch.cat([xyz, rb["features"]]),
                "labels": torch.cat([self.syn_labels, rb["labels"].to(_I32)]), "num_clouds": Bs + int(rb["num_clouds"]),
                "cloud_group": torch.cat([self.syn_group, rb["cloud_group"] + G]),
                "group_cloud_off": torch.cat([self.syn_gco, rb["group_cloud_off"][1:] + Bs]),
                "pt_off": torch.cat([self.syn_pt_off, pt[1:] + Ns]), "n_dev": None, "extent": rb.get("extent"), "gmin_floor": rb.get("gmin_floor"), "max_per": rb.get("max_per")}

    def _run_deferred(self, stash):
        """The loss-value-only work a deferred step left behind: both convolutions' kernel gradients over the stashed operands (the
        first layer's on the second side stream) and the matching-loss terms over them and the two instance-norm biases;
        stash["loss"] = the step's complete per-class loss.  Enqueued on the side stream(s); the caller joins."""
        G, dev, bt = self.G, self.device, stash["ld"]
        side, side2 = self.stream_defer, self.stream_side2   # not stream_side: this step's classifier gradients must not queue behind them
        if side2 is not side:
            side2.wait_stream(side)
        with torch.cuda.stream(side2):
            g_W0 = conv_kernel_gradient(stash["F0"], 3, stash["gY0"], stash["nbr1"], bt, bt, stash["gro"][0], 2 * G, dev)
        with torch.cuda.stream(side):
            g_W1 = conv_kernel_gradient(stash["P0"], 128, stash["gY1"], stash["nbr2"], bt, bt, stash["gro"][1], 2 * G, dev)
            if side2 is not side:
                side.wait_stream(side2)
            gw = [g_W0, stash["g_b1"], g_W1, stash["g_b5"]]
            loss_b, _ = grouped_match_loss([t[:G] for t in gw], [t[G:] for t in gw], self.dist_opt, G, [False] * 4)
            torch.add(stash["loss_a"], loss_b, out=stash["loss"])

    def flush(self):
        """Deferred-loss mode: complete the last step's loss value now (the same launches the next step's graph would make) and return it
        (0-d device tensor).  Without a pending step, or outside the deferred mode: the last returned loss."""
        if self._pending is None:
            return self._last_loss
        main = torch.cuda.current_stream(self.device)
        self.stream_defer.wait_stream(main)
        with L.ws_scope(self._scope + "flush"):
            self._run_deferred(self._pending)
        main.wait_stream(self.stream_defer)
        return self._pending["loss"].sum()

    def _do_exchange(self, flat):
        self.exchange(flat)

    @staticmethod
    def _finish(gen):
        """Run a step generator to its end (it must not yield again) and return its value."""
        try:
            next(gen)
        except StopIteration as e:
            return e.value
        raise RuntimeError("the step yielded twice")

    def _compute(self, mb, static, stash=None):
        """_compute_gen with the exchange of a split real batch (if any) done in line."""
        gen = self._compute_gen(mb, static, stash)
        try:
            flat = next(gen)
        except StopIteration as e:
            return e.value
        self._do_exchange(flat)
        return self._finish(gen)

    def _compute_gen(self, mb, static, stash=None):
        """Everything after the merged batch ``mb`` exists: one forward / first-order backward over synthetic + real clouds, the matching
        loss, the second-order backward into the synthetic points, the SGD update.  ``stash``: deferred-loss mode (see __init__).
        A generator: with a split real batch it yields ONCE, the flat buffer of partial real gradients that the caller sums over the
        ranks sharing the classes (between two graph captures, or in line); otherwise it runs through."""
        dev, G, ipc, npts = self.device, self.G, self.ipc, self.num_points
        # cached layouts of the frozen weights are (re)built here if a weight changed
        net = self.network
        extent = mb.get("extent") or self.site_extent()
        compact_site_major(net.classifier.weight, 128, net.dense_extent, extent)
        transposed_kernel(net.features[4].kernel)
        p = _Pass(net, mb["coordinates"], int(mb["num_clouds"]), G + self.Gr, mb["cloud_group"], mb["group_cloud_off"], G * ipc, G * ipc * npts,
                  self.stream_side, mb.get("n_dev"), static, mb.get("pt_off"), npts, extent, self.stream_side2, stash, Gs=G,
                  dl_scale=self.dl_scale, gmin_floor=mb.get("gmin_floor"), max_per_group=max(ipc, int(mb.get("max_per") or mb["num_clouds"])),
                  flags=mb.get("flags"), counters=self._counters,
                  feats3=mb["features"].detach().contiguous() if os.environ.get("DD3D_MAPS_F0", "1") != "0" and mb["features"].shape[1] == 3 else None)
        # Second-order pass started early (DD3D_SO_EARLY, default on): of the four differentiable gradient entries only the first level's IN
        # weight waits for the tail of the first-order backward (the input gradient of the 128-channel convolution and the norm / pool
        # backward of level 1: ~260 us of matrix- and bandwidth-bound launches).  The loss over the other three and the second-order pass
        # down to the second convolution -- a chain of ten small, latency-bound launches -- run on the side stream NEXT to those.
        so_early = (stash is None and not self.split and self.dist_opt in ("ours", "modmse", "mse") and os.environ.get("DD3D_LOSS_SPLIT", "1") != "0"
                    and self.so_early)
        early = None
        if so_early:
            def early(ps, ev_l2):
                sd = ps.cls_stream
                sd.wait_event(ev_l2)
                with torch.cuda.stream(sd):
                    gs, gr = [t[:G] for t in ps.g_head], [t[G:] for t in ps.g_head]
                    loss_a1, d3 = grouped_match_loss(gs, gr, self.dist_opt, G, [True] * 3)
                    hP0 = ps.backward_to_conv(d3[0], d3[1], d3[2], ipc)
                    ev_h = torch.cuda.Event()
                    ev_h.record(sd)
                if not static:   # allocated on the side stream, consumed on the main one
                    cur = torch.cuda.current_stream(dev)
                    loss_a1.record_stream(cur)
                    hP0.record_stream(cur)
                return loss_a1, hP0, ev_h
        p.run(mb["features"], mb["labels"], defer=stash is not None, early=early)
        gw_syn, gw_real = [None if t is None else t[:G] for t in p.gw], [None if t is None else t[G:] for t in p.gw]
        # Only these entries of gw_syn depend on the synthetic points: IN weights, classifier weight and bias.  For the
        # layer-additive metrics the loss is split accordingly: the part over these four entries (with its gradient) runs on the main stream
        # as soon as they exist, and the second-order pass follows at once; the VALUE of the other four (both kernel gradients, both IN
        # biases) is added from the side stream, where the kernel gradients are still being computed.  'cos' normalises over all
        # layers together and keeps the joint launch.
        main, side = torch.cuda.current_stream(dev), self.stream_side
        diff, const = [1, 4, 6, 7], [0, 2, 3, 5]
        grad = total = None   # set by the branch that has already enqueued the update / the one-launch loss sum
        if self.split:
            # second sharding level: this rank's PARTIAL real gradients (already weighted n_part / n) are summed over the ranks that share
            # the classes; every stream joins first (a graph capture may end here)
            main.wait_stream(side)
            main.wait_stream(p.side2)
            flat = torch.cat([t.reshape(-1) for t in gw_real])
            yield flat
            full, off = [], 0
            for t in gw_real:
                full.append(flat[off:off + t.numel()].view(t.shape))
                off += t.numel()
            gw_real = [t.index_select(0, self.own_idx) for t in full] if self.real_classes != self.classes else full
            need = [i in diff for i in range(8)]
            loss_g, d = grouped_match_loss(gw_syn, gw_real, self.dist_opt, G, need)
            dfeats = p.backward_to_points(d[1], d[4], d[6], d[7], ipc)
        elif stash is not None:
            main.wait_event(p.ev_cls)
            loss_a, d4 = grouped_match_loss([gw_syn[i] for i in diff], [gw_real[i] for i in diff], self.dist_opt, G, [True] * 4, loss_out=stash["loss_a"])
            dfeats = p.backward_to_points(d4[0], d4[1], d4[2], d4[3], ipc)
            main.wait_stream(side)   # the classifier gradients' stream (and whatever else the caller put there) joins before the update
            loss_g = loss_a
        elif so_early:
            loss_a1, hP0, ev_h = p.early_out
            loss_a2, d1 = grouped_match_loss([gw_syn[1]], [gw_real[1]], self.dist_opt, G, [True])   # the first level's IN weight
            # The constant entries' loss terms, in two parts: the 128-channel kernel gradient and the second level's IN bias right behind
            # that kernel gradient on its own stream (next to the first layer's kernel gradient, which ends last), the first layer's kernel
            # gradient and IN bias behind it -- a list small enough for the one-launch form, so the tail of a step is kernel gradient ->
            # slab sum -> ONE loss launch -> total instead of ... -> two loss launches over all four entries -> total
            with torch.cuda.stream(side):
                loss_b, _ = grouped_match_loss([gw_syn[i] for i in (3, 5)], [gw_real[i] for i in (3, 5)], self.dist_opt, G, [False] * 2)
            side.wait_stream(p.side2)
            with torch.cuda.stream(side):
                loss_b0, _ = grouped_match_loss([gw_syn[i] for i in (0, 2)], [gw_real[i] for i in (0, 2)], self.dist_opt, G, [False] * 2)
            main.wait_event(ev_h)
            dfeats = p.backward_from_conv(hP0, d1[0])
            # the update needs the point gradient only: it goes into the queue BEFORE the join with the loss terms of the constant entries
            # (which end last: they wait for the first layer's kernel gradient), and the per-class and total loss are ONE launch behind the
            # join instead of two additions and a reduction -- two dependent graph nodes less at the very end of a step
            grad = self._scatter_update(dfeats, G, ipc, npts)
            main.wait_stream(side)
            if not static:
                loss_b.record_stream(main)
                loss_b0.record_stream(main)
            loss_g, total = torch.empty(G, dtype=_F32, device=dev), torch.empty((), dtype=_F32, device=dev)
            L.call("dd3d_loss_total", L.ptr(loss_a1), L.ptr(loss_a2), L.ptr(loss_b), L.ptr(loss_b0), G, L.ptr(loss_g), L.ptr(total), L.stream())
        elif self.dist_opt in ("ours", "modmse", "mse") and os.environ.get("DD3D_LOSS_SPLIT", "1") != "0":
            main.wait_event(p.ev_cls)
            loss_a, d4 = grouped_match_loss([gw_syn[i] for i in diff], [gw_real[i] for i in diff], self.dist_opt, G, [True] * 4)
            side.wait_stream(p.side2)
            with torch.cuda.stream(side):
                loss_b, _ = grouped_match_loss([gw_syn[i] for i in const], [gw_real[i] for i in const], self.dist_opt, G, [False] * 4)
            dfeats = p.backward_to_points(d4[0], d4[1], d4[2], d4[3], ipc)
            main.wait_stream(side)
            if not static:
                loss_b.record_stream(main)
            loss_g = loss_a + loss_b
        else:
            main.wait_stream(side)
            main.wait_stream(p.side2)
            need = [i in diff for i in range(8)]
            loss_g, d = grouped_match_loss(gw_syn, gw_real, self.dist_opt, G, need)
            dfeats = p.backward_to_points(d[1], d[4], d[6], d[7], ipc)
        if grad is None:
            grad = self._scatter_update(dfeats, G, ipc, npts)
        Bs = G * ipc
        self.last = {"per_class": loss_g, "grad": grad, "gw_real": _GradList(gw_real, net.dense_extent, extent),
                     "gw_syn": _GradList(gw_syn, net.dense_extent, extent), "logits": (p.logits[Bs:], p.logits[:Bs]),
                     "conv1_maps": [(p.maps[4], p.M[1], p.cnt[1])], "passes": (p,)}
        return total if total is not None else loss_g.sum()

    def _scatter_update(self, dfeats, G, ipc, npts):
        """The point gradient of the owned clouds into a tensor shaped like image_syn, and (unless the caller asked for the gradient only)
        the SGD step with momentum, on the current stream.  Returns the gradient tensor."""
        grad = torch.empty_like(self.image_syn)
        fused_tail = self._apply_update and self.image_syn.shape[1] == npts
        if not fused_tail:
            grad.zero_()
            grad[self.syn_rows] = dfeats.reshape(G * ipc, npts, 3)
        if fused_tail:
            # gradient scatter + SGD in one launch; momentum_buf starts at zero, so the first step needs no special case (0.5 * 0 + g == g)
            L.call("dd3d_sgd_scatter", L.ptr(self.image_syn), L.ptr(self.momentum_buf), L.ptr(dfeats), L.ptr(self.dest_cloud),
                   self.image_syn.shape[0], npts, self.lr, self.momentum, L.ptr(grad), L.stream())
        elif self._apply_update:
            L.call("dd3d_sgd_momentum_clamp", L.ptr(self.image_syn), L.ptr(grad), L.ptr(self.momentum_buf), self.image_syn.numel(), self.lr,
                   self.momentum, 0, 0, 0.0, 27.0, L.stream())
        return grad

    def _static_body(self, parity=0):
        """_static_body_gen with the exchange of a split real batch (if any) done in line (the eager first step)."""
        gen = self._static_body_gen(parity)
        try:
            flat = next(gen)
        except StopIteration as e:
            return e.value
        self._do_exchange(flat)
        return self._finish(gen)

    def _static_body_gen(self, parity=0):
        """clamp + on-device batch assembly + the step, all on fixed addresses with device-side sizes (what the graph captures).
        Deferred-loss mode: the previous step's pending loss-value work first, on the side streams, next to this step's own kernels.
        A generator like _compute_gen (yields the exchange buffer of a split real batch)."""
        st = self._static
        stash = self._stash[parity] if self.defer else None
        Bs, Ns = self.G * self.ipc, self.G * self.ipc * self.num_points
        # clamp + the owned synthetic clouds into the head of the merged batch (coordinates = xyz / quantization_size, features = xyz:
        # the same tensor twice, Mnist2D.py:84-92), one launch
        cr = self.clamp_range
        # DD3D_ASSEMBLE (default): clamp + synthetic head + real-batch gather + the map builder's cleared flag words in ONE launch
        # (dd3d_assemble_batch: the gathering workgroups compute their own offsets); 0: the separate launches
        one = os.environ.get("DD3D_ASSEMBLE", "1") != "0" and st.B > 0
        lds_maps = 0 < self.num_points <= int(L.lib.dd3d_cloud_maps_max_points())
        flags = st.flags if one and lds_maps else None
        if one:
            L.call("dd3d_assemble_batch", L.ptr(self.image_syn), self.image_syn.shape[0], self.num_points, L.ptr(self.dest_cloud), int(cr is not None),
                   float(cr[0]) if cr else 0.0, float(cr[1]) if cr else 0.0, int(self.zero_first_axis), self.quant, L.ptr(st.pool_xyz), L.ptr(st.pool_off),
                   st.n_pool, L.ptr(st.ids), st.B, self.num_points, st.bound_total, Bs, Ns, L.ptr(st.coords), L.ptr(st.feats), L.ptr(st.pt_off), L.ptr(st.n),
                   L.ptr(st.err), L.ptr(st.flags), int(st.flags.numel()), L.stream())
        else:
            L.call("dd3d_syn_head", L.ptr(self.image_syn), self.image_syn.shape[0], self.num_points, L.ptr(self.dest_cloud), int(cr is not None),
                   float(cr[0]) if cr else 0.0, float(cr[1]) if cr else 0.0, int(self.zero_first_axis), self.quant, L.ptr(st.coords), L.ptr(st.feats),
                   L.stream())
        # (forked behind the first kernel of the step: a fork from a capture stream that has no node yet crashed hipStreamEndCapture)
        if self.defer and self._pending is not None:
            self.stream_defer.wait_stream(torch.cuda.current_stream(self.device))
            with L.ws_scope(self._scope + "d%d" % parity):
                self._run_deferred(self._pending)
        if not one:
            L.call("dd3d_gather_clouds", L.ptr(st.pool_xyz), L.ptr(st.pool_off), st.n_pool, L.ptr(st.ids), st.B, self.num_points, st.bound_total,
                   Bs, Ns, self.quant, L.ptr(st.coords), L.ptr(st.feats), L.ptr(st.pt_off), L.ptr(st.n), L.ptr(st.err), L.stream())
        mb = {"coordinates": st.coords, "features": st.feats, "labels": st.labels, "num_clouds": Bs + st.B, "cloud_group": st.cloud_group,
              "group_cloud_off": st.gco, "n_dev": st.n, "pt_off": st.pt_off, "gmin_floor": st.gmin_floor, "max_per": st.max_per, "flags": flags}
        with L.ws_scope(self._scope + ("p%d" % parity if self.defer else "")):
            out = yield from self._compute_gen(mb, True, stash)
        if self.defer:
            main = torch.cuda.current_stream(self.device)
            main.wait_stream(self.stream_defer)   # (also what lets a capture end: every forked stream rejoins)
            if self.stream_side2 is not self.stream_defer:
                main.wait_stream(self.stream_side2)
            prev, self._pending = self._pending, stash
            return prev["loss"].sum() if prev is not None else torch.zeros((), device=self.device)
        return out

    def row_counts(self):
        """((M0, M1, M2),) of the last step's merged batch (one device->host read in graph mode)."""
        return tuple(tuple(int(x) for x in p.cm.counts[:3].tolist()) for p in self.last["passes"])

    def check(self):
        """Raise if any kernel of the last step flagged an error (graph mode defers these checks; eager mode raises in the step)."""
        for p in self.last["passes"]:
            p.cm.check()
        if self._static is not None and int(self._static.err.item()) & 4:
            raise RuntimeError("real-batch assembly: sample index outside the pool, or the batch exceeds its point bound")

    def outer_step(self, real_batch=None, apply_update=True, force_eager=False):
        """One matching step (mnist_distill.py:309-369 without evaluation/logging).  Returns the summed matching loss (0-d device
        tensor; in graph mode it is a fixed buffer overwritten by the next step)."""
        if self._closed:
            raise RuntimeError("this FusedDistiller was closed")
        self._apply_update = apply_update
        # The side streams take the priority of the stream the step is launched on: a step launched on a high-priority stream whose side
        # streams had the default priority never finished (measured on ROCm 7.2: the launching stream waits for its side streams' events
        # while its own queue outranks theirs)
        prio = getattr(torch.cuda.current_stream(self.device), "priority", 0)
        if self.use_graph and 0 < int(os.environ.get("GPU_MAX_HW_QUEUES", "4") or 4) < 4:
            # measured on ROCm 7.2: with two hardware queues per process the first launch of the captured three-branch graph ends in a HOST
            # segmentation fault inside hipGraphLaunch (with eight the replay is three times slower); HIP's default is four
            raise RuntimeError("graph mode needs GPU_MAX_HW_QUEUES >= 4 (HIP's default): with fewer hardware queues hipGraphLaunch faults on "
                               "ROCm 7.2 for this step's three-branch graph; unset the variable or use graph=False")
        if prio != 0 and self.use_graph:
            # (eager steps run there once the side streams share the priority; the REPLAY of the captured graph did not return: the
            # runtime's own graph streams keep the default priority)
            raise RuntimeError("graph mode needs a stream of default priority: a HIP-graph replay launched on a high-priority stream hangs on ROCm 7.2 "
                               "(use graph=False on such a stream)")
        if prio != getattr(self.stream_side, "priority", 0):
            if self.steps_done > 0:
                raise RuntimeError("this FusedDistiller has taken steps on a stream of another priority; use one stream priority per distiller")
            self.stream_side, self.stream_side2, self.stream_data = (torch.cuda.Stream(self.device, priority=prio) for _ in range(3))
            self.stream_defer = self.stream_side2
        if self.pipeline and real_batch is None and apply_update:
            return self._outer_step_pipelined()
        if self.use_graph and real_batch is None and apply_update:
            if self._static is None:
                self._build_static()
            self._upload_ids()
            if self._graph is not None and self._param_stamp() != self._graph_stamp:
                self.invalidate()   # a weight changed (load_state_dict, optimizer step): the captured graph reads stale cached layouts
            parity = self.steps_done % 2 if self.defer else 0
            if force_eager or self.steps_done == 0 or self._needs_eager:
                # the first step runs eagerly: it sizes the workspaces and builds the cached weight layouts the graph will reuse
                loss = self._static_body(parity)
                self._needs_eager = False
            else:
                if self._graphs[parity] is None:
                    g = torch.cuda.CUDAGraph()
                    L.note_capture()
                    # thread_local: only this thread's calls are checked during capture (an RCCL watchdog thread polling its events
                    # must not invalidate it)
                    with torch.cuda.graph(g, capture_error_mode="thread_local"):
                        gl = self._static_body(parity)
                    # deferred mode: the graph of parity p runs the pending work of the step before it (stash 1 - p) and leaves its own in stash p
                    self._graphs[parity] = (g, gl, self.last, self._pending)
                    self._graph_stamp = self._param_stamp()
                    self._graph, loss = g, gl
                    self._graph.replay()
                else:
                    self._graph, loss, self.last, self._pending = self._graphs[parity][:4]
                    self._graph.replay()
            self._last_loss = loss
            self.steps_done += 1
            return loss
        self.clamp()
        main = torch.cuda.current_stream(self.device)
        if real_batch is not None:
            if self.split:
                raise ValueError("a split real batch is sampled inside the step (every rank of the group draws the same permutation)")
            rb = dict(real_batch)
            rb["extent"] = self.site_extent(real_batch)
        elif self._prefetched is not None:
            rb, self._prefetched = self._prefetched, None
            main.wait_stream(self.stream_data)
            for t in rb.values():   # allocated on the data stream, consumed on the main one
                if torch.is_tensor(t) and t.is_cuda:
                    t.record_stream(main)
        else:
            rb = self.sample_real()
        xyz = self.image_syn[self.syn_rows].reshape(-1, 3)
        total = self._compute(self._merge_eager(rb, xyz), False)
        if apply_update:
            self.steps_done += 1
        if real_batch is None and not self.use_graph and self.prefetch:
            with torch.cuda.stream(self.stream_data):
                self._prefetched = self.sample_real()
        return total


class ChunkedDistiller:
    """The matching step over CHUNKS of classes in sequence on one device: one FusedDistiller per chunk, all of them updating their own
    rows of ONE synthetic-point tensor (and momentum buffer).  The classes are independent inside a step (mnist_distill.py:335), so
    this is the class-sharding mechanism of dd3d_amd/sharding.py run in sequence; it is what a configuration needs whose per-group
    gradient lists do not fit beside each other (BASELINE configs[4]: 55 classes, a 2.25 M-column classifier -- 54 GB of classifier
    gradients for all 110 groups at once).  Eager launches: the chunks reuse one another's memory through the allocator, and kernels of
    this size are GPU-bound anyway.  Same results as one FusedDistiller over all classes (tests/test_gpu_fused.py); a chunk's clamp also goes
    over the clouds the chunks before it have just moved -- the clamp the next step applies to them anyway."""

    def __init__(self, images_all, indices_class, network, ipc, num_classes, chunk, classes=None, image_syn=None, **kw):
        kw.pop("graph", None)
        self.classes = list(range(num_classes)) if classes is None else list(classes)
        self.chunks = [self.classes[a:a + chunk] for a in range(0, len(self.classes), chunk)]
        self.parts = []
        for ch in self.chunks:
            d = FusedDistiller(images_all, indices_class, network, ipc, num_classes=num_classes, classes=ch, image_syn=image_syn, graph=False, **kw)
            if self.parts:   # every chunk works on the first one's tensors and draws from its sample stream (class after class, as one batch would)
                d.image_syn, d.momentum_buf, d.rng = self.parts[0].image_syn, self.parts[0].momentum_buf, self.parts[0].rng
            d.prefetch = False
            self.parts.append(d)
        self.image_syn, self.momentum_buf = self.parts[0].image_syn, self.parts[0].momentum_buf
        self.defer, self.use_graph, self.last, self._rows = False, False, {}, []
        self.count_pairs, self.pairs = False, []   # (measurement) pairs of both kernel maps of every chunk of the last step

    def outer_step(self, **kw):
        kw.pop("force_eager", None)
        total, self._rows, self.pairs = None, [], []
        for d in self.parts:
            loss = d.outer_step(**kw)
            total = loss if total is None else total + loss
            self._rows.append(d.row_counts()[0])
            if self.count_pairs:
                ps, (M0, M1, _) = d.last["passes"][0], self._rows[-1]
                self.pairs.append((int((ps.maps[0][:, :M0] >= 0).sum().item()), int((ps.maps[4][:, :M1] >= 0).sum().item())))
            self.last = d.last
            if d is not self.parts[-1]:
                d.last = {}       # the chunk's gradient lists go back to the allocator before the next chunk runs
        return total

    def row_counts(self):
        """((M0, M1, M2),) summed over the chunks of the last step."""
        return (tuple(sum(r[i] for r in self._rows) for i in range(3)),)

    def check(self):
        self.parts[-1].check()

    def close(self):
        for d in self.parts:
            d.close()
