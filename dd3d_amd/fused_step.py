"""Class-batched gradient-matching step: the same arithmetic as ``Distiller.outer_step`` (mnist_distill.py:309-374 of the reference),
but ALL classes owned by this process go through every kernel together and nothing runs through torch autograd.

Why it exists (SURVEY.md section 7 "tiny problem, big GPU"): one class is ~7k voxels at the 128-channel layer -- about a hundred
64-row tiles for 256 CUs -- and the reference loop pays hundreds of launches plus three host syncs per class.  Classes are
independent inside a step (mnist_distill.py:335), so a (class, cloud) batch index turns 10 small problems into one large one:
the class of a cloud is its *group*; per-group reductions (kernel gradients, affine gradients, classifier gradients, the matching
loss) come out of the same launches.  One device->host read per pass (row counts) is the only synchronisation.

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
from typing import List, Optional

import numpy as np
import torch

from . import _lib as L
from .distillation_loss import _METRICS, _rows_cols
from .ops import DensePlacement, _site_strides, sl_bwd_x_call, sl_fwd_call, transposed_kernel
from .sparse import CloudCoordinateManager, CoordinateManager
from .utils.Mnist2D import batch_from_images

_F32, _I32 = torch.float32, torch.int32


def _empty(shape, dev, dtype=_F32):
    return torch.empty(shape, dtype=dtype, device=dev)


class _Pass:
    """Forward + first-order backward of MEConvExp over one class-batched set of clouds; keeps what the second-order pass needs."""

    def __init__(self, net, coords, B, G, cloud_group, group_cloud_off, n_dev=None, static=False, pt_off=None, max_points=0):
        """Phase 1 (no host sync): every coordinate map of the pass, enqueued on the current stream.
        ``static``: never read a size back -- every launch covers its bound and takes the true row count from the device
        (``n_dev`` = number of valid rows of ``coords``), so the pass can be captured into a HIP graph."""
        dev = coords.device
        self.dev, self.B, self.G, self.net = dev, B, G, net
        self.cloud_group, self.gco = cloud_group, group_cloud_off
        self.static = static
        if pt_off is not None and 0 < max_points <= int(L.lib.dd3d_cloud_maps_max_points()):
            # clouds small enough for LDS: the whole pyramid (+ dense placement) in three launches
            cm = CloudCoordinateManager(coords, pt_off, B, max_points, net.dense_extent, cloud_group, G, group_cloud_off)
        else:
            cm = CoordinateManager(coords, B, n_dev=n_dev)
        self.cm = cm
        self.N = cm.N
        k0, p0 = cm.kernel_map(1, 1, 3), cm.kernel_map(1, 2, 2)
        k1, p1 = cm.kernel_map(2, 2, 3), cm.kernel_map(2, 4, 2)
        seg1, seg2 = cm.seg_off(1), cm.seg_off(2)
        self.dp = DensePlacement(cm, 4, net.dense_extent, cloud_group, G, group_cloud_off)
        if getattr(cm, "group_row_off", None) is not None:
            gro1, gro2 = cm.group_row_off[0], cm.group_row_off[1]
        else:
            gidx = group_cloud_off.long()
            gro1, gro2 = seg1[gidx].contiguous(), seg2[gidx].contiguous()   # rows of every group at tensor strides 1 and 2
        self._pre = (k0, p0, k1, p1, seg1, seg2, gro1, gro2)

    def run(self, feats, labels, keep_ga: bool):
        """Phase 2: ONE device->host read (row counts), then forward + first-order backward on the current stream."""
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
        from .ops import site_major
        Wsm = self.Wsm = site_major(cls_w, 128, S)
        sst = _site_strides(J, 128, S)
        W1t = self.W1t = transposed_kernel(conv1.kernel)
        self.params = (W0, W1, w1, b1, w5, b5)

        # ---------------- forward ----------------
        feats = feats.detach().contiguous()
        F0 = self.F0 = _empty((M0, 3), dev)
        L.call("dd3d_voxel_mean_fwd", L.ptr(feats), self.N, 3, L.ptr(cm.vox_off), L.ptr(cm.vox_pts), M0, c0, L.ptr(F0), st)
        Y0 = _empty((M0, 128), dev)
        L.call("dd3d_conv_fwd", L.ptr(F0), M0, 3, L.ptr(W0), 128, L.ptr(k0), ld0, 27, M0, c0, 0, L.ptr(Y0), st)
        xh0, is0 = self.xh0, self.is0 = _empty((M0, 128), dev), _empty((B, 128), dev)
        L.call("dd3d_inorm_fwd", L.ptr(Y0), L.ptr(seg1), B, 128, L.ptr(xh0), L.ptr(is0), st)
        P0, arg0 = self.P0, self.arg0 = _empty((M1, 128), dev), _empty((M1, 128), dev, _I32)
        L.call("dd3d_affine_relu_pool_fwd", L.ptr(xh0), L.ptr(w1), L.ptr(b1), 1, 128, L.ptr(p0), pl0, 8, M1, c1, L.ptr(P0), L.ptr(arg0), st)
        Y1 = _empty((M1, 128), dev)
        L.call("dd3d_conv_fwd", L.ptr(P0), M1, 128, L.ptr(W1), 128, L.ptr(k1), ld1, 27, M1, c1, 0, L.ptr(Y1), st)
        xh1, is1 = self.xh1, self.is1 = _empty((M1, 128), dev), _empty((B, 128), dev)
        L.call("dd3d_inorm_fwd", L.ptr(Y1), L.ptr(seg2), B, 128, L.ptr(xh1), L.ptr(is1), st)
        P1, arg1 = self.P1, self.arg1 = _empty((M2, 128), dev), _empty((M2, 128), dev, _I32)
        L.call("dd3d_affine_relu_pool_fwd", L.ptr(xh1), L.ptr(w5), L.ptr(b5), 1, 128, L.ptr(p1), pl1, 8, M2, c2, L.ptr(P1), L.ptr(arg1), st)
        logits = self.logits = _empty((B, J), dev)
        sl_fwd_call(P1, dp, None, Wsm, sst, J, cls_b, logits)
        prob, dl = self.prob, self.dl = _empty((B, J), dev), _empty((B, J), dev)
        closs, self.ce = _empty((B,), dev), _empty((G,), dev)
        L.call("dd3d_softmax_ce", L.ptr(logits), L.ptr(labels), L.ptr(cloud_group), L.ptr(group_cloud_off), G, B, J, L.ptr(prob), L.ptr(dl),
               L.ptr(closs), L.ptr(self.ce), st)

        # ---------------- first-order backward: one gradient list per group ----------------
        g_cb = _empty((G, J), dev)
        L.call("dd3d_group_sum_rows", L.ptr(dl), L.ptr(group_cloud_off), G, J, L.ptr(g_cb), st)
        g_cw = _empty((G, J, 128 * S), dev)
        pst = (ctypes.c_int64 * 4)(J * 128 * S, 128 * S, S, 1)
        L.call("dd3d_sparse_linear_bwd_w", L.ptr(dl), L.ptr(P1), L.ptr(dp.vox_at), S, B, L.ptr(group_cloud_off), G, 128, J, L.ptr(g_cw), pst,
               g_cw.numel(), st)
        gP1 = _empty((M2, 128), dev)
        sl_bwd_x_call(dl, dp, None, Wsm, sst, J, 0, gP1)
        gY1, self.ga1, g_w5, g_b5 = self._arp_inorm_bwd(gP1, arg1, cm.parent(4), xh1, is1, w5, b5, seg2, M1, keep_ga, True)
        g_W1 = self._bww(P0, 128, gY1, k1, ld1, M1, gro2)
        gP0 = _empty((M1, 128), dev)
        L.call("dd3d_conv_bwd_data", L.ptr(gY1), M1, c1, 128, L.ptr(W1t), 128, L.ptr(k1), ld1, 27, L.ptr(gP0), st)
        gY0, self.ga0, g_w1, g_b1 = self._arp_inorm_bwd(gP0, arg0, cm.parent(2), xh0, is0, w1, b1, seg1, M0, keep_ga, True)
        g_W0 = self._bww(F0, 3, gY0, k0, ld0, M0, gro1)
        # order of network.parameters()
        self.gw = [g_W0, g_w1, g_b1, g_W1, g_w5, g_b5, g_cw, g_cb]
        return self

    def _arp_inorm_bwd(self, g_pool, arg, parent, xhat, inv_std, w, b, seg, M_in, keep_ga, want_gw, ga_prev=None, gw_extra=None):
        """Backward of max-pool -> ReLU -> affine -> instance norm in one pass per cloud: (dY, ga, d weight [G,128], d bias [G,128])."""
        dev, G, B = self.dev, self.G, self.B
        dY = _empty((M_in, 128), dev)
        ga = _empty((M_in, 128), dev) if keep_ga else None
        gw_w = _empty((G, 128), dev) if want_gw else None
        gw_b = _empty((G, 128), dev) if want_gw else None
        ws = L.workspace(int(L.lib.dd3d_arp_inorm_bwd_ws_bytes(B)), dev)
        L.call("dd3d_arp_inorm_bwd", L.ptr(g_pool), L.ptr(arg), L.ptr(parent), L.ptr(xhat), L.ptr(inv_std), L.ptr(w), L.ptr(b), 128, L.ptr(seg), B,
               L.ptr(self.cloud_group), L.ptr(self.gco), G, L.ptr(dY), L.ptr(ga), L.ptr(ga_prev), L.ptr(gw_extra), L.ptr(gw_w), L.ptr(gw_b),
               L.ptr(ws), ws.numel(), L.stream())
        return dY, ga, gw_w, gw_b

    def _bww(self, X, Cin, dY, nbr, ld, M, gro):
        G = self.G
        dW = _empty((G, 27, Cin, 128), self.dev)
        ws = L.workspace(int(L.lib.dd3d_conv_bwd_weight_ws_bytes(M, G, 27, Cin, 128)), self.dev)
        L.call("dd3d_conv_bwd_weight", L.ptr(X), Cin, L.ptr(dY), 128, L.ptr(nbr), ld, 27, M, L.ptr(gro), G, L.ptr(dW), L.ptr(ws), ws.numel(),
               L.stream())
        return dW

    # ---------------- second-order backward into the point features (synthetic pass only) ----------------
    def backward_to_points(self, G_w1, G_w5, G_cw, G_cb):
        """d(sum_g matching loss) / d feats [N,3], given its gradients w.r.t. the four differentiable entries of gw_syn."""
        dev, B, G, J, S = self.dev, self.B, self.G, self.J, self.S
        M0, M1, M2 = self.M
        k0, ld0, p0, pl0, k1, ld1, p1, pl1, seg1, seg2, gro1, gro2 = self.maps
        W0, W1, w1, b1, w5, b5 = self.params
        dp, cm, st = self.dp, self.cm, L.stream()
        sst = _site_strides(J, 128, S)
        Gsm = _empty((G, S, J, 128), dev)
        L.call("dd3d_classifier_layout", L.ptr(G_cw), G, J, 128, S, 0, L.ptr(Gsm), st)
        # u = dL/d(dlogits): through gw[W] = dlogits^T dense and gw[b] = sum dlogits
        u = _empty((B, J), dev)
        sl_fwd_call(self.P1, dp, self.cloud_group, Gsm, sst, J, None, u)
        dz = _empty((B, J), dev)
        per = int((self.gco[1:] - self.gco[:-1]).max().item()) if self._max_per is None else self._max_per
        L.call("dd3d_softmax_ce_bwd2", L.ptr(self.prob), L.ptr(u), L.ptr(G_cb), L.ptr(self.gco), G, J, per, L.ptr(dz), st)
        # H = dL/dP1: through dense in gw[W] (weights G_cw, coefficients dlogits) and through the logits (weights W, coefficients dz)
        H = _empty((M2, 128), dev)
        sl_bwd_x_call(self.dl, dp, self.cloud_group, Gsm, sst, J, 0, H)
        sl_bwd_x_call(dz, dp, None, self.Wsm, sst, J, 1, H)
        hY1, _, _, _ = self._arp_inorm_bwd(H, self.arg1, cm.parent(4), self.xh1, self.is1, w5, b5, seg2, M1, False, False, self.ga1, G_w5)
        hP0 = _empty((M1, 128), dev)
        L.call("dd3d_conv_bwd_data", L.ptr(hY1), M1, self.cnt[1], 128, L.ptr(self.W1t), 128, L.ptr(k1), ld1, 27, L.ptr(hP0), st)
        hY0, _, _, _ = self._arp_inorm_bwd(hP0, self.arg0, cm.parent(2), self.xh0, self.is0, w1, b1, seg1, M0, False, False, self.ga0, G_w1)
        hF0 = _empty((M0, 3), dev)
        L.call("dd3d_conv_bwd_data", L.ptr(hY0), M0, self.cnt[0], 128, L.ptr(W0), 3, L.ptr(k0), ld0, 27, L.ptr(hF0), st)
        dfeats = _empty((self.N, 3), dev)
        L.call("dd3d_voxel_mean_bwd", L.ptr(hF0), M0, 3, L.ptr(cm.levels[1].inv), L.ptr(cm.vox_off), self.N, L.ptr(dfeats), st)
        return dfeats

    _max_per: Optional[int] = None


def grouped_match_loss(gw_syn: List[torch.Tensor], gw_real: List[torch.Tensor], dis_metric: str, G: int, need_grad, pyt: bool = False):
    """distillation_loss.match_loss for G classes at once: tensors are [G, *param_shape]; returns (loss[G], grads list)."""
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
    loss = _empty((G,), dev)
    ws = L.workspace(int(L.lib.dd3d_match_loss_ws_bytes(segs, n, G)), dev)
    L.call("dd3d_match_loss_fwd", segs, n, G, metric, L.ptr(loss), L.ptr(stats), L.ptr(ws), ws.numel(), L.stream())
    L.call("dd3d_match_loss_bwd", segs, n, G, metric, L.ptr(stats), None, L.ptr(ws), ws.numel(), L.stream())
    return loss, dsyn


class FusedDistiller:
    """Same constructor and state as ``mnist_distill.Distiller`` (MEConvExp only); ``outer_step`` runs the class-batched path.

    ``graph=True`` (default): the whole step -- real-batch assembly from a point pool resident in HBM, both passes, the matching loss,
    the backward into the synthetic points and the SGD update -- is captured ONCE into a HIP graph and replayed; per step the host
    draws the sample indices (``np.random.permutation`` per class, as mnist_distill.py:200-203 does), uploads them and launches the
    graph.  Every launch inside covers a bound and reads the true row counts from device memory, so batches of different sizes
    replay through the same graph.  ``graph=False``: the same kernels enqueued eagerly with exact sizes (one size read-back per pass).
    Both modes produce bit-identical results (``tests/test_gpu_fused.py``)."""

    def __init__(self, images_all, indices_class, network, ipc, num_classes=10, num_points=250, batch_size=256, dist_opt="modmse",
                 dist_lr=0.1, device=None, seed=0, image_syn=None, classes=None, momentum=0.5, graph=True):
        self.device = dev = torch.device(device or "cuda")
        self.images_all = images_all.to(dev)
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
        self.lr, self.momentum, self.steps_done = dist_lr, momentum, 0
        G = self.G = len(self.classes)
        cls = torch.tensor(self.classes, device=dev)
        # rows of image_syn owned by this process, class-major
        self.syn_rows = (cls[:, None] * ipc + torch.arange(ipc, device=dev)[None, :]).reshape(-1)
        self.syn_labels = cls.repeat_interleave(ipc).to(_I32)
        self.syn_group = torch.arange(G, device=dev, dtype=_I32).repeat_interleave(ipc)
        self.syn_gco = (torch.arange(G + 1, device=dev) * ipc).to(_I32)
        self.syn_pt_off = (torch.arange(G * ipc + 1, device=dev) * num_points).to(_I32)
        self.syn_batch_col = torch.arange(G * ipc, device=dev, dtype=_F32).repeat_interleave(num_points)[:, None]
        self.last = {}
        # the real and the synthetic pass are independent until the loss: they run on two streams, so the small synthetic kernels and
        # both chains of latency-bound map kernels overlap with the large real-batch kernels
        self.stream_real, self.stream_syn = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
        # eager mode: next step's real batch is sampled on its own stream while this step's kernels run
        self.stream_data, self._prefetched = torch.cuda.Stream(dev), None
        self.use_graph = bool(graph) and G > 0
        self._graph = self._static = None
        self._scope = "graph%x" % id(self)

    def __del__(self):
        try:
            L.drop_scope(self._scope)
        except Exception:
            pass

    def clamp(self):
        s = self.image_syn
        s.clamp_(0, 27)
        s[:, :, 0] = 0

    # ------------------------------------------------------------------ eager-mode real batch (host-sized tensors)
    def _draw(self):
        """mnist_distill.py:200-203 for every owned class: np.random.permutation(indices_class[c])[:batch_size]."""
        return np.concatenate([self.rng.permutation(self.indices_class[c])[:self.batch_size] for c in self.classes])

    def sample_real(self):
        """mnist_distill.py:200-208 for every owned class at once: one gather + one nonzero."""
        idx = self._draw()
        sizes = [min(self.batch_size, len(self.indices_class[c])) for c in self.classes]
        return self.real_batch_from_images(self.images_all[torch.as_tensor(idx, device=self.device)], sizes)

    def real_batch_from_images(self, images, sizes):
        dev = self.device
        labels = torch.tensor(np.repeat(self.classes, sizes), device=dev)
        b = batch_from_images(images, labels, self.num_points)
        b["cloud_group"] = torch.tensor(np.repeat(np.arange(self.G), sizes), device=dev, dtype=_I32)
        b["group_cloud_off"] = torch.tensor(np.concatenate([[0], np.cumsum(sizes)]), device=dev, dtype=_I32)
        b["max_per"] = int(max(sizes))
        return b

    # ------------------------------------------------------------------ static (graph) mode: point pool + fixed-address buffers
    def _build_static(self):
        """Point pool of the whole image set (CSR, resident in HBM) and the fixed-address buffers of the captured step."""
        dev, npts = self.device, self.num_points
        n_img = int(self.images_all.shape[0])
        off = [torch.zeros(1, dtype=torch.int64, device=dev)]
        xyz = []
        total = 0
        for a in range(0, n_img, 8192):           # chunks keep the one-time nonzero() temporaries small
            imgs = self.images_all[a:a + 8192]
            b = batch_from_images(imgs, torch.zeros(imgs.shape[0], dtype=torch.int64, device=dev), npts)
            cnt = torch.bincount(b["coordinates"][:, 0].long(), minlength=imgs.shape[0])
            off.append(torch.cumsum(cnt, 0) + total)
            total += int(cnt.sum().item())
            xyz.append(b["features"])
        st = type("Static", (), {})()
        st.pool_off = torch.cat(off).to(_I32).contiguous()
        st.pool_xyz = torch.cat(xyz).contiguous() if total else torch.zeros((1, 3), device=dev)
        st.n_pool = n_img
        counts = (st.pool_off[1:] - st.pool_off[:-1]).cpu().numpy()
        sizes = [min(self.batch_size, len(self.indices_class[c])) for c in self.classes]
        # tight bound on the points of one sampled batch: the `size` largest clouds of every class
        bound = sum(int(np.sort(counts[self.indices_class[c]])[::-1][:n].sum()) for c, n in zip(self.classes, sizes))
        st.bound = max(256, (bound + 255) // 256 * 256)
        st.sizes, st.B = sizes, int(sum(sizes))
        st.ids = torch.zeros(max(st.B, 1), dtype=_I32, device=dev)
        st.coords = torch.zeros((st.bound, 4), dtype=_F32, device=dev)
        st.feats = torch.zeros((st.bound, 3), dtype=_F32, device=dev)
        st.pt_off = torch.zeros(st.B + 1, dtype=_I32, device=dev)
        st.n = torch.zeros(1, dtype=_I32, device=dev)
        st.err = torch.zeros(1, dtype=_I32, device=dev)
        st.labels = torch.tensor(np.repeat(self.classes, sizes), device=dev, dtype=_I32)
        st.cloud_group = torch.tensor(np.repeat(np.arange(self.G), sizes), device=dev, dtype=_I32)
        st.gco = torch.tensor(np.concatenate([[0], np.cumsum(sizes)]), device=dev, dtype=_I32)
        st.max_per = int(max(sizes))
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
        st.pin_ev[i].record()

    # ------------------------------------------------------------------ the step
    def _compute(self, rb, static):
        """Everything after the real batch exists: both passes, matching loss, backward into the points, SGD update."""
        dev, G, ipc, npts = self.device, self.G, self.ipc, self.num_points
        main = torch.cuda.current_stream(dev)
        B_r = int(rb["num_clouds"])
        xyz = self.image_syn[self.syn_rows].reshape(-1, 3)           # [G*ipc*npts, 3]; coordinates and features are the same tensor
        coords = torch.cat([self.syn_batch_col, xyz], 1)
        # cached layouts of the frozen weights are (re)built here, on the main stream, before the two passes fork
        from .ops import site_major
        net = self.network
        site_major(net.classifier.weight, 128, net.dense_extent[0] * net.dense_extent[1] * net.dense_extent[2])
        transposed_kernel(net.features[4].kernel)
        sr, ss = self.stream_real, self.stream_syn
        if os.environ.get("DD3D_SINGLE_STREAM"):    # experiment switch: both passes on the main stream
            sr = ss = main
        sr.wait_stream(main)
        ss.wait_stream(main)
        if not static:
            for t in rb.values():
                if torch.is_tensor(t):
                    t.record_stream(sr)
        with torch.cuda.stream(sr):
            real = _Pass(self.network, rb["coordinates"], B_r, G, rb["cloud_group"], rb["group_cloud_off"], rb.get("n_dev"), static,
                         rb.get("pt_off"), npts)
        with torch.cuda.stream(ss):
            syn = _Pass(self.network, coords, G * ipc, G, self.syn_group, self.syn_gco, None, static, self.syn_pt_off, npts)
        with torch.cuda.stream(sr):
            real.run(rb["features"], rb["labels"], False)
        with torch.cuda.stream(ss):
            syn.run(xyz, self.syn_labels, True)
        syn._max_per = ipc
        main.wait_stream(sr)
        main.wait_stream(ss)
        # only these entries of gw_syn depend on the synthetic points: IN weights, classifier weight and bias
        need = [False, True, False, False, True, False, True, True]
        loss_g, d = grouped_match_loss(syn.gw, real.gw, self.dist_opt, G, need)
        dfeats = syn.backward_to_points(d[1], d[4], d[6], d[7])
        grad = torch.zeros_like(self.image_syn)
        grad[self.syn_rows] = dfeats.reshape(G * ipc, npts, 3)
        self.last = {"per_class": loss_g, "grad": grad, "gw_real": real.gw, "gw_syn": syn.gw, "logits": (real.logits, syn.logits),
                     "conv1_maps": [(real.maps[4], real.M[1], real.cnt[1]), (syn.maps[4], syn.M[1], syn.cnt[1])], "passes": (real, syn)}
        if self._apply_update:
            # momentum_buf starts at zero, so the first step needs no special case (0.5 * 0 + g == g exactly)
            L.call("dd3d_sgd_momentum_clamp", L.ptr(self.image_syn), L.ptr(grad), L.ptr(self.momentum_buf), self.image_syn.numel(), self.lr,
                   self.momentum, 0, 0, 0.0, 27.0, L.stream())
        return loss_g.sum()

    def _static_body(self):
        """clamp + on-device batch assembly + the step, all on fixed addresses with device-side sizes (what the graph captures)."""
        st = self._static
        self.clamp()
        L.call("dd3d_gather_clouds", L.ptr(st.pool_xyz), L.ptr(st.pool_off), st.n_pool, L.ptr(st.ids), st.B, self.num_points, st.bound,
               L.ptr(st.coords), L.ptr(st.feats), L.ptr(st.pt_off), L.ptr(st.n), L.ptr(st.err), L.stream())
        rb = {"coordinates": st.coords, "features": st.feats, "labels": st.labels, "num_clouds": st.B, "cloud_group": st.cloud_group,
              "group_cloud_off": st.gco, "n_dev": st.n, "pt_off": st.pt_off}
        with L.ws_scope(self._scope):
            return self._compute(rb, True)

    def row_counts(self):
        """((M0,M1,M2) real, (M0,M1,M2) synthetic) of the last step (one device->host read in graph mode)."""
        real, syn = self.last["passes"]
        return tuple(tuple(int(x) for x in p.cm.counts[:3].tolist()) for p in (real, syn))

    def check(self):
        """Raise if any kernel of the last step flagged an error (graph mode defers these checks; eager mode raises in the step)."""
        real, syn = self.last["passes"]
        real.cm.check()
        syn.cm.check()
        if self._static is not None and int(self._static.err.item()) & 4:
            raise RuntimeError("real-batch assembly: sample index outside the pool, or the batch exceeds its point bound")

    def outer_step(self, real_batch=None, apply_update=True, force_eager=False):
        """One matching step (mnist_distill.py:309-369 without evaluation/logging).  Returns the summed matching loss (0-d device
        tensor; in graph mode it is a fixed buffer overwritten by the next step)."""
        self._apply_update = apply_update
        if self.use_graph and real_batch is None and apply_update:
            if self._static is None:
                self._build_static()
            self._upload_ids()
            if force_eager or self.steps_done == 0:
                # the first step runs eagerly: it sizes the workspaces and builds the cached weight layouts the graph will reuse
                loss = self._static_body()
            else:
                if self._graph is None:
                    self._graph = torch.cuda.CUDAGraph()
                    # thread_local: only this thread's calls are checked during capture (an RCCL watchdog thread polling its events
                    # must not invalidate it)
                    with torch.cuda.graph(self._graph, capture_error_mode="thread_local"):
                        self._graph_loss = self._static_body()
                    self._graph_last = self.last
                self._graph.replay()
                loss, self.last = self._graph_loss, self._graph_last
            self.steps_done += 1
            return loss
        self.clamp()
        main = torch.cuda.current_stream(self.device)
        if real_batch is not None:
            rb = real_batch
        elif self._prefetched is not None:
            rb, self._prefetched = self._prefetched, None
            main.wait_stream(self.stream_data)
        else:
            rb = self.sample_real()
        rb = dict(rb)
        rb["labels"] = rb["labels"].to(_I32)
        total = self._compute(rb, False)
        if apply_update:
            self.steps_done += 1
        if real_batch is None and not self.use_graph:
            with torch.cuda.stream(self.stream_data):
                self._prefetched = self.sample_real()
        return total
