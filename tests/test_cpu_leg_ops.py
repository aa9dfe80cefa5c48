"""The op-level entry points of the oracle's C++/OpenMP leg (oracle/cpu/dd3d_cpu.h, numpy fronts in oracle/cpu_leg.py) against the torch
restatement that the golden fixtures pin (oracle/minkowski_ref.py, oracle/ref_model.py): coordinate maps bit for bit, feature operators and
the gradient distance in fp64 to rounding, the fp32 variants to fp32 rounding.  The two legs share no code.  CPU only."""
import ctypes
import os
import re
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from detgen import det_tensor  # noqa: E402
from oracle import cpu_leg as CL  # noqa: E402
from oracle import minkowski_ref as ME  # noqa: E402
from oracle import ref_model as R  # noqa: E402
from dd3d_amd.synthetic import shell_clouds, stroke_digits  # noqa: E402

HDR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle", "cpu", "dd3d_cpu.h")


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


def _clouds(kind):
    """(coords [N,3] already divided by the voxel size, pt_off, B): flat digit clouds (voxel 1) or volumetric shells (voxel 0.05, duplicates)."""
    if kind == "digits":
        cl = CL.image_clouds(stroke_digits(7, seed=3).numpy())
        cl[2] = cl[2][:1]                                     # a one-voxel cloud
        q = 1.0
    else:
        cl = [c.double().numpy() for c in shell_clouds(5, 400, seed=9)]
        cl[1] = np.concatenate([cl[1], cl[1][:50]])           # repeated points: several points per voxel
        q = 0.05
    off = np.zeros(len(cl) + 1, np.int32)
    off[1:] = np.cumsum([c.shape[0] for c in cl])
    pts = np.concatenate(cl)
    return (pts.astype(np.float32) / np.float32(q)).astype(np.float64), pts, off, len(cl)


def _maps(kind):
    coords, pts, off, B = _clouds(kind)
    C0, inv, seg0 = CL.quantize_unique(coords, off)
    C1, par1, seg1 = CL.stride_coords(C0, B, 2)
    C2, par2, seg2 = CL.stride_coords(C1, B, 4)
    return dict(coords=coords, pts=pts, off=off, B=B, C0=C0, inv=inv, seg0=seg0, C1=C1, par1=par1, seg1=seg1, C2=C2, par2=par2, seg2=seg2,
                nbr0=CL.kernel_map(C0, 1), nbr1=CL.kernel_map(C1, 2), kid1=CL.pool_map(C0, C1, 1), kid2=CL.pool_map(C1, C2, 2))


def test_header_declares_what_the_library_exports():
    """Every function oracle/cpu/dd3d_cpu.h declares is exported by libdd3d_cpu.so and bound by the Python front."""
    text = open(HDR).read()
    plain = set(re.findall(r"\b(dd3d_cpu_[a-z0-9_]+)\(", text))
    macro = set(re.findall(r"\b(dd3d_cpu_[a-z0-9_]+_)##SFX\(", text))   # declared once per arithmetic type
    want = set(plain) | {m + s for m in macro for s in ("f64", "f32")}
    assert {"dd3d_cpu_" + n for n in CL.OP_NAMES} <= want
    assert {f"dd3d_cpu_{n}_{s}" for n in CL.FEATURE_OP_NAMES for s in ("f64", "f32")} <= want
    lib = CL.lib()
    for n in sorted(want):
        assert isinstance(getattr(lib, n), ctypes._CFuncPtr), n
    assert len(want) == 39


@pytest.mark.parametrize("kind", ["digits", "shells"])
def test_maps_are_the_restatements_bit_for_bit(kind):
    m = _maps(kind)
    B = m["B"]
    bcol = np.repeat(np.arange(B), np.diff(m["off"]))[:, None].astype(np.float64)
    C0, inv, _ = ME.quantize_unique(torch.from_numpy(np.concatenate([bcol, m["coords"]], 1)))
    assert np.array_equal(m["C0"], C0.numpy()) and np.array_equal(m["inv"], inv.numpy())
    C1, par1 = ME.stride_map(C0, 2)
    C2, par2 = ME.stride_map(C1, 4)
    assert np.array_equal(m["C1"], C1.numpy()) and np.array_equal(m["par1"], par1.numpy())
    assert np.array_equal(m["C2"], C2.numpy()) and np.array_equal(m["par2"], par2.numpy())
    for C, seg in ((C0, m["seg0"]), (C1, m["seg1"]), (C2, m["seg2"])):
        assert np.array_equal(seg, np.concatenate([[0], np.cumsum(np.bincount(C[:, 0].numpy(), minlength=B))]))
    assert np.array_equal(m["nbr0"], ME.kernel_map(C0, C0, 3, 1).numpy())
    assert np.array_equal(m["nbr1"], ME.kernel_map(C1, C1, 3, 2).numpy())
    assert np.array_equal(m["kid1"], ME.kernel_map(C0, C1, 2, 1).numpy())
    assert np.array_equal(m["kid2"], ME.kernel_map(C1, C2, 2, 2).numpy())
    assert (m["nbr0"] >= 0).sum() > m["C0"].shape[0]   # neighbours beyond the centre exist
    ext = (1, 7, 7) if kind == "digits" else (12, 12, 12)
    site = CL.dense_sites(m["C2"], 4, ext)
    c2 = m["C2"].astype(np.int64)
    rel_ = (c2[:, 1:] - c2[:, 1:].min(0)) // 4        # models/MEConv.py:251-254: the minimum corner of the WHOLE batch
    assert np.array_equal(site, (rel_[:, 0] * ext[1] + rel_[:, 1]) * ext[2] + rel_[:, 2])
    with pytest.raises(RuntimeError):
        CL.dense_sites(m["C2"], 4, (1, 2, 2) if kind == "digits" else (2, 2, 2))
    far = m["C0"].copy()
    far[0, 2] = 40000
    with pytest.raises(ValueError, match="16-bit"):
        CL.kernel_map(far, 1)


@pytest.mark.parametrize("kind", ["digits", "shells"])
def test_feature_operators_vs_the_restatement(kind):
    m = _maps(kind)
    B, M0, M1, M2 = m["B"], m["C0"].shape[0], m["C1"].shape[0], m["C2"].shape[0]
    t = lambda a: torch.from_numpy(np.asarray(a))
    inv, nbr0, nbr1, kid1 = t(m["inv"]).long(), t(m["nbr0"]).long(), t(m["nbr1"]).long(), t(m["kid1"]).long()
    # voxel mean, forward and backward
    feats = t(m["pts"]).double().requires_grad_()
    F0 = ME.VoxelMeanFn.apply(feats, inv, M0)
    gF = det_tensor((M0, 3), "ops.gF", 1.0).double()
    (dfeats,) = torch.autograd.grad(F0, feats, gF)
    assert rel(CL.voxel_mean_fwd(m["pts"], m["inv"], M0), F0.detach()) < 1e-14
    assert rel(CL.voxel_mean_bwd(gF.numpy(), m["inv"]), dfeats) < 1e-14
    # convolutions 3 -> 128 and 128 -> 128 (and the 3-channel input gradient: the plain loop)
    for (X, W, nbr, nb, M) in ((F0.detach(), det_tensor((27, 3, 128), "ops.W0", 0.2).double(), nbr0, m["nbr0"], M0),
                               (det_tensor((M1, 128), "ops.X1", 1.0).double(), det_tensor((27, 128, 128), "ops.W1", 0.05).double(), nbr1, m["nbr1"], M1)):
        X, W = X.clone().requires_grad_(), W.clone().requires_grad_()
        Y = ME.SparseConvFn.apply(X, W, nbr)
        g = det_tensor((M, 128), "ops.gY%d" % X.shape[1], 1.0).double()
        dX, dW = torch.autograd.grad(Y, (X, W), g)
        assert rel(CL.conv_fwd(X.detach().numpy(), W.detach().numpy(), nb), Y.detach()) < 1e-13
        assert rel(CL.conv_bwd_data(g.numpy(), W.detach().numpy(), nb), dX) < 1e-13
        assert rel(CL.conv_bwd_weight(X.detach().numpy(), g.numpy(), nb), dW) < 1e-13
        wide = np.full((27, M + 5), -1, np.int32)           # a table with a leading dimension beyond its rows
        wide[:, :M] = nb
        assert np.array_equal(CL.conv_fwd(X.detach().numpy(), W.detach().numpy(), wide, M=M), CL.conv_fwd(X.detach().numpy(), W.detach().numpy(), nb))
        assert rel(CL.conv_fwd(X.detach().numpy(), W.detach().numpy(), nb, dt=np.float32), Y.detach()) < 2e-5
    # instance norm
    Y0 = det_tensor((M0, 128), "ops.Y0", 1.0).double().requires_grad_()
    bidx = t(m["C0"][:, 0]).long()
    xh = ME.InstanceNormFn.apply(Y0, bidx, B)
    gx = det_tensor((M0, 128), "ops.gx", 1.0).double()
    (dY0,) = torch.autograd.grad(xh, Y0, gx)
    xhat, inv_std = CL.inorm_fwd(Y0.detach().numpy(), m["seg0"])
    assert rel(xhat, xh.detach()) < 1e-12
    assert float(np.abs(xhat[m["seg0"][2]:m["seg0"][3]]).max()) == 0.0 if kind == "digits" else True   # the one-voxel cloud: 0 * 1e4 = 0
    assert rel(CL.inorm_bwd(gx.numpy(), xhat, inv_std, m["seg0"]), dY0) < 1e-12
    # affine + ReLU + max pool
    w, b = det_tensor((128,), "ops.w", 1.0).double() + 1.0, det_tensor((128,), "ops.b", 0.3).double()
    xin = xh.detach().clone().requires_grad_()
    P, arg = ME.MaxPoolFn.apply(torch.relu(xin * w + b), kid1)
    gp = det_tensor((M1, 128), "ops.gp", 1.0).double()
    (gxin,) = torch.autograd.grad(P, xin, gp)
    out, arg_c = CL.affine_relu_pool_fwd(xhat, w.numpy(), b.numpy(), m["kid1"])
    assert rel(out, P.detach()) < 1e-13 and np.array_equal(arg_c, arg.numpy())
    ga, gxh = CL.affine_relu_pool_bwd(gp.numpy(), arg_c, m["par1"], xhat, w.numpy(), b.numpy())
    assert rel(gxh, gxin) < 1e-13 and rel(ga * w.numpy(), gxin) < 1e-13
    # dense() + Linear, cross entropy
    ext = (1, 7, 7) if kind == "digits" else (12, 12, 12)
    S, J = ext[0] * ext[1] * ext[2], 6
    site = CL.dense_sites(m["C2"], 4, ext)
    F2 = det_tensor((M2, 128), "ops.F2", 1.0).double().requires_grad_()
    Wc, bc = det_tensor((J, 128 * S), "ops.Wc", 0.05).double().requires_grad_(), det_tensor((J,), "ops.bc", 0.1).double()
    dense = torch.zeros(B, S, 128, dtype=torch.float64).index_put((t(m["C2"][:, 0]).long(), t(site).long()), F2).permute(0, 2, 1)   # [B, C, sites]
    logits = dense.reshape(B, -1) @ Wc.t() + bc
    labels = torch.arange(B) % J
    ce = torch.nn.functional.cross_entropy(logits, labels)
    dl, dF2, dWc = torch.autograd.grad(ce, (logits, F2, Wc), retain_graph=True)
    lg = CL.sparse_linear_fwd(F2.detach().numpy(), site, m["seg2"], S, Wc.detach().numpy(), bc.numpy())
    assert rel(lg, logits.detach()) < 1e-13
    loss_c, dl_c = CL.softmax_ce(lg, labels.numpy())
    assert abs(loss_c - ce.item()) < 1e-13 * abs(ce.item()) and rel(dl_c, dl) < 1e-12
    assert rel(CL.sparse_linear_bwd_x(dl_c, site, m["seg2"], S, Wc.detach().numpy()), dF2) < 1e-12
    assert rel(CL.sparse_linear_bwd_w(dl_c, F2.detach().numpy(), site, m["seg2"], S), dWc) < 1e-12


@pytest.mark.parametrize("metric", ["ours", "modmse", "mse", "cos"])
def test_match_loss_vs_the_restatement_all_eight_gradients(metric):
    """distillation_loss.match_loss (oracle/ref_model.py, pinned by tests/golden/g1_match_loss.npz) and its gradient with respect to ALL eight
    entries of gw_syn (the class step needs four of them)."""
    shapes = [(27, 3, 128), (1, 128), (1, 128), (27, 128, 128), (1, 128), (1, 128), (10, 128 * 49), (10,)]
    real = [det_tensor(s, f"ml.real.{i}", 0.05).double() for i, s in enumerate(shapes)]
    syn = [det_tensor(s, f"ml.syn.{i}", 0.05).double().requires_grad_() for i, s in enumerate(shapes)]
    v = R.match_loss(syn, real, metric, "cpu", False)
    grads = torch.autograd.grad(v, syn)
    loss, d = CL.match_loss([s.detach().numpy() for s in syn], [r.numpy() for r in real], metric)
    assert abs(loss - v.item()) < 1e-12 * abs(v.item())
    for i in range(8):
        assert rel(d[i], grads[i]) < 1e-11, i
    l32, _ = CL.match_loss([s.detach().numpy() for s in syn], [r.numpy() for r in real], metric, dt=np.float32)
    assert abs(l32 - v.item()) < 1e-4 * abs(v.item())
