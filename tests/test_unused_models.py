"""The reference's UNUSED model variants (models/MEConv.py:10-164: MEConv, MEConvImage).
CPU: constructors against the reference's own classes (tests/golden/g7_unused_models.npz, written by tests/golden/make_golden.py
importing models/MEConv.py over the ME restatement) and known-answer checks of the oracle's average pooling.
GPU: the HIP pooling kernels against the oracle, and MEConvImage(full_minkowski=True) -- the one configuration that runs in the
reference -- against the reference's own forward + backward.  MinkowskiEngine itself is absent: parity unpinned at that boundary."""
import os

import numpy as np
import pytest
import torch

from oracle import minkowski_ref as ME
from dd3d_amd.synthetic import stroke_digits
from tests.detgen import det_state_dict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G7 = os.path.join(ROOT, "tests", "golden", "g7_unused_models.npz")


def rel_err(a, b):
    a = torch.as_tensor(a).detach().double().cpu()
    b = torch.as_tensor(b).detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def test_unused_model_constructors_match_reference_golden():
    from dd3d_amd.models.MEConv import MEConv, MEConvImage
    g = np.load(G7)
    for name, cls in (("MEConv", MEConv), ("MEConvImage", MEConvImage)):
        for full in (False, True):
            net = cls(in_channel=3, out_channel=10, full_minkowski=full)
            tag = f"{name}.{'full' if full else 'dense'}"
            assert [n for n, _ in net.named_parameters()] == list(g[tag + ".names"])
            assert [str(tuple(p.shape)) for p in net.parameters()] == list(g[tag + ".shapes"])
            assert list(net.state_dict().keys()) == list(g[tag + ".sd_keys"])
    net = MEConvImage(in_channel=3, out_channel=10, full_minkowski=True)   # weight_initialization runs in this constructor only (:92)
    std = [net.features[i].kernel.std().item() for i in (0, 4, 8)]
    np.testing.assert_allclose(std, g["MEConvImage.kernel_std"], rtol=0.05)


def test_oracle_average_pooling_known_answers():
    X = torch.tensor([[1.0, 10.0], [3.0, 30.0], [5.0, 50.0], [7.0, 70.0]], dtype=torch.float64, requires_grad=True)
    kids = torch.tensor([[0, 2, -1], [1, -1, -1], [-1, 3, -1]])     # output 0 <- rows 0, 1; output 1 <- rows 2, 3; output 2 <- nothing
    out = ME.AvgPoolFn.apply(X, kids)
    assert torch.equal(out.detach(), torch.tensor([[2.0, 20.0], [6.0, 60.0], [0.0, 0.0]], dtype=torch.float64))
    out.sum().backward()
    assert torch.equal(X.grad, torch.full((4, 2), 0.5, dtype=torch.float64))   # every child gets g / (number of children)
    Y = torch.tensor([[1.0], [2.0], [6.0]], dtype=torch.float64, requires_grad=True)
    pooled = ME.GlobalAvgPoolFn.apply(Y, torch.tensor([0, 0, 2]), 3)
    assert torch.equal(pooled.detach(), torch.tensor([[1.5], [0.0], [6.0]], dtype=torch.float64))
    (pooled * torch.tensor([[2.0], [5.0], [3.0]], dtype=torch.float64)).sum().backward()
    assert torch.equal(Y.grad, torch.tensor([[1.0], [1.0], [3.0]], dtype=torch.float64))


# ------------------------------------------------------------------------------------------------ GPU
DEV = "cuda:0"


def _coords(B, pts, extent, seed, dims3=True):
    rng = np.random.default_rng(seed)
    rows = []
    for b in range(B):
        n = int(rng.integers(max(pts // 2, 1), pts + 1))
        c = rng.uniform(0, extent, size=(n, 3))
        if not dims3:
            c[:, 0] = 0
        rows.append(np.concatenate([np.full((n, 1), b), c], 1))
    return torch.tensor(np.concatenate(rows, 0), dtype=torch.float32)


@pytest.mark.gpu
@pytest.mark.parametrize("case", [dict(B=4, pts=200, extent=9, seed=1), dict(B=3, pts=250, extent=28, seed=2, dims3=False),
                                  dict(B=1, pts=1, extent=3, seed=3)])
@pytest.mark.parametrize("C", [128, 16])
def test_average_pooling_kernels_vs_oracle(case, C):
    from dd3d_amd.ops import AvgPoolFn, GlobalAvgPoolFn
    from dd3d_amd.sparse import CoordinateManager
    coords = _coords(**case)
    cm = CoordinateManager(coords.to(DEV), None)
    kids, ld = cm.kernel_map(1, 2, 2)
    parent, seg2 = cm.parent(2), cm.seg_off(2)
    cm.sync()
    M0, M1, B = cm.num_rows(1), cm.num_rows(2), cm.B
    C0, C1 = cm.coordinates(1).cpu(), cm.coordinates(2).cpu()
    ref_kids = ME.kernel_map(C0, C1, 2, 1)
    g = torch.Generator().manual_seed(4)
    X = torch.randn(M0, C, generator=g)
    x, xr = X.to(DEV).requires_grad_(), X.double().requires_grad_()
    out, ref = AvgPoolFn.apply(x, kids, ld, parent, M1), ME.AvgPoolFn.apply(xr, ref_kids)
    assert rel_err(out, ref) < 1e-6
    pooled, pref = GlobalAvgPoolFn.apply(out, seg2, B), ME.GlobalAvgPoolFn.apply(ref, C1[:, 0].long(), B)
    assert rel_err(pooled, pref) < 1e-6
    w = torch.randn(B, C, generator=g, dtype=torch.float64)
    (pref * w).sum().backward()
    (pooled * w.float().to(DEV)).sum().backward()
    assert rel_err(x.grad, xr.grad) < 1e-6


def _conv_maps(B, pts, extent, seed):
    from dd3d_amd.sparse import CoordinateManager
    coords = _coords(B=B, pts=pts, extent=extent, seed=seed)
    cm = CoordinateManager(coords.to(DEV), None)
    nbr, ld = cm.kernel_map(1, 1, 3)
    cm.sync()
    M = cm.num_rows(1)
    ref_nbr = ME.kernel_map(cm.coordinates(1).cpu(), cm.coordinates(1).cpu(), 3, 1)
    return nbr, ld, M, ref_nbr


ANY_SHAPES = ((27, 1, 128), (27, 128, 64), (27, 64, 32), (27, 32, 16), (1, 16, 1), (1, 128, 128), (27, 5, 7), (27, 100, 33))


@pytest.mark.gpu
@pytest.mark.parametrize("maps", [dict(B=3, pts=150, extent=7, seed=5), dict(B=30, pts=900, extent=24, seed=9)])
def test_generic_convolution_kernels_vs_oracle(maps):
    """Channel counts and kernel sizes that only the unused MEConv / MEConvImage stacks use (models/MEConv.py:25-39,102-130: 1 -> 128,
    128 -> 64 -> 32 -> 16 at kernel size 3, 16 -> 1 and 128 -> 128 at kernel size 1) plus two odd ones: forward, input gradient and kernel
    gradient of the hand-written kernels (conv_any.hip, fp32 MFMA) against the fp64 restatement."""
    from dd3d_amd.ops import sparse_conv_any
    nbr, ld, M, ref_nbr = _conv_maps(**maps)
    g = torch.Generator().manual_seed(6)
    for K, Cin, Cout in ANY_SHAPES:
        X, W = torch.randn(M, Cin, generator=g), torch.randn(K, Cin, Cout, generator=g) * 0.2
        x, w = X.to(DEV).requires_grad_(), W.to(DEV).requires_grad_()
        xr, wr = X.double().requires_grad_(), W.double().requires_grad_()
        Y = sparse_conv_any(x, w, nbr if K == 27 else None, ld if K == 27 else 0, M)
        Yr = ME.SparseConvFn.apply(xr, wr, ref_nbr if K == 27 else torch.arange(M)[None])
        assert rel_err(Y, Yr) < 1e-5, (K, Cin, Cout)
        gy = torch.randn(M, Cout, generator=g, dtype=torch.float64)
        (Yr * gy).sum().backward()
        (Y * gy.float().to(DEV)).sum().backward()
        assert rel_err(x.grad, xr.grad) < 1e-5 and rel_err(w.grad, wr.grad) < 1e-5, (K, Cin, Cout)


@pytest.mark.gpu
def test_generic_convolution_bound_sized_launch_is_bitwise_the_exact_one():
    """M as a bound with the true count on the device (what lets a caller capture the launch): rows below the count carry the same bits as
    the exact launch, rows beyond it are not touched, the kernel gradient is the same bits (its row chunks follow the true count); and
    the argument errors of the C-ABI."""
    from dd3d_amd import _lib as L
    nbr, ld, M, _ = _conv_maps(B=6, pts=300, extent=10, seed=12)
    K, Cin, Cout, bound = 27, 64, 32, M + 777
    g = torch.Generator().manual_seed(3)
    X = torch.randn(bound, Cin, generator=g).to(DEV)
    W = (torch.randn(K, Cin, Cout, generator=g) * 0.2).to(DEV)
    gY = torch.randn(bound, Cout, generator=g).to(DEV)
    nbr_b = torch.full((K, bound), -1, dtype=torch.int32, device=DEV)
    nbr_b[:, :M] = nbr[:, :M]
    nbr_b[:, M:] = 0                                         # stale entries beyond the count must not be read as neighbours
    m_dev = torch.tensor([M], dtype=torch.int32, device=DEV)
    ws = torch.empty(int(L.lib.dd3d_conv_any_bwd_weight_ws_bytes(K, Cin, Cout)), dtype=torch.uint8, device=DEV)
    st = L.stream()

    def run(Mx, md, table, ldx):
        Y = torch.full((bound, Cout), 7.0, device=DEV)
        dX = torch.full((bound, Cin), 7.0, device=DEV)
        dW = torch.empty((K, Cin, Cout), device=DEV)
        L.call("dd3d_conv_any_fwd", L.ptr(X), bound, Cin, L.ptr(W), Cout, L.ptr(table), ldx, K, Mx, L.ptr(md), L.ptr(Y), st)
        L.call("dd3d_conv_any_bwd_data", L.ptr(gY), Mx, L.ptr(md), Cout, L.ptr(W), Cin, L.ptr(table), ldx, K, L.ptr(dX), st)
        L.call("dd3d_conv_any_bwd_weight", L.ptr(X), Cin, L.ptr(gY), Cout, L.ptr(table), ldx, K, Mx, L.ptr(md), L.ptr(dW), L.ptr(ws), ws.numel(), st)
        return Y, dX, dW

    exact, bounded = run(M, None, nbr, ld), run(bound, m_dev, nbr_b, bound)
    for a, b in zip(exact, bounded):
        assert torch.equal(a, b)
    assert bool((bounded[0][M:] == 7.0).all()) and bool((bounded[1][M:] == 7.0).all())
    assert L.lib.dd3d_conv_any_max_channels() == 128
    Y = torch.empty((M, 200), device=DEV)
    for args in (("dd3d_conv_any_fwd", L.ptr(X), M, 129, L.ptr(W), Cout, L.ptr(nbr), ld, K, M, None, L.ptr(Y), st),           # too wide
                 ("dd3d_conv_any_fwd", L.ptr(X), M, Cin, L.ptr(W), Cout, None, 0, K, M, None, L.ptr(Y), st),                    # no table but K > 1
                 ("dd3d_conv_any_bwd_data", L.ptr(gY), M, None, Cout, L.ptr(W), Cin, L.ptr(nbr), ld, 8, L.ptr(Y), st)):         # even K has no mirror
        with pytest.raises(RuntimeError):
            L.call(*args)


def _plain_conv(X, W, nbr):
    Y = torch.zeros(nbr.shape[1], W.shape[2], dtype=X.dtype)
    for k in range(nbr.shape[0]):
        o = torch.nonzero(nbr[k] >= 0).reshape(-1)
        if o.numel():
            Y = Y.index_add(0, o, X[nbr[k][o]] @ W[k])
    return Y


@pytest.mark.gpu
@pytest.mark.parametrize("Cin,Cout", [(128, 128), (3, 128), (32, 16)])
def test_convolution_gradient_of_gradient_vs_fp64(Cin, Cout):
    """SparseConvTrueFn (the true-second-order research mode): the backward of the convolution is itself
    differentiable.  The gradient of a function of (dL/dX, dL/dW) with respect to X and W against plain fp64 torch autograd -- for the
    hot shapes (their own kernels) and a generic one."""
    from dd3d_amd.ops import SparseConvTrueFn
    nbr, ld, M, ref_nbr = _conv_maps(B=4, pts=200, extent=8, seed=21)
    g = torch.Generator().manual_seed(8)
    X, W = torch.randn(M, Cin, generator=g), torch.randn(27, Cin, Cout, generator=g) * 0.1
    r1, r2 = torch.randn(M, Cin, generator=g, dtype=torch.float64), torch.randn(27, Cin, Cout, generator=g, dtype=torch.float64)

    def second(x, w, conv, r1, r2):
        y = conv(x, w)
        gx, gw = torch.autograd.grad(0.5 * (y * y).sum(), (x, w), create_graph=True)
        return (gx, gw) + torch.autograd.grad((gx * r1).sum() + (gw * r2).sum() + (gx * gx).sum(), (x, w))

    x, w = X.to(DEV).requires_grad_(), W.to(DEV).requires_grad_()
    xr, wr = X.double().requires_grad_(), W.double().requires_grad_()
    got = second(x, w, lambda a, b: SparseConvTrueFn.apply(a, b, nbr, ld, M), r1.float().to(DEV), r2.float().to(DEV))
    ref = second(xr, wr, lambda a, b: _plain_conv(a, b, ref_nbr), r1, r2)
    for a, b in zip(got, ref):
        assert rel_err(a, b) < 2e-5


@pytest.mark.gpu
def test_meconvimage_full_minkowski_vs_reference_golden():
    """Forward + backward of MEConvImage(full_minkowski=True) on six stroke digits: the reference's own class (over the ME restatement)
    against this package's (HIP convolutions, instance norm, average pooling, global average pooling)."""
    from dd3d_amd.models.MEConv import MEConvImage, create_input_batch
    from dd3d_amd.utils.Mnist2D import Mnist2Dreal, get_mnist_dataloader
    g = np.load(G7)
    net = MEConvImage(in_channel=3, out_channel=10, full_minkowski=True)
    net.load_state_dict(det_state_dict(net, "net.MEConvImage"))
    net = net.to(DEV).train()
    imgs = torch.cat([stroke_digits(3, seed=300 + c) for c in range(2)])
    labels = torch.tensor([0, 0, 0, 1, 1, 1])
    for batch in get_mnist_dataloader(Mnist2Dreal(imgs, labels, False), imgs.shape[0]):
        logits = net(create_input_batch(batch, True, device=DEV, quantization_size=1))
        loss = torch.nn.CrossEntropyLoss()(logits, batch["labels"].to(DEV))
    grads = torch.autograd.grad(loss, list(net.parameters()))
    assert rel_err(logits, g["fwd.logits"]) < 1e-4
    assert loss.item() == pytest.approx(float(g["fwd.loss"]), rel=1e-4)
    np.testing.assert_allclose([t.double().norm().item() for t in grads], g["fwd.grad_norm"], rtol=2e-4)
    assert rel_err(grads[0], g["fwd.grad_kernel0"]) < 2e-4


@pytest.mark.gpu
def test_unused_models_fail_where_the_reference_fails():
    """MEConv cannot run in the reference either: its dense head reads an attribute the constructor never sets (models/MEConv.py:18,76),
    its full-Minkowski head feeds 16 channels into Linear(128, 10) (:21,82).  The feature stack itself (1 -> 128 -> 64 -> 32 -> 16,
    average pooling down to tensor stride 16) runs."""
    from dd3d_amd.models.MEConv import MEConv
    from dd3d_amd.sparse import TensorField
    coords = _coords(B=2, pts=200, extent=20, seed=8)
    feats = torch.ones(coords.shape[0], 1)
    for full, err in ((False, AttributeError), (True, RuntimeError)):
        net = MEConv(in_channel=3, out_channel=10, full_minkowski=full).to(DEV)
        feat_out = net.features[:16](TensorField(coordinates=coords.to(DEV), features=feats.to(DEV)).sparse())
        assert feat_out.F.shape[1] == 16 and feat_out.tensor_stride == 16 and bool(torch.isfinite(feat_out.F).all())
        with pytest.raises(err):
            net(TensorField(coordinates=coords.to(DEV), features=feats.to(DEV)))
