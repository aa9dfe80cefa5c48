"""GPU parity tests, op by op: every C-ABI entry point of libdd3d_hip.so against the CPU oracle (oracle/minkowski_ref.py) on the
same seeded inputs.  Integer outputs (coordinates, maps, pair lists) must be bit-exact; fp32 outputs are compared with an fp64
oracle at rel 2e-5 of the tensor's max magnitude."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import minkowski_ref as ME


def dev():
    return torch.device("cuda:0")


def rel_err(a, b):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def make_coords(B, pts, extent, seed, dims3=False, neg=False):
    rng = np.random.default_rng(seed)
    rows = []
    for b in range(B):
        n = int(rng.integers(1, pts + 1))
        c = rng.uniform(-extent if neg else 0, extent, size=(n, 3))
        if not dims3:
            c[:, 0] = 0
        rows.append(np.concatenate([np.full((n, 1), b), c], 1))
    return torch.tensor(np.concatenate(rows, 0), dtype=torch.float32)


def manager(coords, B=None):
    from dd3d_amd.sparse import CoordinateManager
    return CoordinateManager(coords.to(dev()), B)


CASES = [dict(B=5, pts=60, extent=12, seed=0), dict(B=3, pts=250, extent=28, seed=1),
         dict(B=7, pts=100, extent=9, seed=2, dims3=True, neg=True), dict(B=1, pts=1, extent=3, seed=3),
         dict(B=40, pts=300, extent=28, seed=4)]


@pytest.mark.parametrize("case", CASES)
def test_coordinate_maps_bit_exact(case):
    coords = make_coords(**case)
    cm = manager(coords)
    C0, inv, uidx = ME.quantize_unique(coords)
    M0 = cm.num_rows(1)
    assert M0 == C0.shape[0]
    assert torch.equal(cm.coordinates(1).cpu(), C0)
    assert torch.equal(cm.levels[1].inv[:coords.shape[0]].cpu().long(), inv)
    assert torch.equal(cm.levels[1].uidx[:M0].cpu().long(), uidx)
    off, pts = ME.voxel_csr(inv, M0)
    assert torch.equal(cm.vox_off[:M0 + 1].cpu().long(), off)
    assert torch.equal(cm.vox_pts[:coords.shape[0]].cpu().long(), pts)
    Cprev = C0
    for ts in (2, 4):
        Cs, parent = ME.stride_map(Cprev, ts)
        assert cm.num_rows(ts) == Cs.shape[0]
        assert torch.equal(cm.coordinates(ts).cpu(), Cs)
        assert torch.equal(cm.parent(ts)[:Cprev.shape[0]].cpu().long(), parent)
        Cprev = Cs
    for (si, so, k) in ((1, 1, 3), (2, 2, 3), (1, 2, 2), (2, 4, 2)):
        nbr, ld = cm.kernel_map(si, so, k)
        Mo = cm.num_rows(so)
        ref = ME.kernel_map(cm.coordinates(si).cpu(), cm.coordinates(so).cpu(), k, si)
        assert torch.equal(nbr[:, :Mo].cpu().long(), ref)
        ins, outs, offs = cm.pairs(si, so, k)
        rins, routs, roffs = ME.pairs_from_nbr(ref)
        assert torch.equal(offs.cpu().long(), roffs)
        assert torch.equal(ins.cpu().long(), rins) and torch.equal(outs.cpu().long(), routs)
    B = case["B"]
    seg = cm.seg_off(1).cpu().long()
    b = C0[:, 0].long()
    assert torch.equal(seg, torch.searchsorted(b, torch.arange(B + 1)))
    cm.sync()


def test_lazy_pyramid_single_sync_matches_eager():
    coords = make_coords(B=6, pts=120, extent=20, seed=9)
    cm = manager(coords, 6)
    k0 = cm.kernel_map(1, 1, 3)
    p0 = cm.kernel_map(1, 2, 2)
    k1 = cm.kernel_map(2, 2, 3)
    p1 = cm.kernel_map(2, 4, 2)
    assert all(lv.M is None for lv in cm.levels.values())  # nothing read back yet
    cm.sync()
    cm2 = manager(coords, 6)
    for ts in (1, 2, 4):
        cm2.num_rows(ts)  # eager: sync at every level
        assert torch.equal(cm.coordinates(ts), cm2.coordinates(ts))
    for (key, (nbr, ld)) in (((1, 1, 3), k0), ((1, 2, 2), p0), ((2, 2, 3), k1), ((2, 4, 2), p1)):
        n2, _ = cm2.kernel_map(*key)
        Mo = cm.num_rows(key[1])
        assert torch.equal(nbr[:, :Mo], n2[:, :Mo])


def test_empty_and_duplicate_inputs():
    cm = manager(torch.zeros((0, 4)), 0)
    assert cm.num_rows(1) == 0 and cm.num_rows(2) == 0
    c = torch.tensor([[0, 0, 3.5, 4.5]] * 70 + [[1, 0, 3.5, 4.5]] * 3)
    cm = manager(c)
    assert cm.num_rows(1) == 2
    assert cm.levels[1].inv[:73].cpu().tolist() == [0] * 70 + [1] * 3
    assert cm.vox_pts[:73].cpu().tolist() == list(range(73))


def test_unsorted_batch_index_is_reported():
    c = torch.tensor([[1, 0, 1.0, 1.0], [0, 0, 2.0, 2.0]])
    cm = manager(c, 2)
    cm.seg_off(1)
    with pytest.raises(RuntimeError):
        cm.sync()


@pytest.mark.parametrize("case", CASES[:3])
def test_voxel_mean(case):
    from dd3d_amd.ops import voxel_mean
    coords = make_coords(**case)
    g = torch.Generator().manual_seed(5)
    feats = torch.randn(coords.shape[0], 3, generator=g)
    cm = manager(coords)
    f = feats.to(dev()).requires_grad_()
    F0 = voxel_mean(f, cm)
    _, inv, _ = ME.quantize_unique(coords)
    fr = feats.double().requires_grad_()
    R0 = ME.VoxelMeanFn.apply(fr, inv, cm.num_rows(1))
    assert rel_err(F0, R0) < 1e-6
    w = torch.randn(R0.shape, generator=g, dtype=torch.float64)
    (R0 * w).sum().backward()
    (F0 * w.float().to(dev())).sum().backward()
    assert rel_err(f.grad, fr.grad) < 1e-6


def _conv_case(case, Cin, seed=0, stride=1):
    coords = make_coords(**case)
    cm = manager(coords)
    M = cm.num_rows(stride)
    nbr, ld = cm.kernel_map(stride, stride, 3)
    g = torch.Generator().manual_seed(seed)
    X = torch.randn(M, Cin, generator=g)
    W = torch.randn(27, Cin, 128, generator=g) * 0.1
    return cm, M, nbr, ld, X, W


@pytest.mark.parametrize("case", CASES[:3] + [CASES[4]])
@pytest.mark.parametrize("Cin", [3, 128])
def test_conv_forward_backward(case, Cin):
    from dd3d_amd.ops import SparseConvFn
    cm, M, nbr, ld, X, W = _conv_case(case, Cin)
    x = X.to(dev()).requires_grad_()
    w = W.to(dev()).requires_grad_()
    Y = SparseConvFn.apply(x, w, nbr, ld, M)
    ref_nbr = ME.kernel_map(cm.coordinates(1).cpu(), cm.coordinates(1).cpu(), 3, 1)
    xr, wr = X.double().requires_grad_(), W.double().requires_grad_()
    Yr = ME.SparseConvFn.apply(xr, wr, ref_nbr)
    assert rel_err(Y, Yr) < 2e-5
    gy = torch.randn(Yr.shape, generator=torch.Generator().manual_seed(3), dtype=torch.float64)
    (Yr * gy).sum().backward()
    (Y * gy.float().to(dev())).sum().backward()
    assert rel_err(x.grad, xr.grad) < 2e-5
    assert rel_err(w.grad, wr.grad) < 2e-5


def test_first_layer_conv_on_dense_cubes_every_offset_present():
    """Cin = 3 kernels where rows have far more than nine neighbours (filled 6^3 cubes: 19 of 27 on average, all 27 inside): the forward kernel's last 9-step chunk (im2col columns
    64..80), all three output tiles of the kernel gradient and the input gradient's second and third batch of nine offsets carry real
    data (stroke digits and sparse clouds leave them almost empty).  Against the fp64 reference, tolerance as everywhere."""
    from dd3d_amd.ops import SparseConvFn
    cm, M, nbr, ld, X, W = _conv_case(dict(B=3, pts=3000, extent=6, seed=11, dims3=True), 3)
    assert float((nbr[:, :M] >= 0).sum()) / M > 12  # neighbours per row
    x = X.to(dev()).requires_grad_()
    w = W.to(dev()).requires_grad_()
    Y = SparseConvFn.apply(x, w, nbr, ld, M)
    ref_nbr = ME.kernel_map(cm.coordinates(1).cpu(), cm.coordinates(1).cpu(), 3, 1)
    xr, wr = X.double().requires_grad_(), W.double().requires_grad_()
    Yr = ME.SparseConvFn.apply(xr, wr, ref_nbr)
    assert rel_err(Y, Yr) < 2e-5
    gy = torch.randn(Yr.shape, generator=torch.Generator().manual_seed(3), dtype=torch.float64)
    (Yr * gy).sum().backward()
    (Y * gy.float().to(dev())).sum().backward()
    assert rel_err(x.grad, xr.grad) < 2e-5
    assert rel_err(w.grad, wr.grad) < 2e-5


def test_conv_bwd_weight_grouped_and_deterministic():
    from dd3d_amd.ops import conv_bwd_weight
    cm, M, nbr, ld, X, W = _conv_case(dict(B=40, pts=300, extent=28, seed=4), 128)
    seg = cm.seg_off(1)
    gco = [0, 7, 7, 25, 40]  # 4 groups of clouds, one of them empty
    gro = seg[torch.tensor(gco, device=dev())].contiguous()
    dY = torch.randn(M, 128, generator=torch.Generator().manual_seed(8))
    x, dy = X.to(dev()), dY.to(dev())
    dW = conv_bwd_weight(x, dy, nbr, ld, M, gro, 4)
    dW2 = conv_bwd_weight(x, dy, nbr, ld, M, gro, 4)
    assert torch.equal(dW, dW2)  # bitwise reproducible
    ref_nbr = ME.kernel_map(cm.coordinates(1).cpu(), cm.coordinates(1).cpu(), 3, 1)
    rows = gro.cpu().tolist()
    for gi in range(4):
        a, b = rows[gi], rows[gi + 1]
        ref = torch.zeros(27, 128, 128, dtype=torch.float64)
        for k in range(27):
            o = torch.nonzero(ref_nbr[k, a:b] >= 0).reshape(-1) + a
            if o.numel():
                ref[k] = X.double()[ref_nbr[k][o]].t() @ dY.double()[o]
        if a == b:
            assert float(dW[gi].abs().max()) == 0.0
        else:
            assert rel_err(dW[gi], ref) < 2e-5


@pytest.mark.parametrize("case", CASES[:3])
def test_instance_norm(case):
    from dd3d_amd.ops import InstanceNormFn
    coords = make_coords(**case)
    cm = manager(coords)
    M, B = cm.num_rows(1), cm.B
    g = torch.Generator().manual_seed(1)
    X = torch.randn(M, 128, generator=g) * 3 + 1
    x = X.to(dev()).requires_grad_()
    out = InstanceNormFn.apply(x, cm.seg_off(1), B)
    xr = X.double().requires_grad_()
    b = cm.coordinates(1)[:, 0].cpu().long()
    ref = ME.InstanceNormFn.apply(xr, b, B)
    assert rel_err(out, ref) < 2e-5
    gy = torch.randn(M, 128, generator=g, dtype=torch.float64)
    (ref * gy).sum().backward()
    (out * gy.float().to(dev())).sum().backward()
    assert rel_err(x.grad, xr.grad) < 5e-5


def test_instance_norm_single_voxel_and_empty_cloud():
    from dd3d_amd.ops import InstanceNormFn
    coords = torch.tensor([[0, 0, 1.0, 1.0], [2, 0, 1.0, 1.0], [2, 0, 5.0, 1.0]])  # cloud 1 is empty
    cm = manager(coords, 3)
    X = torch.tensor([[3.0] * 128, [1.0] * 128, [2.0] * 128])
    out = InstanceNormFn.apply(X.to(dev()), cm.seg_off(1), 3).cpu()
    assert float(out[0].abs().max()) == 0.0
    np.testing.assert_allclose(out[1:].numpy(), np.array([[-1.0] * 128, [1.0] * 128]), rtol=1e-5)


@pytest.mark.parametrize("case", CASES[:3])
def test_max_pool(case):
    from dd3d_amd.ops import MaxPoolFn
    coords = make_coords(**case)
    cm = manager(coords)
    M, Mo = cm.num_rows(1), cm.num_rows(2)
    kids, ld = cm.kernel_map(1, 2, 2)
    g = torch.Generator().manual_seed(2)
    X = torch.randn(M, 128, generator=g) - 2.0  # mostly negative: absent children must not act as zeros
    x = X.to(dev()).requires_grad_()
    out = MaxPoolFn.apply(x, kids, ld, cm.parent(2), Mo)
    rk = ME.kernel_map(cm.coordinates(1).cpu(), cm.coordinates(2).cpu(), 2, 1)
    xr = X.double().requires_grad_()
    ref, _ = ME.MaxPoolFn.apply(xr, rk)
    assert torch.equal(out.cpu().double(), ref.detach())
    gy = torch.randn(Mo, 128, generator=g, dtype=torch.float64)
    (ref * gy).sum().backward()
    (out * gy.float().to(dev())).sum().backward()
    assert torch.equal(x.grad.cpu().double(), xr.grad.float().double())


def _block_oracle(Y, w, b, bidx, B, kids):
    xh = ME.InstanceNormFn.apply(Y, bidx, B)
    a = torch.relu(xh * w + b)
    out, _ = ME.MaxPoolFn.apply(a, kids)
    return out


@pytest.mark.parametrize("case", CASES[:3])
def test_fused_norm_act_pool_first_and_second_order(case):
    """NormActPoolFn against the op-by-op oracle, including the create_graph path: d/dY of <gw_weight, t>."""
    from dd3d_amd.ops import NormActPoolFn, _single_group
    coords = make_coords(**case)
    cm = manager(coords)
    M, Mo, B = cm.num_rows(1), cm.num_rows(2), cm.B
    kids, ld = cm.kernel_map(1, 2, 2)
    g = torch.Generator().manual_seed(4)
    Y0 = torch.randn(M, 128, generator=g) * 2
    w0 = 1 + 0.2 * torch.randn(1, 128, generator=g)
    b0 = 0.2 * torch.randn(1, 128, generator=g)
    gp = torch.randn(Mo, 128, generator=g, dtype=torch.float64)
    t = torch.randn(1, 128, generator=g, dtype=torch.float64)
    y, w, b = (v.to(dev()).requires_grad_() for v in (Y0, w0, b0))
    out = NormActPoolFn.apply(y, w, b, cm.seg_off(1), B, kids, ld, cm.parent(2), Mo, _single_group(M, dev()), 1)
    yr, wr, br = (v.double().requires_grad_() for v in (Y0, w0, b0))
    bidx = cm.coordinates(1)[:, 0].cpu().long()
    rk = ME.kernel_map(cm.coordinates(1).cpu(), cm.coordinates(2).cpu(), 2, 1)
    ref = _block_oracle(yr, wr, br, bidx, B, rk)
    assert rel_err(out, ref) < 2e-5
    gy, gw, gb = torch.autograd.grad((out * gp.float().to(dev())).sum(), [y, w, b], create_graph=True)
    ry, rw, rb = torch.autograd.grad((ref * gp).sum(), [yr, wr, br], create_graph=True)
    assert rel_err(gy, ry) < 5e-5 and rel_err(gw, rw) < 5e-5 and rel_err(gb, rb) < 5e-5
    assert not ry.requires_grad  # oracle: a constant, as out of MinkowskiEngine's native backward
    (g2,) = torch.autograd.grad((gw * t.float().to(dev())).sum(), [y])
    (r2,) = torch.autograd.grad((rw * t).sum(), [yr])
    assert rel_err(g2, r2) < 5e-5


@pytest.mark.parametrize("case", CASES + [dict(B=4, pts=2500, extent=28, seed=8, dims3=True)])
def test_one_launch_norm_act_pool_is_bitwise_the_two_launches(case):
    """dd3d_inorm_arp_fwd == dd3d_inorm_fwd followed by dd3d_affine_relu_pool_fwd, bit for bit (clouds larger than the LDS cache too)."""
    from dd3d_amd import _lib as L
    coords = make_coords(**case)
    cm = manager(coords)
    M, Mo, B = cm.num_rows(1), cm.num_rows(2), cm.B
    kids, ld = cm.kernel_map(1, 2, 2)
    g = torch.Generator().manual_seed(9)
    Y = (torch.randn(M, 128, generator=g) * 2 - 0.5).to(dev())
    w = (1 + 0.2 * torch.randn(128, generator=g)).to(dev())
    b = (0.2 * torch.randn(128, generator=g)).to(dev())
    st = L.stream()

    def bufs():
        return (torch.full((M, 128), 7.0, device=dev()), torch.full((B, 128), 7.0, device=dev()), torch.full((Mo, 128), 7.0, device=dev()),
                torch.full((Mo, 128), -3, device=dev(), dtype=torch.int32))
    xa, ia, pa, aa = bufs()
    L.call("dd3d_inorm_fwd", L.ptr(Y), L.ptr(cm.seg_off(1)), B, 128, L.ptr(xa), L.ptr(ia), st)
    L.call("dd3d_affine_relu_pool_fwd", L.ptr(xa), L.ptr(w), L.ptr(b), 1, 128, L.ptr(kids), ld, 8, Mo, None, L.ptr(pa), L.ptr(aa), st)
    xb, ib, pb, ab = bufs()
    L.call("dd3d_inorm_arp_fwd", L.ptr(Y), L.ptr(cm.seg_off(1)), B, 128, L.ptr(xb), L.ptr(ib), L.ptr(w), L.ptr(b), L.ptr(kids), ld, 8,
           L.ptr(cm.seg_off(2)), L.ptr(pb), L.ptr(ab), st)
    torch.cuda.synchronize()
    for u, v in ((xa, xb), (ia, ib), (pa, pb), (aa, ab)):
        assert torch.equal(u, v)


@pytest.mark.parametrize("case", CASES + [dict(B=4, pts=2500, extent=28, seed=8, dims3=True)])
def test_norm_act_pool_without_stored_xhat_is_bitwise_the_stored_form(case):
    """dd3d_inorm_arp_fwd_y / dd3d_arp_inorm_bwd_y (what the step runs: Y, mean and inv_std instead of normalised rows in memory) against
    dd3d_inorm_arp_fwd / dd3d_arp_inorm_bwd: inv_std, pooled values, winners, and every backward output (first order with the affine
    gradients, second order with ga_prev / gw_extra) bit for bit; mean = the per-cloud column mean."""
    from dd3d_amd import _lib as L
    coords = make_coords(**case)
    cm = manager(coords)
    M, Mo, B = cm.num_rows(1), cm.num_rows(2), cm.B
    kids, ld = cm.kernel_map(1, 2, 2)
    seg, seg_o, parent = cm.seg_off(1), cm.seg_off(2), cm.parent(2)
    g = torch.Generator().manual_seed(19)
    Y = (torch.randn(M, 128, generator=g) * 2 - 0.5).to(dev())
    w = (1 + 0.2 * torch.randn(128, generator=g)).to(dev())
    b = (0.2 * torch.randn(128, generator=g)).to(dev())
    st = L.stream()
    F = lambda *shape, **kw: torch.full(shape, 7.0, device=dev(), **kw)
    xa, ia, pa, aa = F(M, 128), F(B, 128), F(Mo, 128), torch.full((Mo, 128), -3, device=dev(), dtype=torch.int32)
    L.call("dd3d_inorm_arp_fwd", L.ptr(Y), L.ptr(seg), B, 128, L.ptr(xa), L.ptr(ia), L.ptr(w), L.ptr(b), L.ptr(kids), ld, 8, L.ptr(seg_o), L.ptr(pa),
           L.ptr(aa), st)
    mu, ib, pb, ab = F(B, 128), F(B, 128), F(Mo, 128), torch.full((Mo, 128), -3, device=dev(), dtype=torch.int32)
    L.call("dd3d_inorm_arp_fwd_y", L.ptr(Y), L.ptr(seg), B, 128, L.ptr(mu), L.ptr(ib), L.ptr(w), L.ptr(b), L.ptr(kids), ld, 8, L.ptr(seg_o), L.ptr(pb),
           L.ptr(ab), st)
    torch.cuda.synchronize()
    for u, v in ((ia, ib), (pa, pb), (aa, ab)):
        assert torch.equal(u, v)
    n = (seg[1:] - seg[:-1]).long()
    cloud = torch.repeat_interleave(torch.arange(B, device=dev()), n)
    ref_mu = torch.zeros(B, 128, device=dev(), dtype=torch.float64).index_add_(0, cloud, Y.double()) / n.clamp(min=1)[:, None]
    assert float((mu.double() - ref_mu).abs().max()) < 1e-5
    assert torch.equal((Y - mu[cloud]) * ib[cloud], xa)   # the identity the backward relies on
    G = 2
    gco = torch.tensor([0, B // 2, B], dtype=torch.int32, device=dev())
    cg = (torch.arange(B, device=dev()) >= B // 2).to(torch.int32)
    g_pool, ga_prev, gw_extra = (torch.randn(*sh, generator=g).to(dev()) for sh in ((Mo, 128), (M, 128), (G, 128)))
    for second in (False, True):
        outs = []
        for lean in (False, True):
            dY, ga, gw_w, gw_b = F(M, 128), F(M, 128), F(G, 128), F(G, 128)
            ws = torch.empty(int(L.lib.dd3d_arp_inorm_bwd_ws_bytes(B)), dtype=torch.uint8, device=dev())
            head = (L.ptr(g_pool), L.ptr(aa), L.ptr(parent)) + ((L.ptr(Y), L.ptr(mu)) if lean else (L.ptr(xa),))
            L.call("dd3d_arp_inorm_bwd_y" if lean else "dd3d_arp_inorm_bwd", *head, L.ptr(ia), L.ptr(w), L.ptr(b), 128, L.ptr(seg), B, L.ptr(cg), L.ptr(gco),
                   G, L.ptr(dY), L.ptr(ga), 0, L.ptr(ga_prev) if second else None, L.ptr(gw_extra) if second else None, L.ptr(gw_w), L.ptr(gw_b),
                   L.ptr(ws), ws.numel(), st)
            torch.cuda.synchronize()
            outs.append((dY, ga, gw_w, gw_b))
        for u, v in zip(*outs):
            assert torch.equal(u, v)
        assert bool(torch.isfinite(outs[1][0]).all()) and (M == B or float(outs[1][0].abs().max()) > 0)   # (single-voxel clouds: a zero gradient)
    # a missing mean is an argument error, not a launch
    assert L.lib.dd3d_arp_inorm_bwd_y(L.ptr(g_pool), L.ptr(aa), L.ptr(parent), L.ptr(Y), None, L.ptr(ia), L.ptr(w), L.ptr(b), 128, L.ptr(seg), B, None, None, 1,
                                      L.ptr(outs[0][0]), None, 0, None, None, None, None, None, 0, st) == -1


def test_sparse_linear_closed_under_differentiation():
    from dd3d_amd.ops import DensePlacement, sparse_linear
    coords = make_coords(B=6, pts=200, extent=28, seed=6)
    cm = manager(coords)
    ts = 4
    M2, B = cm.num_rows(ts), cm.B
    dp = DensePlacement(cm, ts, (1, 28, 28))
    cm.sync()
    g = torch.Generator().manual_seed(7)
    F0 = torch.randn(M2, 128, generator=g)
    W0 = torch.randn(10, 100352, generator=g) * 0.05
    b0 = torch.randn(10, generator=g)
    u = torch.randn(B, 10, generator=g, dtype=torch.float64)
    f, w, b = (v.to(dev()).requires_grad_() for v in (F0, W0, b0))
    out = sparse_linear(f, w, b, dp)
    st = ME.SparseTensor(F0.double().requires_grad_(), cm.coordinates(ts).cpu(), ts)
    fr = st.F
    wr, br = W0.double().requires_grad_(), b0.double().requires_grad_()
    minc = st.C.min(0, keepdim=True)[0][:, 1:]
    dense = st.dense(shape=torch.Size([B, 128, 1, 28, 28]), min_coordinate=minc)[0]
    ref = dense.view(B, -1) @ wr.t() + br
    assert rel_err(out, ref) < 2e-5
    # first order, kept differentiable
    L1 = (torch.tanh(out) * u.float().to(dev())).sum()
    R1 = (torch.tanh(ref) * u).sum()
    gf, gw, gb = torch.autograd.grad(L1, [f, w, b], create_graph=True)
    rf, rw, rb = torch.autograd.grad(R1, [fr, wr, br], create_graph=True)
    assert rel_err(gf, rf) < 2e-5 and rel_err(gw, rw) < 2e-5 and rel_err(gb, rb) < 2e-5
    # second order through dW (the matching-loss path) and through dF
    L2 = (gw ** 2).sum() + (gf * gf).sum() + (gb ** 2).sum()
    R2 = (rw ** 2).sum() + (rf * rf).sum() + (rb ** 2).sum()
    hf, hw = torch.autograd.grad(L2, [f, w])
    qf, qw = torch.autograd.grad(R2, [fr, wr])
    assert rel_err(hf, qf) < 5e-5 and rel_err(hw, qw) < 5e-5


@pytest.mark.parametrize("B,J,gco", [(40, 10, [0, 7, 7, 25, 40]), (1300, 10, [0, 1100, 1300]), (30, 40, [0, 30])])
def test_sparse_linear_bwd_w_layouts_groups_and_long_lists(B, J, gco):
    """dd3d_sparse_linear_bwd_w (compacted cloud lists) against a dense fp64 reference per group: the parameter layout [G, J, C*S] (zero
    fill + occupied sites) and the step's [G, J, S, C] layout (every row written by the kernel, no fill); an empty group, a group of
    more clouds than one compaction pass (1024), more outputs than one register pass (16); both layouts agree bit for bit."""
    from dd3d_amd import _lib as L
    from dd3d_amd.ops import DensePlacement, _param_strides
    import ctypes
    d = dev()
    coords = make_coords(B=B, pts=60, extent=28, seed=B)
    cm = manager(coords, B)
    G = len(gco) - 1
    gco_t = torch.tensor(gco, dtype=torch.int32, device=d)
    cg = torch.bucketize(torch.arange(B), torch.tensor(gco[1:]), right=True).clamp(max=G - 1).to(torch.int32).to(d)
    dp = DensePlacement(cm, 4, (1, 28, 28), cg, G, gco_t)
    cm.sync()
    M2, S, C = cm.num_rows(4), dp.S, 128
    g = torch.Generator().manual_seed(3)
    F0 = torch.randn(M2, C, generator=g)
    dl = torch.randn(B, J, generator=g)
    f, dd = F0.to(d), dl.to(d)
    outs = {}
    for name, strides, shape in (("param", _param_strides(J, C, S), (G, J, C, S)), ("site", (ctypes.c_int64 * 4)(J * C * S, C * S, 1, C), (G, J, S, C))):
        dW = torch.full(shape, 7.0, device=d)
        L.call("dd3d_sparse_linear_bwd_w", L.ptr(dd), L.ptr(f), L.ptr(dp.vox_at), S, B, L.ptr(gco_t), G, C, J, L.ptr(dW), strides, dW.numel(), L.stream())
        outs[name] = dW if name == "param" else dW.permute(0, 1, 3, 2).contiguous()
    assert torch.equal(outs["param"], outs["site"])
    site = dp.site[:M2].cpu().long()
    cloud = cm.coordinates(4)[:, 0].cpu().long()
    ref = torch.zeros(G, J, C, S, dtype=torch.float64)
    for gi in range(G):
        rows = torch.nonzero((cloud >= gco[gi]) & (cloud < gco[gi + 1])).reshape(-1)
        if rows.numel():
            contrib = dl.double()[cloud[rows]][:, :, None] * F0.double()[rows][:, None, :]     # [n, J, C]
            ref[gi].index_add_(2, site[rows], contrib.permute(1, 2, 0))
        else:
            assert float(outs["param"][gi].abs().max()) == 0.0
    assert rel_err(outs["param"], ref) < 2e-5


_SMALL_LOSS_SCRIPT = """
import sys, torch
sys.path.insert(0, %r)
from dd3d_amd.fused_step import grouped_match_loss
g = torch.Generator().manual_seed(11)
G = 9
shapes = [(1, 128), (4, 2000), (10,), (27, 3, 128)]
out = {}
for metric in ("ours", "modmse", "mse", "cos"):
    syn = [torch.randn((G,) + s, generator=g).cuda() for s in shapes]
    real = [torch.randn((G,) + s, generator=g).cuda() for s in shapes]
    loss, d = grouped_match_loss(syn, real, metric, G, [True, True, False, True])
    loss2, _ = grouped_match_loss(syn[:1], real[:1], metric, G, [False])
    loss3, _ = grouped_match_loss([syn[3], syn[0]], [real[3], real[0]], metric, G, [False, False])
    out[metric] = [loss.cpu(), loss2.cpu(), loss3.cpu(), d[0].cpu(), d[1].cpu(), d[3].cpu()]
torch.save(out, sys.argv[1])
"""


def test_one_launch_match_loss_is_bitwise_the_three_launches(tmp_path):
    """dd3d_match_loss_fwd_bwd / dd3d_match_loss_fwd on a SMALL list (an instance-norm weight's gradient, [1,128] per class, the first
    layer's kernel gradient [27,3,128], whose rows are its output channels; here six row-major rows of at most 2000 columns and that kernel
    gradient, the one-row list alone, and the pair the step's tail uses) run as one launch; the same seeded inputs in a process with DD3D_GD_SMALL=0 run as
    the three launches.  Every loss and every gradient must have the same bits, for the four metrics."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for flag in ("1", "0"):
        f = str(tmp_path / f"loss_{flag}.pt")
        env = dict(os.environ, DD3D_GD_SMALL=flag)
        subprocess.run([sys.executable, "-c", _SMALL_LOSS_SCRIPT % root, f], check=True, env=env, timeout=240)
        outs.append(torch.load(f))
    for metric in ("ours", "modmse", "mse", "cos"):
        for a, b in zip(outs[0][metric], outs[1][metric]):
            assert torch.equal(a, b), metric
        assert bool(torch.isfinite(outs[0][metric][0]).all())


def test_loss_total_is_the_two_additions_and_the_sum():
    """dd3d_loss_total: per_class = ((a + b) + c) + d with torch's own bits, total = their sum (fp32 rounding of another order), nullable terms and
    outputs, more groups than lanes."""
    from dd3d_amd import _lib as L
    g = torch.Generator().manual_seed(2)
    for G in (1, 10, 40, 200):
        a, b, c = (torch.rand(G, generator=g).to(dev()) * 1e3 for _ in range(3))
        per, tot = torch.empty(G, device=dev()), torch.empty((), device=dev())
        d = torch.rand(G, generator=g).to(dev()) * 1e3
        L.call("dd3d_loss_total", L.ptr(a), L.ptr(b), L.ptr(c), L.ptr(d), G, L.ptr(per), L.ptr(tot), L.stream())
        assert torch.equal(per, ((a + b) + c) + d)
        assert tot.item() == pytest.approx(per.double().sum().item(), rel=1e-6)
        L.call("dd3d_loss_total", L.ptr(a), None, L.ptr(c), None, G, None, L.ptr(tot), L.stream())
        assert tot.item() == pytest.approx((a + c).double().sum().item(), rel=1e-6)
        L.call("dd3d_loss_total", L.ptr(a), None, None, None, G, L.ptr(per), None, L.stream())
        assert torch.equal(per, a)


def test_true_divide_is_the_host_division_on_the_device():
    """coordinates / quantization_size: the reference divides host tensors (a correctly rounded fp32 division, models/MEConv.py:262).  On the
    device ``tensor / python_scalar`` multiplies by the rounded reciprocal -- different bits for a voxel size like 0.05, different VOXELS for
    points at a face -- so the package divides through _lib.true_divide, which must give the host's bits; so must dd3d_gather_clouds."""
    from dd3d_amd import _lib as L
    g = torch.Generator().manual_seed(0)
    x = (torch.rand(2_000_000, 3, generator=g) * 2 - 1).float()
    faces = (torch.arange(-20, 21).float() * 0.05)[:, None].expand(-1, 3)            # points ON voxel faces and one ulp either side
    x = torch.cat([x, faces, torch.nextafter(faces, torch.full_like(faces, 9.0)), torch.nextafter(faces, torch.full_like(faces, -9.0))]).contiguous()
    for q in (0.05, 0.02, 1.0, 0.3):
        want = x / q                                                                 # host division
        assert torch.equal(L.true_divide(x.to(dev()), q).cpu(), want)
        n = x.shape[0]
        off = torch.tensor([0, n], dtype=torch.int32, device=dev())
        ids = torch.zeros(1, dtype=torch.int32, device=dev())
        coords, feats = torch.empty((n, 4), device=dev()), torch.empty((n, 3), device=dev())
        pt_off, cnt, err = torch.zeros(2, dtype=torch.int32, device=dev()), torch.zeros(1, dtype=torch.int32, device=dev()), torch.zeros(1, dtype=torch.int32, device=dev())
        L.call("dd3d_gather_clouds", L.ptr(x.to(dev())), L.ptr(off), 1, L.ptr(ids), 1, n, n, 0, 0, q, L.ptr(coords), L.ptr(feats), L.ptr(pt_off), L.ptr(cnt),
               L.ptr(err), L.stream())
        assert int(err.item()) == 0 and torch.equal(coords[:, 1:].cpu(), want) and torch.equal(feats.cpu(), x)
    naive = (x.to(dev()) / 0.05).cpu()
    print("tensor / 0.05 on the device differs from the host division in", int((naive != x / 0.05).sum()), "of", x.numel(), "elements;",
          int((torch.floor(naive) != torch.floor(x / 0.05)).sum()), "land in another voxel")


def test_dense_out_of_extent_is_an_error():
    from dd3d_amd.ops import DensePlacement
    coords = torch.tensor([[0, 0, 0.0, 0.0], [0, 0, 0.0, 200.0]])
    cm = manager(coords, 1)
    DensePlacement(cm, 4, (1, 28, 28))
    with pytest.raises(IndexError):
        cm.sync()


@pytest.mark.parametrize("case", CASES + [dict(B=12, pts=1000, extent=20, seed=6, dims3=True, neg=True)])
def test_cloud_local_maps_equal_generic_maps(case):
    """dd3d_cloud_maps (one workgroup per cloud, LDS hashes, 3 launches) against the generic global-hash chain of the same library
    (itself pinned to the oracle above): every integer output bit-identical -- coordinates, counts, inverse / first-row maps, the
    voxel->points CSR, both k=3 kernel maps, both pooling maps, per-cloud row ranges and the dense() placement with 2 groups."""
    from dd3d_amd.ops import DensePlacement
    from dd3d_amd.sparse import CloudCoordinateManager
    coords = make_coords(**case).to(dev())
    B = case["B"]
    # an empty cloud in the middle (and one at the end) must not disturb anything
    coords[:, 0] += (coords[:, 0] >= B // 2).float()
    B += 2
    cnt = torch.bincount(coords[:, 0].long(), minlength=B)
    pt_off = torch.cat([cnt.new_zeros(1), torch.cumsum(cnt, 0)]).to(torch.int32)
    maxp = int(cnt.max())
    G = 2
    cg = (torch.arange(B, device=dev()) >= B // 3).to(torch.int32)
    gco = torch.tensor([0, B // 3, B], dtype=torch.int32, device=dev())
    ext = (64, 64, 64) if case.get("dims3") else (1, 64, 64)
    ref = manager(coords, B)
    dref = DensePlacement(ref, 4, ext, cg, G, gco)
    # bound larger than the point count: trailing rows of `coords` are ignored
    padded = torch.cat([coords, torch.full((37, 4), 3.0, device=dev())])
    cm = CloudCoordinateManager(padded, pt_off, B, maxp, ext, cg, G)
    dcm = DensePlacement(cm, 4, ext, cg, G, gco)
    assert dcm.site is cm.dense_tables[3]
    ref.sync()
    cm.sync()
    N = coords.shape[0]
    Ms = [ref.num_rows(s) for s in (1, 2, 4)]
    assert [cm.num_rows(s) for s in (1, 2, 4)] == Ms
    for s, M in zip((1, 2, 4), Ms):
        assert torch.equal(cm.coordinates(s), ref.coordinates(s))
        assert torch.equal(cm.seg_off(s), ref.seg_off(s))
        assert torch.equal(cm.levels[s].uidx[:M], ref.levels[s].uidx[:M])
    assert torch.equal(cm.levels[1].inv[:N], ref.levels[1].inv[:N])
    assert torch.equal(cm.parent(2)[:Ms[0]], ref.parent(2)[:Ms[0]]) and torch.equal(cm.parent(4)[:Ms[1]], ref.parent(4)[:Ms[1]])
    assert torch.equal(cm.vox_off[:Ms[0] + 1], ref.vox_off[:Ms[0] + 1]) and torch.equal(cm.vox_pts[:N], ref.vox_pts[:N])
    for (si, so, k) in ((1, 1, 3), (2, 2, 3), (1, 2, 2), (2, 4, 2)):
        Mo = ref.num_rows(so)
        assert torch.equal(cm.kernel_map(si, so, k)[0][:, :Mo], ref.kernel_map(si, so, k)[0][:, :Mo]), (si, so, k)
    assert torch.equal(dcm.gmin, dref.gmin)
    assert torch.equal(dcm.site[:Ms[2]], dref.site[:Ms[2]]) and torch.equal(dcm.vox_at, dref.vox_at)
    # dd3d_cloud_maps_f0: the same maps plus the per-voxel feature mean of dd3d_voxel_mean_fwd, bit for bit (with group_cloud_off: the scan
    # inside pass B)
    from dd3d_amd import _lib as L
    feats = torch.randn(padded.shape[0], 3, generator=torch.Generator().manual_seed(5)).to(dev())
    cf = CloudCoordinateManager(padded, pt_off, B, maxp, ext, cg, G, gco, feats3=feats)
    cf.sync()
    want = torch.full((Ms[0], 3), 7.0, device=dev())
    L.call("dd3d_voxel_mean_fwd", L.ptr(feats), N, 3, L.ptr(ref.vox_off), L.ptr(ref.vox_pts), Ms[0], None, L.ptr(want), L.stream())
    assert torch.equal(cf.F0[:Ms[0]], want)
    assert torch.equal(cf.kernel_map(2, 2, 3)[0][:, :Ms[1]], ref.kernel_map(2, 2, 3)[0][:, :Ms[1]]) and torch.equal(cf.seg_off(4), ref.seg_off(4))
    assert torch.equal(cf.group_row_off[:, 1], torch.stack([ref.seg_off(s)[B // 3] for s in (1, 2, 4)]))


def test_cloud_local_maps_flag_errors():
    from dd3d_amd.sparse import CloudCoordinateManager
    coords = make_coords(B=3, pts=50, extent=10, seed=8).to(dev())
    cnt = torch.bincount(coords[:, 0].long(), minlength=3)
    pt_off = torch.cat([cnt.new_zeros(1), torch.cumsum(cnt, 0)]).to(torch.int32)
    bad = coords.clone()
    bad[0, 0] = 2.0                                     # a point whose batch column is not its cloud
    with pytest.raises(RuntimeError):
        CloudCoordinateManager(bad, pt_off, 3, 64).sync()
    with pytest.raises(IndexError):                     # dense extent too small
        CloudCoordinateManager(coords, pt_off, 3, 64, (1, 1, 1)).sync()
    with pytest.raises(ValueError):
        CloudCoordinateManager(coords, pt_off, 3, 100000)


def _random_table(M, K, p, seed):
    """A neighbour table with random partners: the convolution kernels do not care where the rows come from."""
    g = torch.Generator().manual_seed(seed)
    nbr = torch.randint(0, M, (K, M), generator=g, dtype=torch.int32)
    # a few patterns, repeated: rows share neighbour patterns as voxels of strokes do
    pat = (torch.rand((K, 64), generator=g) < p)
    nbr[~pat[:, torch.randint(0, 64, (M,), generator=g)]] = -1
    return nbr


@pytest.mark.parametrize("M,bound", [(70001, 70001), (66000, 80000), (3000, 3000), (150000, 150000), (300000, 300000), (20000, 30000), (40000, 40000), (1, 5),
                                     (1100000, 1100000), (1050000, 1200000), (66000, 300000), (66000, 1100000)])
def test_scheduled_conv128_is_bit_identical(M, bound):
    """dd3d_conv_schedule + dd3d_conv128_ordered against dd3d_conv_fwd (itself checked against the oracle above): same bits for the
    forward and the mirrored launch, with an exact count and with a bound + device-side count; the schedule is a permutation of the rows
    and covers every tile once.  Sizes on both sides of the 2048-tile limit of the placement and of the row counts from which the
    second sort level runs."""
    from dd3d_amd import _lib as L
    d = dev()
    nbr = _random_table(bound, 27, 0.4, M).to(d)
    nbr[nbr >= M] = -1
    X = torch.randn(bound, 128, generator=torch.Generator().manual_seed(1)).to(d)
    W = (torch.randn(27, 128, 128, generator=torch.Generator().manual_seed(2)) * 0.05).to(d)
    m_dev = torch.tensor([M], dtype=torch.int32, device=d) if bound != M else None
    nblk = int(L.lib.dd3d_conv_schedule_blocks(bound))
    perm = torch.full((bound,), -7, dtype=torch.int32, device=d)
    tmap = torch.full((nblk,), -7, dtype=torch.int32, device=d)
    ws = torch.empty(int(L.lib.dd3d_conv_schedule_ws_bytes(bound)), dtype=torch.uint8, device=d)
    tr = int(L.lib.dd3d_conv_schedule_tile_rows(bound))   # 128-row tiles from 65536 rows on, 32-row tiles below
    # optional outputs: the table in pattern order and the offsets of every tile
    nbr_s = torch.full((27, bound), -9, dtype=torch.int32, device=d)
    tmask = torch.zeros(((bound + tr - 1) // tr,), dtype=torch.int32, device=d)
    L.call("dd3d_conv_schedule", L.ptr(nbr), bound, 27, bound, L.ptr(m_dev), L.ptr(perm), L.ptr(tmap), L.ptr(nbr_s), L.ptr(tmask), L.ptr(ws),
           ws.numel(), L.stream())
    assert sorted(perm[:M].tolist()) == list(range(M))
    tiles = [t for t in tmap.tolist() if t >= 0]
    assert sorted(tiles) == list(range((M + tr - 1) // tr)) and min(tmap.tolist()) >= -1
    nt = (M + tr - 1) // tr
    pres = (nbr[:, perm[:M].long()] >= 0)
    pres = torch.cat([pres, pres.new_zeros(27, nt * tr - M)], 1).reshape(27, nt, tr).any(2)
    want_mask = (pres.long() << torch.arange(27, device=d)[:, None]).sum(0)
    assert torch.equal(tmask[:nt].long() & 0x7ffffff, want_mask)
    if not bool((tmask[:nt] < 0).any()):   # bit 31: a second sort level ran and the table was left alone
        assert torch.equal(nbr_s[:, :M], nbr[:, perm[:M].long()])
    else:
        assert bool((tmask[:nt] < 0).all())
    # a schedule without the optional outputs is the same schedule up to the order inside a bucket: same tiles, same result
    perm_b, tmap_b = torch.empty_like(perm), torch.empty_like(tmap)
    L.call("dd3d_conv_schedule", L.ptr(nbr), bound, 27, bound, L.ptr(m_dev), L.ptr(perm_b), L.ptr(tmap_b), None, None, L.ptr(ws), ws.numel(), L.stream())
    assert sorted(perm_b[:M].tolist()) == list(range(M))
    for mirror in (0, 1):
        Y = torch.full((bound, 128), 7.0, device=d)
        Y2 = torch.full((bound, 128), 7.0, device=d)
        Y3 = torch.full((bound, 128), 7.0, device=d)
        L.call("dd3d_conv_fwd", L.ptr(X), bound, 128, L.ptr(W), 128, L.ptr(nbr), bound, 27, bound, L.ptr(m_dev), mirror, L.ptr(Y), L.stream())
        L.call("dd3d_conv128_ordered", L.ptr(X), bound, L.ptr(W), L.ptr(nbr), bound, 27, bound, L.ptr(m_dev), mirror, L.ptr(perm), L.ptr(tmap),
               L.ptr(nbr_s), L.ptr(tmask), L.ptr(Y2), L.stream())
        L.call("dd3d_conv128_ordered", L.ptr(X), bound, L.ptr(W), L.ptr(nbr), bound, 27, bound, L.ptr(m_dev), mirror, L.ptr(perm_b), L.ptr(tmap_b),
               None, None, L.ptr(Y3), L.stream())
        assert torch.equal(Y, Y2) and torch.equal(Y, Y3)
        assert bool((Y[M:] == 7.0).all())   # rows beyond the device-side count are not touched
    # the patterns are grouped: fewer (32-row block, offset) products than in row order
    present = (nbr[:, :M] >= 0)
    def blocks(order):
        pm = present[:, order.long()]
        pm = torch.cat([pm, pm.new_zeros(27, (-M) % 32)], 1)
        return int(pm.reshape(27, -1, 32).any(2).sum())
    assert M < 64 or blocks(perm[:M]) < blocks(torch.arange(M, device=d))


@pytest.mark.parametrize("M,bound,mirror", [(1, 1, 0), (31, 40, 1), (1600, 1600, 0), (3500, 20000, 1), (5000, 5000, 1), (9000, 9000, 0), (20000, 20000, 1), (80000, 80000, 0)])
def test_offsets_split_conv128_matches_plain_and_oracle(M, bound, mirror):
    """dd3d_conv128_split (every split count from 9 down to 1) against dd3d_conv_fwd (fp32 rounding apart) and, for the small cases, the fp64 oracle;
    rows beyond the device-side count stay untouched; two calls give the same bits."""
    from dd3d_amd import _lib as L
    g = torch.Generator().manual_seed(M + mirror)
    nbr = torch.randint(0, M, (27, bound), generator=g, dtype=torch.int32)
    nbr[torch.rand(27, bound, generator=g) > 0.3] = -1
    nbr[13, :M] = torch.arange(M, dtype=torch.int32)
    X = torch.randn(M, 128, generator=g)
    W = torch.randn(27, 128, 128, generator=g) * 0.05
    nbr_d, X_d, W_d = nbr.to(dev()), X.to(dev()), W.to(dev())
    m_dev = torch.tensor([M], dtype=torch.int32, device=dev()) if bound != M else None
    st = L.stream()
    Y0 = torch.full((bound, 128), 7.0, device=dev())
    L.call("dd3d_conv_fwd", L.ptr(X_d), M, 128, L.ptr(W_d), 128, L.ptr(nbr_d), bound, 27, bound, L.ptr(m_dev) if m_dev is not None else None, mirror, L.ptr(Y0), st)
    ws = torch.empty(int(L.lib.dd3d_conv128_split_ws_bytes(bound)), dtype=torch.uint8, device=dev())
    outs = []
    for _ in range(2):
        ws.fill_(0xFF)  # NaN bit patterns: every partial that is read must have been written by this call
        Y = torch.full((bound, 128), 7.0, device=dev())
        L.call("dd3d_conv128_split", L.ptr(X_d), M, L.ptr(W_d), L.ptr(nbr_d), bound, 27, bound, L.ptr(m_dev) if m_dev is not None else None, mirror, L.ptr(Y),
               L.ptr(ws), ws.numel(), st)
        outs.append(Y)
    torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1])
    assert torch.equal(outs[0][M:], Y0[M:])
    assert rel_err(outs[0][:M], Y0[:M]) < 2e-6
    if M <= 9000:
        ref = torch.zeros(M, 128, dtype=torch.float64)
        Xd, Wd = X.double(), W.double()
        for k in range(27):
            idx = nbr[26 - k if mirror else k, :M].long()
            has = idx >= 0
            ref[has] += Xd[idx[has]] @ Wd[k]
        assert rel_err(outs[0][:M], ref) < 2e-5
    assert L.lib.dd3d_conv128_split(L.ptr(X_d), M, L.ptr(W_d), L.ptr(nbr_d), bound, 27, bound, None, 0, L.ptr(outs[0]), L.ptr(ws), 16, st) == -3


def _fp64_conv(X, W, nbr, M, mirror):
    """Y[o] = sum_k X[nbr[k'][o]] . W[k], k' = 26 - k for the mirrored launch -- per-offset fp64 matmuls on the CPU."""
    ref = torch.zeros(M, 128, dtype=torch.float64)
    Xd, Wd = X.double(), W.double()
    for k in range(27):
        idx = nbr[26 - k if mirror else k, :M].long()
        has = idx >= 0
        ref[has] += Xd[idx[has]] @ Wd[k]
    return ref


@pytest.mark.parametrize("M,bound", [(70001, 70001), (90000, 131072)])
@pytest.mark.parametrize("mirror", [0, 1])
def test_large_conv128_rows_kernel_vs_fp64_oracle(M, bound, mirror):
    """The launches the bench runs: from 65536 rows on dd3d_conv_fwd and dd3d_conv128_ordered select k_conv128_rows<4> (128-row tiles,
    row-private waves).  Both, forward and mirrored (the backward-data launch), against per-offset fp64 matmuls; rel 2e-5 of the
    tensor's max magnitude, with an exact count and with a bound + device-side count."""
    from dd3d_amd import _lib as L
    d = dev()
    assert int(L.lib.dd3d_conv_schedule_tile_rows(bound)) == 128
    nbr = _random_table(bound, 27, 0.4, M + mirror)
    nbr[nbr >= M] = -1
    g = torch.Generator().manual_seed(M)
    X = torch.randn(bound, 128, generator=g)
    W = torch.randn(27, 128, 128, generator=g) * 0.05
    ref = _fp64_conv(X, W, nbr, M, mirror)
    nbr_d, X_d, W_d = nbr.to(d), X.to(d), W.to(d)
    m_dev = torch.tensor([M], dtype=torch.int32, device=d) if bound != M else None
    Y = torch.full((bound, 128), 7.0, device=d)
    L.call("dd3d_conv_fwd", L.ptr(X_d), bound, 128, L.ptr(W_d), 128, L.ptr(nbr_d), bound, 27, bound, L.ptr(m_dev), mirror, L.ptr(Y), L.stream())
    e1 = rel_err(Y[:M], ref)
    perm = torch.empty((bound,), dtype=torch.int32, device=d)
    tmap = torch.empty((int(L.lib.dd3d_conv_schedule_blocks(bound)),), dtype=torch.int32, device=d)
    ws = torch.empty(int(L.lib.dd3d_conv_schedule_ws_bytes(bound)), dtype=torch.uint8, device=d)
    nbr_s = torch.empty_like(nbr_d)
    tmask = torch.empty(((bound + 127) // 128,), dtype=torch.int32, device=d)
    L.call("dd3d_conv_schedule", L.ptr(nbr_d), bound, 27, bound, L.ptr(m_dev), L.ptr(perm), L.ptr(tmap), L.ptr(nbr_s), L.ptr(tmask), L.ptr(ws),
           ws.numel(), L.stream())
    Y2 = torch.full((bound, 128), 7.0, device=d)
    L.call("dd3d_conv128_ordered", L.ptr(X_d), bound, L.ptr(W_d), L.ptr(nbr_d), bound, 27, bound, L.ptr(m_dev), mirror, L.ptr(perm), L.ptr(tmap),
           L.ptr(nbr_s), L.ptr(tmask), L.ptr(Y2), L.stream())
    e2 = rel_err(Y2[:M], ref)
    print(f"conv128_rows M={M} mirror={mirror}: plain {e1:.2e} ordered {e2:.2e}")
    assert e1 < 2e-5 and e2 < 2e-5
    assert bool((Y[M:] == 7.0).all()) and bool((Y2[M:] == 7.0).all())


@pytest.mark.parametrize("M,bound", [(70001, 70001), (90000, 131072)])
def test_large_conv128_bwd_weight_vs_fp64_oracle(M, bound):
    """dd3d_conv_bwd_weight, Cin = 128, at the row counts of the bench (>= 65536 rows: 512-row chunks), two groups of rows, against fp64
    X[nbr]^T dY per offset and group; bitwise reproducible; a bound-sized launch gives the bits of the exact one."""
    from dd3d_amd import _lib as L
    d = dev()
    nbr = _random_table(bound, 27, 0.4, M + 5)
    nbr[nbr >= M] = -1
    g = torch.Generator().manual_seed(M + 1)
    X = torch.randn(bound, 128, generator=g)
    dY = torch.randn(bound, 128, generator=g)
    cut = M // 3
    gro = torch.tensor([0, cut, M], dtype=torch.int32, device=d)
    nbr_d, X_d, dY_d = nbr.to(d), X.to(d), dY.to(d)

    def run(bnd):
        dW = torch.full((2, 27, 128, 128), 7.0, device=d)
        ws = torch.empty(int(L.lib.dd3d_conv_bwd_weight_ws_bytes(bnd, 2, 27, 128, 128)), dtype=torch.uint8, device=d)
        ws.fill_(0xFF)
        L.call("dd3d_conv_bwd_weight", L.ptr(X_d), 128, L.ptr(dY_d), 128, L.ptr(nbr_d), bound, 27, bnd, L.ptr(gro), 2, L.ptr(dW), L.ptr(ws), ws.numel(),
               L.stream())
        return dW
    dW = run(bound)
    assert torch.equal(dW, run(bound))
    if bound != M:
        assert torch.equal(dW, run(M))
    Xd, Dd = X.double(), dY.double()
    for gi, (a, b) in enumerate(((0, cut), (cut, M))):
        ref = torch.zeros(27, 128, 128, dtype=torch.float64)
        for k in range(27):
            o = torch.nonzero(nbr[k, a:b] >= 0).reshape(-1) + a
            if o.numel():
                ref[k] = Xd[nbr[k][o].long()].t() @ Dd[o]
        e = rel_err(dW[gi], ref)
        print(f"conv128 bwd-weight M={M} group {gi}: {e:.2e}")
        assert e < 2e-5


def _cloud_batch_maps(B, pts, seed, groups):
    """Stride-2 kernel map of a batch of flat clouds + per-cloud / per-group row ranges (what the fused step hands the kernel gradient)."""
    coords = make_coords(B=B, pts=pts, extent=28, seed=seed)
    cm = manager(coords, B)
    nbr, ld = cm.kernel_map(2, 2, 3)
    seg = cm.seg_off(2)
    M = cm.num_rows(2)
    gco = torch.tensor(groups, device=dev())
    return cm, nbr, ld, seg, M, seg[gco.long()].contiguous().to(torch.int32)


@pytest.mark.parametrize("B,pts,groups", [(40, 300, [0, 7, 7, 25, 40]), (700, 250, [0, 100, 350, 700]), (3, 40, [0, 3]), (2600, 120, [0, 100, 200, 1400, 2600])])
def test_cloud_window_bwd_weight_vs_fp64_and_row_chunk_kernel(B, pts, groups):
    """dd3d_conv128_bwd_weight_clouds (LDS windows of whole clouds) against fp64 X[nbr]^T dY per offset and group and against the row-chunk
    kernel dd3d_conv_bwd_weight; bitwise reproducible; a bound-sized launch (rows beyond the count garbage) gives the same bits."""
    from dd3d_amd import _lib as L
    from dd3d_amd.ops import conv_bwd_weight
    d = dev()
    cm, nbr, ld, seg, M, gro = _cloud_batch_maps(B, pts, 31 + B, groups)
    G = len(groups) - 1
    g = torch.Generator().manual_seed(B)
    X = torch.randn(M, 128, generator=g)
    dY = torch.randn(M, 128, generator=g)
    err = torch.zeros(1, dtype=torch.int32, device=d)

    def run(bound, Xd, Dd):
        dW = torch.full((G, 27, 128, 128), 7.0, device=d)
        ws = torch.empty(int(L.lib.dd3d_conv128_bwd_weight_clouds_ws_bytes(bound, G)), dtype=torch.uint8, device=d)
        ws.fill_(0xFF)
        L.call("dd3d_conv128_bwd_weight_clouds", L.ptr(Xd), L.ptr(Dd), L.ptr(nbr), ld, 27, bound, L.ptr(seg), B, L.ptr(gro), G, L.ptr(dW), L.ptr(err),
               L.ptr(ws), ws.numel(), L.stream())
        return dW
    Xd, Dd = X.to(d), dY.to(d)
    dW = run(M, Xd, Dd)
    assert int(err.item()) == 0
    assert torch.equal(dW, run(M, Xd, Dd))
    pad = lambda t: torch.cat([t, torch.full((ld - M, 128), float("nan"), device=d)]) if ld > M else t
    assert torch.equal(dW, run(ld, pad(Xd), pad(Dd)))
    old = conv_bwd_weight(Xd, Dd, nbr, ld, M, gro, G)
    ref_nbr = nbr[:, :M].cpu().long()
    rows = gro.cpu().tolist()
    X64, D64 = X.double(), dY.double()
    for gi in range(G):
        a, b = rows[gi], rows[gi + 1]
        ref = torch.zeros(27, 128, 128, dtype=torch.float64)
        for k in range(27):
            o = torch.nonzero(ref_nbr[k, a:b] >= 0).reshape(-1) + a
            if o.numel():
                ref[k] = X64[ref_nbr[k][o]].t() @ D64[o]
        if a == b:
            assert float(dW[gi].abs().max()) == 0.0
        else:
            assert rel_err(dW[gi], ref) < 2e-5 and rel_err(dW[gi], old[gi]) < 2e-5


def test_cloud_window_bwd_weight_flags_a_cloud_larger_than_the_window():
    from dd3d_amd import _lib as L
    d = dev()
    cm, nbr, ld, seg, M, gro = _cloud_batch_maps(2, 2500, 5, [0, 2])   # ~190 stride-2 voxels per cloud of a 28 x 28 grid at most: force one cloud
    one = torch.tensor([0, M, M], dtype=torch.int32, device=d)         # pretend all rows are ONE cloud: more than 256 rows
    if M <= int(L.lib.dd3d_conv128_bwd_weight_clouds_max_rows()):
        pytest.skip("batch too small to overflow the window")
    err = torch.zeros(1, dtype=torch.int32, device=d)
    X = torch.randn(M, 128, device=d)
    dW = torch.empty((1, 27, 128, 128), device=d)
    ws = torch.empty(int(L.lib.dd3d_conv128_bwd_weight_clouds_ws_bytes(M, 1)), dtype=torch.uint8, device=d)
    L.call("dd3d_conv128_bwd_weight_clouds", L.ptr(X), L.ptr(X), L.ptr(nbr), ld, 27, M, L.ptr(one), 2, L.ptr(gro), 1, L.ptr(dW), L.ptr(err), L.ptr(ws),
           ws.numel(), L.stream())
    assert int(err.item()) & 16
    assert L.lib.dd3d_conv128_bwd_weight_clouds(L.ptr(X), L.ptr(X), L.ptr(nbr), ld, 27, M, L.ptr(one), 2, L.ptr(gro), 1, L.ptr(dW), L.ptr(err), L.ptr(ws), 16,
                                                L.stream()) == -3


@pytest.mark.parametrize("B,G,J", [(90, 3, 10), (300, 2, 7), (5, 5, 10)])
def test_classifier_head_is_bitwise_the_separate_entry_points(B, G, J):
    """dd3d_classifier_head (sparse linear forward + softmax CE + logit gradient + per-group CE and bias gradient, two launches) against
    dd3d_sparse_linear_fwd, dd3d_softmax_ce and dd3d_group_sum_rows: the same bits, with an empty group, groups of more than 64 clouds
    (several workgroups per group, one of them elected to sum), twice in a row on the same counters (they must be left at zero)."""
    import ctypes
    from dd3d_amd import _lib as L
    from dd3d_amd.ops import DensePlacement, _site_strides, site_major
    d = dev()
    coords = make_coords(B=B, pts=120, extent=28, seed=B)
    cm = manager(coords, B)
    sizes = np.diff(np.linspace(0, B, G, dtype=int)).tolist() + [0]            # G - 1 groups and an empty one at the end
    gco = torch.tensor(np.concatenate([[0], np.cumsum(sizes)]), dtype=torch.int32, device=d)
    cg = torch.repeat_interleave(torch.arange(G, device=d), torch.tensor(sizes, device=d)).to(torch.int32)
    dp = DensePlacement(cm, 4, (1, 28, 28), cg, G, gco)
    cm.sync()
    M2, S = cm.num_rows(4), dp.S
    g = torch.Generator().manual_seed(3)
    F = torch.randn(M2, 128, generator=g).to(d)
    Wsm = site_major((torch.randn(J, 128 * S, generator=g) * 0.05).to(d), 128, S)
    sst = _site_strides(J, 128, S)
    bias = torch.randn(J, generator=g).to(d)
    labels = torch.randint(0, J, (B,), generator=g).to(d, torch.int32)
    ws = torch.empty(int(L.lib.dd3d_sparse_linear_ws_bytes(dp.M_bound, J)), dtype=torch.uint8, device=d)
    # the separate entry points
    logits, prob, dl, closs = [torch.empty(B, J, device=d) for _ in range(3)] + [torch.empty(B, device=d)]
    ce, gcb = torch.full((G,), 7.0, device=d), torch.full((G, J), 7.0, device=d)
    L.call("dd3d_sparse_linear_fwd", L.ptr(F), L.ptr(dp.site), L.ptr(dp.C), dp.M_bound, L.ptr(dp.seg_off), B, 128, None, L.ptr(Wsm), sst, J, L.ptr(bias),
           L.ptr(logits), L.ptr(ws), ws.numel(), L.stream())
    L.call("dd3d_softmax_ce", L.ptr(logits), L.ptr(labels), L.ptr(cg), L.ptr(gco), G, B, J, L.ptr(prob), L.ptr(dl), L.ptr(closs), L.ptr(ce), L.stream())
    L.call("dd3d_group_sum_rows", L.ptr(dl), L.ptr(gco), G, J, L.ptr(gcb), L.stream())
    counters = torch.zeros(G, dtype=torch.int32, device=d)
    for _ in range(2):
        l2, p2, d2, c2 = [torch.full((B, J), 3.0, device=d) for _ in range(3)] + [torch.full((B,), 3.0, device=d)]
        ce2, gcb2 = torch.full((G,), 3.0, device=d), torch.full((G, J), 3.0, device=d)
        L.call("dd3d_classifier_head", L.ptr(F), L.ptr(dp.site), L.ptr(dp.C), dp.M_bound, L.ptr(dp.seg_off), B, 128, None, L.ptr(Wsm), sst, J,
               L.ptr(bias), L.ptr(labels), L.ptr(gco), G, max(sizes), L.ptr(l2), L.ptr(p2), L.ptr(d2), L.ptr(c2), L.ptr(ce2), L.ptr(gcb2),
               L.ptr(counters), L.ptr(ws), ws.numel(), L.stream())
        assert torch.equal(l2, logits) and torch.equal(p2, prob) and torch.equal(d2, dl) and torch.equal(c2, closs)
        assert torch.equal(ce2, ce) and torch.equal(gcb2, gcb)
        assert int(counters.abs().sum()) == 0
    assert float(ce[-1]) == 0.0 and float(gcb[-1].abs().sum()) == 0.0      # the empty group


def test_cloud_maps_minimum_corner_floor_places_a_part_like_the_whole_batch():
    """dd3d_cloud_maps(gmin_floor): the even clouds of a batch, given the minimum stride-4 corner of the whole batch's groups, get the dense
    sites they have in the whole batch (what a rank holding part of a class's real batch needs, MEConv.py:251-254); without the floor
    at least one group is placed by a different corner."""
    from dd3d_amd.ops import DensePlacement
    from dd3d_amd.sparse import CloudCoordinateManager
    d = dev()
    B, ext, G = 12, (1, 64, 64), 2
    coords = make_coords(B=B, pts=80, extent=24, seed=9)
    coords[:, 2:] += 8.0
    coords[coords[:, 0] == 1, 2:] *= 0.2           # cloud 1 (odd: not in the part) reaches into the low corner of group 0
    coords[coords[:, 0] == 9, 2:] *= 0.2           # cloud 9 likewise for group 1

    def maps(c, nb, floor=None):
        c = c.to(d)
        cnt = torch.bincount(c[:, 0].long(), minlength=nb)
        pt = torch.cat([cnt.new_zeros(1), torch.cumsum(cnt, 0)]).to(torch.int32)
        cg = (torch.arange(nb, device=d) >= nb // 2).to(torch.int32)
        gco = torch.tensor([0, nb // 2, nb], dtype=torch.int32, device=d)
        cm = CloudCoordinateManager(c, pt, nb, int(cnt.max()), ext, cg, G, gco, gmin_floor=floor)
        dp = DensePlacement(cm, 4, ext, cg, G, gco)
        cm.sync()
        return cm, dp

    whole, dw = maps(coords, B)
    keep = (coords[:, 0].long() % 2) == 0
    part = coords[keep].clone()
    part[:, 0] = torch.div(part[:, 0], 2, rounding_mode="floor")     # clouds 0, 2, 4, ... renumbered 0, 1, 2, ...
    (pm, dpm), (nm, dnm) = maps(part, B // 2, dw.gmin.clone()), maps(part, B // 2)
    assert torch.equal(dpm.gmin, dw.gmin) and not torch.equal(dnm.gmin, dw.gmin)
    # the stride-4 voxels of cloud 2b in the whole batch = those of cloud b in the part, in the same order
    M4 = pm.num_rows(4)
    segw, segp = whole.seg_off(4).tolist(), pm.seg_off(4).tolist()
    want = torch.cat([dw.site[segw[2 * b]:segw[2 * b + 1]] for b in range(B // 2)])
    assert torch.equal(dpm.site[:M4], want) and not torch.equal(dnm.site[:M4], want)
    pm.check(), whole.check()
