"""One full-size cfg2 step on the GPU, taken apart stage by stage against the op-level entry points of the oracle's C++/OpenMP leg
(oracle/cpu/dd3d_cpu.h): the merged batch's coordinate maps bit for bit, then every forward stage from the DEVICE's own input of that stage
(so an error cannot hide behind, or be blamed on, the stage before it), then the first-order parameter gradients of one class."""
import numpy as np
import pytest
import torch

from oracle import cpu_leg as CL
from dd3d_amd.synthetic import init_synthetic, mnist3d_like
from tests.detgen import det_state_dict

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def rel(a, b):
    a = a.detach().cpu().double().numpy() if torch.is_tensor(a) else np.asarray(a, np.float64)
    b = b.detach().cpu().double().numpy() if torch.is_tensor(b) else np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


def test_full_cfg2_batch_stage_by_stage_vs_cpp_leg():
    from dd3d_amd.fused_step import FusedDistiller
    from dd3d_amd.models.MEConv import MEConvExp
    torch.manual_seed(0)
    net = MEConvExp(in_channel=3, out_channel=10)
    net.load_state_dict(det_state_dict(net, "net.MEConvExp"))
    images, _, idx = mnist3d_like(10, 256, seed=0)
    with FusedDistiller(images, idx, net.to(DEV).train(), ipc=10, device=DEV, seed=7, image_syn=init_synthetic(10, 10, 250, seed=0), graph=True) as f:
        f.outer_step()
        f.outer_step()                                   # capture + replay: the buffers below are the replayed graph's
        assert f._graph is not None
        torch.cuda.synchronize()
        p, st = f.last["passes"][0], f._static
        M0, M1, M2 = f.row_counts()[0]
        N, B = int(st.n.item()), 10 * 10 + st.B
        host = lambda t: t.detach().cpu().numpy()
        # ---- maps of the merged batch (2660 clouds), bit for bit
        coords, pt_off = host(st.coords[:N]).astype(np.float64), host(st.pt_off)
        assert int(pt_off[-1]) == N and np.array_equal(coords[:, 0], np.repeat(np.arange(B), np.diff(pt_off)))
        C0, inv, seg0 = CL.quantize_unique(coords[:, 1:], pt_off)
        C1, par1, seg1 = CL.stride_coords(C0, B, 2)
        C2, par2, seg2 = CL.stride_coords(C1, B, 4)
        assert (C0.shape[0], C1.shape[0], C2.shape[0]) == (M0, M1, M2)
        cm = p.cm
        for s_, C, seg in ((1, C0, seg0), (2, C1, seg1), (4, C2, seg2)):
            assert np.array_equal(host(cm.levels[s_].C[:C.shape[0]]), C) and np.array_equal(host(cm.seg_off(s_)), seg), s_
        assert np.array_equal(host(cm.levels[1].inv[:N]), inv)
        assert np.array_equal(host(cm.parent(2)[:M0]), par1) and np.array_equal(host(cm.parent(4)[:M1]), par2)
        k0, _, p0, _, k1, _, p1 = p.maps[:7]
        nbr0, nbr1, kid1, kid2 = CL.kernel_map(C0, 1), CL.kernel_map(C1, 2), CL.pool_map(C0, C1, 1), CL.pool_map(C1, C2, 2)
        assert np.array_equal(host(k0[:, :M0]), nbr0) and np.array_equal(host(k1[:, :M1]), nbr1)
        assert np.array_equal(host(p0[:, :M1]), kid1) and np.array_equal(host(p1[:, :M2]), kid2)
        assert M1 >= 65536 and int((nbr1 >= 0).sum()) > 500_000
        # ---- forward stages, each from the device's input of that stage (fp64 on the host)
        W0, w1, b1, W1, w5, b5, Wc, bc = [host(t).astype(np.float64) for t in f.network.parameters()]
        F0 = CL.voxel_mean_fwd(host(st.feats[:N]), inv, M0)
        xh0, _ = CL.inorm_fwd(CL.conv_fwd(F0, W0, nbr0), seg0)
        e_xh0 = rel(p.xh0[:M0], xh0)
        P0, arg0 = CL.affine_relu_pool_fwd(host(p.xh0[:M0]), w1, b1, kid1)
        e_P0 = rel(p.P0[:M1], P0)
        same0 = float((host(p.arg0[:M1]) == arg0).mean())
        xh1, _ = CL.inorm_fwd(CL.conv_fwd(host(p.P0[:M1]), W1, nbr1), seg1)
        e_xh1 = rel(p.xh1[:M1], xh1)
        P1, arg1 = CL.affine_relu_pool_fwd(host(p.xh1[:M1]), w5, b5, kid2)
        e_P1 = rel(p.P1[:M2], P1)
        same1 = float((host(p.arg1[:M2]) == arg1).mean())
        print(f"rows {M0} / {M1} / {M2}; x-hat 0 {e_xh0:.1e}, pool 0 {e_P0:.1e} (arg-max equal {same0:.6f}), x-hat 1 {e_xh1:.1e}, pool 1 {e_P1:.1e} "
              f"(arg-max equal {same1:.6f})")
        assert e_xh0 < 2e-5 and e_xh1 < 2e-5 and e_P0 < 1e-6 and e_P1 < 1e-6
        assert same0 > 0.9999 and same1 > 0.9999          # (a tie inside a window, or a pooled 0: either row is a maximum)
        # ---- dense() + Linear per group (the minimum corner is the GROUP's batch, models/MEConv.py:251) and the cross entropy, class 3's real batch
        G, c = f.G, 3
        gco = host(st.gco)
        b0, b1_ = int(gco[G + c]), int(gco[G + c + 1])
        r0, r1 = int(seg2[b0]), int(seg2[b1_])
        # (the device keeps the classifier over the OCCUPIED sites -- 7 x 7 for 28 x 28 digits at stride 4 -- the parameter has all 28 x 28)
        assert tuple(f.site_extent()) == (1, 7, 7) and np.array_equal(host(p.dp.site[r0:r1]), CL.dense_sites(C2[r0:r1], 4, (1, 7, 7)))
        site, S = CL.dense_sites(C2[r0:r1], 4, (1, 28, 28)), 784
        segc = seg2[b0:b1_ + 1] - r0
        logits = CL.sparse_linear_fwd(host(p.P1[r0:r1]), site, segc, S, Wc, bc)
        assert rel(p.logits[b0:b1_], logits) < 1e-5
        labels = np.full(b1_ - b0, f.classes[c], np.int32)
        ce, dl = CL.softmax_ce(logits, labels)
        # first-order gradients of the class's real batch: classifier weight and bias against the device's gradient list
        gW = CL.sparse_linear_bwd_w(dl, host(p.P1[r0:r1]), site, segc, S)
        gwr = f.last["gw_real"]
        assert rel(gwr[7][c], dl.sum(0)) < 2e-5
        assert rel(gwr[6][c].reshape(10, -1), gW) < 2e-5


def test_cfg4_shaped_batch_maps_and_forward_vs_cpp_leg():
    """The same dissection for volumetric clouds (BASELINE configs[3] shape: 1024-point shells in [-1,1]^3, voxel 0.05, all 27 offsets alive;
    8 of the 40 classes x 256 real clouds + ipc 10 = 2 128 clouds, 2.2 M points): the LDS map builder k_cloud_emit<1024> against the C++ leg's
    hash-table builder bit for bit -- the quotient point / voxel size is the device's (st.coords), the floor is the host's --, then the forward
    stages from the device's own inputs."""
    from dd3d_amd.fused_step import FusedDistiller
    from dd3d_amd.models.MEConv import MEConvExp
    from dd3d_amd.synthetic import shell_clouds
    nclass, per, ipc, npts, ext = 8, 256, 10, 1024, (12, 12, 12)
    clouds = shell_clouds(nclass * per, npts, seed=1)
    idx = [list(range(c * per, (c + 1) * per)) for c in range(nclass)]
    syn = (torch.rand((nclass * ipc, npts, 3), generator=torch.Generator().manual_seed(1)) * 1.8 - 0.9).float()
    torch.manual_seed(0)
    net = MEConvExp(3, nclass, num_classes=nclass, dense_extent=ext)
    net.load_state_dict(det_state_dict(net, "net.cfg4s"))
    with FusedDistiller(None, idx, net.to(DEV).train(), graph=True, ipc=ipc, num_classes=nclass, num_points=npts, batch_size=per, device=DEV, seed=5,
                        image_syn=syn, real_clouds=clouds, quantization_size=0.05, clamp_range=(-0.95, 0.95), zero_first_axis=False, dist_lr=1e-6) as f:
        f.outer_step()
        f.outer_step()
        assert f._graph is not None
        torch.cuda.synchronize()
        p, st = f.last["passes"][0], f._static
        M0, M1, M2 = f.row_counts()[0]
        N, B = int(st.n.item()), nclass * ipc + st.B
        host = lambda t: t.detach().cpu().numpy()
        coords, pt_off = host(st.coords[:N]).astype(np.float64), host(st.pt_off)
        C0, inv, seg0 = CL.quantize_unique(coords[:, 1:], pt_off)
        C1, par1, seg1 = CL.stride_coords(C0, B, 2)
        C2, par2, seg2 = CL.stride_coords(C1, B, 4)
        assert (C0.shape[0], C1.shape[0], C2.shape[0]) == (M0, M1, M2) and M0 > 900_000
        cm = p.cm
        for s_, C, seg in ((1, C0, seg0), (2, C1, seg1), (4, C2, seg2)):
            assert np.array_equal(host(cm.levels[s_].C[:C.shape[0]]), C) and np.array_equal(host(cm.seg_off(s_)), seg), s_
        assert np.array_equal(host(cm.levels[1].inv[:N]), inv)
        assert np.array_equal(host(cm.parent(2)[:M0]), par1) and np.array_equal(host(cm.parent(4)[:M1]), par2)
        k0, _, p0, _, k1, _, p1 = p.maps[:7]
        nbr0, nbr1, kid1, kid2 = CL.kernel_map(C0, 1), CL.kernel_map(C1, 2), CL.pool_map(C0, C1, 1), CL.pool_map(C1, C2, 2)
        assert np.array_equal(host(k0[:, :M0]), nbr0) and np.array_equal(host(k1[:, :M1]), nbr1)
        assert np.array_equal(host(p0[:, :M1]), kid1) and np.array_equal(host(p1[:, :M2]), kid2)
        assert all(int((nbr1[k] >= 0).sum()) > 0 for k in range(27))
        W0, w1, b1, W1, w5, b5, Wc, bc = [host(t).astype(np.float64) for t in f.network.parameters()]
        F0 = CL.voxel_mean_fwd(host(st.feats[:N]), inv, M0)
        xh0, _ = CL.inorm_fwd(CL.conv_fwd(F0, W0, nbr0), seg0)
        P0, arg0 = CL.affine_relu_pool_fwd(host(p.xh0[:M0]), w1, b1, kid1)
        xh1, _ = CL.inorm_fwd(CL.conv_fwd(host(p.P0[:M1]), W1, nbr1), seg1)
        P1, arg1 = CL.affine_relu_pool_fwd(host(p.xh1[:M1]), w5, b5, kid2)
        e = [rel(p.xh0[:M0], xh0), rel(p.P0[:M1], P0), rel(p.xh1[:M1], xh1), rel(p.P1[:M2], P1)]
        same = [float((host(p.arg0[:M1]) == arg0).mean()), float((host(p.arg1[:M2]) == arg1).mean())]
        print(f"rows {M0} / {M1} / {M2}; x-hat 0 {e[0]:.1e}, pool 0 {e[1]:.1e}, x-hat 1 {e[2]:.1e}, pool 1 {e[3]:.1e}; arg-max equal {same[0]:.6f} / {same[1]:.6f}")
        assert e[0] < 2e-5 and e[2] < 2e-5 and e[1] < 1e-6 and e[3] < 1e-6 and min(same) > 0.9999


@pytest.mark.parametrize("seed", range(10))
def test_random_ragged_configurations_vs_cpp_leg(seed):
    """Ten random small configurations of the whole step -- 1 to 4 classes (also a subset of them owned), 1 to 9 real clouds per class of
    RAGGED sizes (1 to ``npts`` points: one-point clouds, clouds that share voxels), ipc 1 to 3, 3-D clouds with voxel size 0.05 / 0.1 / 0.5 or
    flat digit images, any of the four metrics, graph or eager -- two steps each against the C++/OpenMP oracle leg in fp64: per-class loss and
    xyz gradient within 1e-4 (a class on a ReLU / arg-max flip would show here as a failure: none of the ten has one)."""
    from dd3d_amd.fused_step import FusedDistiller
    from dd3d_amd.models.MEConv import MEConvExp
    from dd3d_amd.synthetic import shell_clouds, stroke_digits
    rng = np.random.RandomState(100 + seed)
    nclass, per, ipc = int(rng.randint(1, 5)), int(rng.randint(1, 10)), int(rng.randint(1, 4))
    metric = ["modmse", "ours", "mse", "cos"][seed % 4]
    flat = seed % 3 == 0
    own = sorted(rng.choice(nclass, size=int(rng.randint(1, nclass + 1)), replace=False).tolist())
    graph = bool(seed % 2)
    torch.manual_seed(seed)
    if flat:
        npts, q, ext = 250, 1, (1, 28, 28)
        images = torch.cat([stroke_digits(per, seed=300 + seed * 7 + c) for c in range(nclass)])
        images[0] = 0
        images[0, 0, 9, 11] = 1                                             # a one-point cloud
        net = MEConvExp(in_channel=3, out_channel=10)
        kw = dict(images_all=images, real_clouds=None)
        syn = init_synthetic(nclass, ipc, npts, seed=seed)
        clamp = lambda t: t.clamp(0, 27).mul(torch.tensor([0.0, 1.0, 1.0]))
    else:
        npts, q, ext = int(rng.choice([40, 150, 300])), float(rng.choice([0.05, 0.1, 0.5])), (12, 12, 12)
        clouds = [c[:int(rng.randint(1, npts + 1))] for c in shell_clouds(nclass * per, npts, seed=seed)]
        clouds[-1] = clouds[-1][:1]
        net = MEConvExp(3, nclass, num_classes=nclass, dense_extent=ext)
        kw = dict(images_all=None, real_clouds=clouds, quantization_size=q, clamp_range=(-0.95, 0.95), zero_first_axis=False, num_points=npts)
        syn = (torch.rand((nclass * ipc, npts, 3), generator=torch.Generator().manual_seed(seed)) * 1.8 - 0.9).float()
        clamp = lambda t: t.clamp(-0.95, 0.95)
    net.load_state_dict(det_state_dict(net, "net.rand%d" % seed))
    net64 = CL.Network(net, np.float64)
    idx = [list(range(c * per, (c + 1) * per)) for c in range(nclass)]
    images_all = kw.pop("images_all")
    with FusedDistiller(images_all, idx, net.to(DEV).train(), ipc=ipc, num_classes=nclass if not flat else 10, batch_size=per, dist_opt=metric, device=DEV,
                        seed=seed, image_syn=syn, classes=own, graph=graph, dist_lr=1e-3 if not flat else 0.1, **kw) as f:
        draw = np.random.RandomState(seed)
        for step in range(2):
            pts = clamp(f.image_syn.detach().cpu()).double().numpy()
            order = {c: draw.permutation(np.asarray(idx[c]))[:per] for c in own}
            f.outer_step()
            pc, grad = f.last["per_class"].cpu().double().numpy(), f.last["grad"].cpu().double().numpy()
            for gi, c in enumerate(own):
                reals = CL.image_clouds(images_all[order[c]].numpy()) if flat else [clouds[i].double().numpy() for i in order[c]]
                l, g, _, _ = CL.class_step(net64, reals, [pts[c * ipc + i] for i in range(ipc)], c, metric, q)
                mine = grad[c * ipc:(c + 1) * ipc].reshape(-1, 3)
                if l == 0.0:   # one class = a one-output classifier: the cross entropy is constant, every gradient and the loss are exactly zero
                    assert nclass == 1 and pc[gi] == 0.0 and not mine.any() and not g.any()
                    continue
                e_l = abs(pc[gi] - l) / abs(l)
                e_g = float(np.linalg.norm(mine - g) / max(np.linalg.norm(g), 1e-300))
                assert e_l < 1e-4 and e_g < 1e-4, (seed, step, c, metric, flat, e_l, e_g)
        f.check()
