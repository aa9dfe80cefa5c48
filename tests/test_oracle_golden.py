"""The CPU oracle (oracle/ref_model.py over oracle/minkowski_ref.py) against the golden vectors produced by the
reference's own files (tests/golden/make_golden.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import minkowski_ref as ME
from oracle import ref_model as R
from dd3d_amd.synthetic import init_synthetic, stroke_digits
from tests.detgen import det_state_dict, det_tensor, sample_idx

G = os.path.join(os.path.dirname(__file__), "golden")
EXP_SHAPES = [(27, 3, 128), (1, 128), (1, 128), (27, 128, 128), (1, 128), (1, 128), (10, 100352), (10,)]
PYT_SHAPES = [(128, 3, 1, 3, 3), (128,), (128, 128, 1, 3, 3), (128,), (128, 128, 1, 3, 3), (128,), (10, 1152), (10,)]


def grad_lists(shapes, tag, dtype=torch.float32):
    real = [det_tensor(s, f"{tag}.real.{i}", 0.05).to(dtype) for i, s in enumerate(shapes)]
    syn = [det_tensor(s, f"{tag}.syn.{i}", 0.05).to(dtype).requires_grad_() for i, s in enumerate(shapes)]
    return syn, real


@pytest.mark.parametrize("fam,shapes", [("exp", EXP_SHAPES), ("pyt", PYT_SHAPES)])
@pytest.mark.parametrize("pyt", [False, True])
@pytest.mark.parametrize("metric", ["ours", "modmse", "mse", "cos"])
def test_match_loss_oracle_vs_reference(fam, shapes, pyt, metric):
    g = np.load(os.path.join(G, "g1_match_loss.npz"))
    syn, real = grad_lists(shapes, "g1." + fam)
    v = R.match_loss(syn, real, metric, "cpu", pyt)
    grads = torch.autograd.grad(v, syn)
    key = f"{fam}.{int(pyt)}.{metric}"
    assert v.item() == pytest.approx(float(g[key + ".value"]), rel=1e-6)
    for i, gr in enumerate(grads):
        assert gr.double().norm().item() == pytest.approx(g[key + ".gnorm"][i], rel=1e-5, abs=1e-12)
        idx = sample_idx(gr.numel(), f"{fam}.{i}")
        np.testing.assert_allclose(gr.reshape(-1)[idx].numpy(), g[key + ".gsample"][i][:len(idx)], rtol=1e-5, atol=1e-9)


def test_plumbing_oracle_vs_reference():
    g = np.load(os.path.join(G, "g2_plumbing.npz"))
    imgs = stroke_digits(4, seed=11)
    imgs[3] = 0
    imgs[3, 0, 5, 7] = 1
    labels = torch.tensor([3, 1, 4, 1])
    b = R.collate([R.real_item(imgs[i], labels[i]) for i in range(4)])
    np.testing.assert_array_equal(b["coordinates"].numpy(), g["real.coordinates"])
    np.testing.assert_array_equal(b["features"].numpy(), g["real.features"])
    np.testing.assert_array_equal(b["labels"].numpy(), g["real.labels"])
    b20 = R.collate([R.real_item(imgs[i], labels[i], 20) for i in range(4)])
    np.testing.assert_array_equal(b20["coordinates"].numpy(), g["real20.coordinates"])
    syn = init_synthetic(2, 2, num_points=5, seed=3)
    lab = torch.tensor([0, 0, 1, 1])
    sb = R.collate([R.syn_item(syn[i], lab[i]) for i in range(4)])
    np.testing.assert_array_equal(sb["coordinates"].numpy(), g["syn.coordinates"])
    np.testing.assert_array_equal(sb["features"].numpy(), g["syn.features"])
    tf = R.create_input_batch({k: v.clone() for k, v in b.items()}, True, "cpu", 0.5)
    np.testing.assert_array_equal(tf.C.numpy(), g["field.C.q05"])


def _run_step(cls, pyt, metric, dtype=torch.float32):
    torch.manual_seed(0)
    net = cls(3, 10) if pyt else cls(in_channel=3, out_channel=10)
    net.load_state_dict(det_state_dict(net, "net." + cls.__name__))
    net = net.to(dtype).train()
    reals = [stroke_digits(4, seed=100 + c) for c in range(2)]
    image_syn = init_synthetic(2, 2, num_points=250, seed=5)
    R.clamp_synthetic_(image_syn)
    image_syn = image_syn.to(dtype).requires_grad_()
    return R.matching_step(net, image_syn, reals, 2, metric, pyt)


@pytest.mark.parametrize("metric", ["modmse", "ours"])
def test_mepytorch_step_oracle_vs_reference(metric):
    g = np.load(os.path.join(G, "g3_mepytorch.npz"))
    out = _run_step(R.MEPytorch, True, metric)
    assert out["loss"].item() == pytest.approx(float(g[metric + ".loss"]), rel=2e-5)
    np.testing.assert_allclose(out["per_class"].numpy(), g[metric + ".per_class"], rtol=2e-5)
    ref = g[metric + ".grad"]
    err = np.abs(out["grad"].numpy() - ref).max() / (np.abs(ref).max() + 1e-30)
    assert err < 1e-4, err
    for c in range(2):
        np.testing.assert_allclose(out["logits"][c][0].numpy(), g[metric + ".logits_real"][c], rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose([t.double().norm().item() for t in out["gw_real"][c]], g[metric + ".gwr_norm"][c], rtol=1e-4)


@pytest.mark.parametrize("metric", ["modmse", "ours"])
def test_meconvexp_step_oracle_vs_reference(metric):
    g = np.load(os.path.join(G, "g4_meconvexp.npz"))
    out = _run_step(R.MEConvExp, False, metric)
    assert out["loss"].item() == pytest.approx(float(g[metric + ".loss"]), rel=2e-5)
    ref = g[metric + ".grad"]
    err = np.abs(out["grad"].numpy() - ref).max() / (np.abs(ref).max() + 1e-30)
    assert err < 1e-4, err
    for c in range(2):
        np.testing.assert_allclose(out["logits"][c][1].numpy(), g[metric + ".logits_syn"][c], rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose([t.double().norm().item() for t in out["gw_syn"][c]], g[metric + ".gws_norm"][c], rtol=1e-4)


def test_constructor_names_and_shapes():
    g = np.load(os.path.join(G, "g5_ctor.npz"))
    for name, net in (("MEConvExp", R.MEConvExp(in_channel=3, out_channel=10)), ("MEPytorch", R.MEPytorch(3, 10))):
        assert [n for n, _ in net.named_parameters()] == list(g[name + ".names"])
        assert [str(tuple(p.shape)) for p in net.parameters()] == list(g[name + ".shapes"])
        assert list(net.state_dict().keys()) == list(g[name + ".sd_keys"])


# ---------------- known-answer tests for the (unpinned) MinkowskiEngine semantics, SURVEY 8c -----------------

def test_two_points_one_voxel_mean_and_shared_gradient():
    coords = torch.tensor([[0, 0, 3.2, 4.9], [0, 0, 3.7, 4.1], [0, 0, 5.0, 5.0]])
    feats = torch.tensor([[1.0, 2.0, 3.0], [3.0, 6.0, 9.0], [7.0, 7.0, 7.0]], requires_grad=True)
    C0, inv, uidx = ME.quantize_unique(coords)
    assert C0.tolist() == [[0, 0, 3, 4], [0, 0, 5, 5]] and inv.tolist() == [0, 0, 1] and uidx.tolist() == [0, 2]
    F0 = ME.VoxelMeanFn.apply(feats, inv, 2)
    assert F0.tolist() == [[2.0, 4.0, 6.0], [7.0, 7.0, 7.0]]
    F0.sum().backward()
    assert feats.grad.tolist() == [[0.5] * 3, [0.5] * 3, [1.0] * 3]


def test_negative_coordinates_floor_not_truncate():
    C0, inv, _ = ME.quantize_unique(torch.tensor([[0, -0.5, -1.0, 0.5], [0, -0.2, -0.9, 0.1]]))
    assert C0.tolist() == [[0, -1, -1, 0]] and inv.tolist() == [0, 0]
    Cout, parent = ME.stride_map(torch.tensor([[0, -1, -2, 3], [0, -3, 1, 2]], dtype=torch.int32), 2)
    assert Cout.tolist() == [[0, -2, -2, 2], [0, -4, 0, 2]] and parent.tolist() == [0, 1]


def test_plus_shaped_kernel_map_by_hand():
    # plus shape in the (y,x) plane, z == 0: rows 0..4 = centre, up(y-1), down(y+1), left(x-1), right(x+1)
    C = torch.tensor([[0, 0, 5, 5], [0, 0, 4, 5], [0, 0, 6, 5], [0, 0, 5, 4], [0, 0, 5, 6]], dtype=torch.int32)
    nbr = ME.kernel_map(C, C, 3, 1)
    assert torch.equal(nbr, ME.kernel_map_bruteforce(C, C, 3, 1))
    k = lambda d0, d1, d2: (d0 + 1) + 3 * (d1 + 1) + 9 * (d2 + 1)
    assert nbr[k(0, 0, 0)].tolist() == [0, 1, 2, 3, 4]
    assert nbr[k(0, -1, 0)].tolist() == [1, -1, 0, -1, -1]   # neighbour at y-1
    assert nbr[k(0, 1, 0)].tolist() == [2, 0, -1, -1, -1]
    assert nbr[k(0, 0, -1)].tolist() == [3, -1, -1, -1, 0]
    assert nbr[k(0, 0, 1)].tolist() == [4, -1, -1, 0, -1]
    assert int((nbr >= 0).sum()) == 5 + 8 + 8   # self, centre<->arm, arm<->arm diagonals
    ins, outs, offs = ME.pairs_from_nbr(nbr)
    assert offs[-1].item() == 21 and all(np.diff(outs[offs[k]:offs[k + 1]].numpy()).min(initial=1) > 0 for k in range(27))


def test_single_voxel_cloud_instance_norm_is_zero():
    X = torch.tensor([[3.0, -2.0], [1.0, 5.0], [2.0, 7.0]])
    out = ME.InstanceNormFn.apply(X, torch.tensor([0, 1, 1]), 2)
    assert out[0].tolist() == [0.0, 0.0]
    np.testing.assert_allclose(out[1:].numpy(), [[-1, -1], [1, 1]], rtol=1e-6)


def test_pool_negative_max_and_absent_children():
    C = torch.tensor([[0, 0, 0, 0], [0, 0, 0, 1], [0, 0, 1, 0], [0, 0, 1, 1], [0, 0, 2, 2]], dtype=torch.int32)
    Cout, parent = ME.stride_map(C, 2)
    assert Cout.tolist() == [[0, 0, 0, 0], [0, 0, 2, 2]] and parent.tolist() == [0, 0, 0, 0, 1]
    kids = ME.kernel_map(C, Cout, 2, 1)
    X = torch.tensor([[-4.0], [-1.5], [-3.0], [-2.0], [-9.0]], requires_grad=True)
    out, arg = ME.MaxPoolFn.apply(X, kids)
    assert out.tolist() == [[-1.5], [-9.0]] and arg.tolist() == [[1], [4]]   # absent children are not zeros
    out.sum().backward()
    assert X.grad.reshape(-1).tolist() == [0, 1, 0, 0, 1]


def test_dense_min_corner_shift():
    C = torch.tensor([[0, 0, 4, 8], [1, 0, 8, 12], [1, 0, 4, 8]], dtype=torch.int32)
    st = ME.SparseTensor(torch.tensor([[1.0], [2.0], [3.0]]), C, 4)
    minc = C.min(0, keepdim=True)[0][:, 1:]
    d = st.dense(shape=torch.Size([2, 1, 1, 28, 28]), min_coordinate=minc)[0]
    assert d.shape == (2, 1, 1, 28, 28) and d[0, 0, 0, 0, 0] == 1 and d[1, 0, 0, 1, 1] == 2 and d[1, 0, 0, 0, 0] == 3
    assert d.sum() == 6


def test_opaque_backward_truncates_second_order():
    """ME's Function backward outputs are constants under create_graph=True."""
    X = torch.randn(6, 3, dtype=torch.float64, requires_grad=True)
    W = torch.randn(27, 3, 4, dtype=torch.float64, requires_grad=True)
    C = torch.tensor([[0, 0, i, 0] for i in range(6)], dtype=torch.int32)
    nbr = ME.kernel_map(C, C, 3, 1)
    Y = ME.SparseConvFn.apply(X, W, nbr)
    (gw,) = torch.autograd.grad((Y ** 2).sum(), [W], create_graph=True)
    assert not gw.requires_grad


def test_report_table2_gradient_magnitude_anchor_at_iteration_0():
    """The only number the reference holds that touches the (unpinned) MinkowskiEngine semantics: Report Table 2, mean |gradient| on the
    synthetic xyz with MinkowskiEngine layers -- 4e-4 for the layer-wise cosine ('ours'), 0.033 for the sum of squares ('modmse') --
    protocol: 1 synthetic cloud per class, Kaiming init, values over 1250 iterations (/root/reference/README.md:12).
    The CPU oracle can afford iteration 0 only (4 s per step); the full 1250-iteration protocol runs on the GPU
    (tests/test_gpu_step.py::test_report_table2_anchor_over_1250_iterations).  At iteration 0, on stroke digits instead of MNIST:
    'ours' 4.9e-4 (report 4e-4); 'modmse' 0.55 -- the sum of squares starts high and decays as the clouds move (the GPU test records the
    average), so it is only bounded from below here."""
    import numpy as np
    from dd3d_amd.synthetic import init_synthetic, mnist3d_like
    images, _, idx = mnist3d_like(10, 256, seed=0)
    torch.manual_seed(0)
    net = R.MEConvExp(in_channel=3, out_channel=10).train()   # the reference's own initialisation (MEConv.py:232-242)
    got = {}
    for metric in ("ours", "modmse"):
        syn = init_synthetic(10, 1, 250, seed=0)
        R.clamp_synthetic_(syn)
        syn.requires_grad_()
        out = R.matching_step(net, syn, [images[np.asarray(idx[c])] for c in range(10)], 1, metric, False)
        got[metric] = float(out["grad"].abs().mean())
    print("mean |xyz-grad| at iteration 0:", got)
    assert 4e-5 < got["ours"] < 4e-3          # within a decade of the report's 4e-4
    assert 0.033 / 10 < got["modmse"] < 3.3    # iteration 0 sits above the 1250-iteration average of 0.033
