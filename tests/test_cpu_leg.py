"""The oracle's C++/OpenMP leg (oracle/cpu/dd3d_cpu.cpp: one class of the matching step of MEConvExp) against
the torch restatement that the golden fixtures pin (oracle/ref_model.py over oracle/minkowski_ref.py).  The two share no code: maps by a
packed-key hash table here, by sorted keys there; hand-written loops here, torch autograd there -- in particular the second-order pass is
written out by hand here and produced by autograd around opaque Functions there.  CPU only."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from detgen import det_state_dict  # noqa: E402
from oracle import cpu_leg as CL  # noqa: E402
from oracle import ref_model as R  # noqa: E402
from dd3d_amd.synthetic import init_synthetic, shell_clouds, stroke_digits  # noqa: E402


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


def _mnist_case(nclass=3, per=6, ipc=2):
    onet = R.MEConvExp(in_channel=3, out_channel=10)
    onet.load_state_dict(det_state_dict(onet, "net.MEConvExp"))
    onet = onet.double().train()
    reals = [stroke_digits(per, seed=200 + c) for c in range(nclass)]
    syn = init_synthetic(nclass, ipc, num_points=250, seed=5).double()
    R.clamp_synthetic_(syn)
    return onet, reals, syn, ipc


@pytest.mark.parametrize("metric", ["modmse", "ours", "mse", "cos"])
def test_cpu_leg_equals_the_torch_restatement_fp64(metric):
    onet, reals, syn, ipc = _mnist_case()
    ref = R.matching_step(onet, syn.clone().requires_grad_(), reals, ipc, metric, False, coord_dtype=torch.float32)
    got = CL.matching_step(CL.Network(onet, np.float64), syn.numpy(), [r.numpy() for r in reals], ipc, metric, want_lists=True)
    assert got["loss"] == pytest.approx(ref["loss"].item(), rel=1e-12)
    assert rel(got["per_class"], ref["per_class"].numpy()) < 1e-12
    assert rel(got["grad"], ref["grad"].numpy()) < 1e-11
    for c in range(len(reals)):
        for i in range(8):
            assert rel(got["gw_real"][c][i], ref["gw_real"][c][i].numpy()) < 1e-11, (c, i)
            assert rel(got["gw_syn"][c][i], ref["gw_syn"][c][i].numpy()) < 1e-11, (c, i)


def test_cpu_leg_volumetric_clouds_and_other_geometry():
    """3-D clouds (negative coordinates, voxel size 0.05, a 12^3 dense extent, 3 outputs, labels given by class_ids): every map is 3-D."""
    nclass, ext, vox = 3, (12, 12, 12), 0.05
    onet = R.MEConvExp(3, nclass, num_classes=nclass, dense_extent=ext)
    onet.load_state_dict(det_state_dict(onet, "net.split"))
    onet = onet.double().train()
    reals = [[c.double() for c in shell_clouds(4, 150, seed=30 + k)] for k in range(nclass)]
    syn = torch.stack([c.double() for c in shell_clouds(nclass * 2, 120, seed=77)])
    ref = R.matching_step(onet, syn.clone().requires_grad_(), reals, 2, "ours", False, num_points=1024, quantization_size=vox,
                          class_ids=[2, 0, 1], coord_dtype=torch.float32)
    got = CL.matching_step(CL.Network(onet, np.float64), syn.numpy(), [[c.numpy() for c in r] for r in reals], 2, "ours", num_points=1024,
                           quantization_size=vox, class_ids=[2, 0, 1])
    assert got["loss"] == pytest.approx(ref["loss"].item(), rel=1e-12)
    assert rel(got["grad"], ref["grad"].numpy()) < 1e-11


def test_cpu_leg_fp32_is_within_the_step_tolerance_of_fp64_and_thread_count_does_not_matter():
    onet, reals, syn, ipc = _mnist_case()
    args = (syn.numpy(), [r.numpy() for r in reals], ipc, "modmse")
    a = CL.matching_step(CL.Network(onet, np.float64), *args)
    b = CL.matching_step(CL.Network(onet, np.float32), *args)
    assert b["loss"] == pytest.approx(a["loss"], rel=1e-4) and rel(b["grad"], a["grad"]) < 1e-4     # north_star's tolerance
    n = CL.lib().dd3d_cpu_max_threads()
    try:
        CL.lib().dd3d_cpu_set_threads(1)
        c = CL.matching_step(CL.Network(onet, np.float64), *args)
    finally:
        CL.lib().dd3d_cpu_set_threads(n)
    assert c["loss"] == pytest.approx(a["loss"], rel=1e-13) and rel(c["grad"], a["grad"]) < 1e-12


def test_cpu_leg_refuses_what_dense_refuses():
    """A stride-4 voxel outside the dense extent: MinkowskiEngine's dense() raises (models/MEConv.py:254), so does the leg; bad arguments."""
    onet, reals, syn, ipc = _mnist_case(nclass=1)
    net = CL.Network(onet, np.float64)
    far = syn.numpy().copy()
    far[0, 0] = [0.0, 200.0, 3.0]
    with pytest.raises(RuntimeError):
        CL.matching_step(net, far, [reals[0].numpy()], ipc, "modmse")
    with pytest.raises(ValueError):
        CL.class_step(net, CL.image_clouds(reals[0].numpy()), [syn.numpy()[0]], 10, "modmse")      # label outside the classifier's outputs
    wide = syn.numpy().copy()
    wide[0, 0] = [0.0, 4.0e4, 3.0]
    with pytest.raises(ValueError, match="16-bit"):
        CL.matching_step(net, wide, [reals[0].numpy()], ipc, "modmse")                              # beyond the packed voxel keys: refused, not aliased
    with pytest.raises(ValueError):
        CL.class_step(net, [np.zeros((0, 3))] * 4, [syn.numpy()[0]], 0, "modmse")                   # a real batch without a single point


@pytest.mark.parametrize("seed", range(8))
def test_cpu_leg_random_edge_cases_vs_torch_restatement(seed):
    """Seeded random geometry, both legs in fp64: clouds of 1..40 points with negative coordinates, duplicate points, clouds that collapse
    into ONE voxel (zero variance: x-hat = 0 * 1e4), one or several clouds per class, a random voxel size, a dense extent that just holds
    them, 2..4 outputs, every metric."""
    rng = np.random.default_rng(100 + seed)
    nclass = int(rng.integers(2, 5))
    ipc = int(rng.integers(1, 3))
    vox = float(rng.choice([1.0, 0.5, 0.25]))
    span = 10.0 * vox                                   # stride-4 voxels span 4 * vox: at most 4 sites per axis
    ext = (4, 4, 4)
    metric = ["modmse", "ours", "mse", "cos"][seed % 4]

    def cloud():
        n = int(rng.integers(1, 41))
        kind = rng.integers(0, 4)
        if kind == 0:                                   # everything in one voxel
            pts = rng.uniform(0.1 * vox, 0.9 * vox, size=(n, 3)) + np.floor(rng.uniform(-span / 2, span / 2 - vox, size=3) / vox) * vox
        else:
            pts = rng.uniform(-span / 2, span / 2 - 1e-3, size=(n, 3))
        if kind == 1 and n > 2:                         # exact duplicates
            pts[n // 2:] = pts[:n - n // 2]
        return np.round(pts, 4)

    onet = R.MEConvExp(3, nclass, num_classes=nclass, dense_extent=ext)
    onet.load_state_dict(det_state_dict(onet, "net.edge%d" % seed))
    onet = onet.double().train()
    reals = [[cloud() for _ in range(int(rng.integers(1, 5)))] for _ in range(nclass)]
    npts = 24
    syn = np.stack([np.round(rng.uniform(-span / 2, span / 2 - 1e-3, size=(npts, 3)), 4) for _ in range(nclass * ipc)])
    ref = R.matching_step(onet, torch.tensor(syn).requires_grad_(), [[torch.tensor(c) for c in r] for r in reals], ipc, metric, False,
                          num_points=1024, quantization_size=vox, coord_dtype=torch.float32)
    got = CL.matching_step(CL.Network(onet, np.float64), syn, reals, ipc, metric, num_points=1024, quantization_size=vox, want_lists=True)
    # (tolerance: a nearly constant channel of a tiny cloud has 1 / sqrt(var + 1e-8) up to 1e4, at both norm layers and again in their
    # backward passes, which magnifies the two legs' different fp64 rounding to 1e-7 ... 1e-6 here; on ordinary clouds they agree to 1e-12)
    assert np.isfinite(got["loss"]) and got["loss"] == pytest.approx(ref["loss"].item(), rel=1e-6)
    g = ref["grad"].numpy()
    assert np.abs(got["grad"] - g).max() <= 1e-5 * max(np.abs(g).max(), 1e-30)
    for c in range(nclass):
        for i in range(8):
            assert rel(got["gw_syn"][c][i], ref["gw_syn"][c][i].numpy()) < 1e-5, (c, i)


@pytest.mark.parametrize("metric", ["modmse", "ours"])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_cpu_leg_vs_the_reference_golden_step(metric, dtype):
    """The C++ leg DIRECTLY against tests/golden/g4_meconvexp.npz -- the matching step computed by the reference's own models/MEConv.py,
    distillation_loss.py and mnist_distill.py loop body over the MinkowskiEngine restatement (tests/golden/make_golden.py; fp32) -- with the
    inputs of tests/test_oracle_golden.py::test_meconvexp_step_oracle_vs_reference: loss, xyz gradient and the norm of every synthetic
    gradient entry, at the tolerances that test holds the torch leg to."""
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g4_meconvexp.npz"))
    torch.manual_seed(0)
    onet = R.MEConvExp(in_channel=3, out_channel=10)
    onet.load_state_dict(det_state_dict(onet, "net.MEConvExp"))
    reals = [stroke_digits(4, seed=100 + c).numpy() for c in range(2)]
    syn = init_synthetic(2, 2, num_points=250, seed=5)
    R.clamp_synthetic_(syn)
    out = CL.matching_step(CL.Network(onet, dtype), syn.numpy(), reals, 2, metric, want_lists=True)
    assert out["loss"] == pytest.approx(float(g[metric + ".loss"]), rel=2e-5)
    ref = g[metric + ".grad"]
    assert np.abs(out["grad"] - ref).max() / np.abs(ref).max() < 1e-4
    for c in range(2):
        np.testing.assert_allclose([np.linalg.norm(t.astype(np.float64)) for t in out["gw_syn"][c]], g[metric + ".gws_norm"][c], rtol=1e-4)
