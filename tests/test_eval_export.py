"""Evaluation block and export against the reference's own functions (tests/golden/g6_eval.npz,
made by tests/golden/make_golden.py from models/MNIST.py and mnist_distill.py) and against the oracle."""
import os

import numpy as np
import pytest
import torch

from tests.detgen import det_tensor, eval_inputs

G6 = np.load(os.path.join(os.path.dirname(__file__), "golden", "g6_eval.npz"))


def _convnet():
    from dd3d_amd.models.MNIST import ConvNet
    return ConvNet(1, 10, 128, 3, "relu", "instancenorm", "avgpooling")  # mnist_distill.py:326


def test_convnet_matches_reference_constructor_and_forward():
    torch.manual_seed(0)
    net = _convnet()
    assert [n for n, _ in net.named_parameters()] == list(G6["convnet.names"])
    assert [str(tuple(p.shape)) for p in net.parameters()] == list(G6["convnet.shapes"])
    assert list(net.state_dict().keys()) == list(G6["convnet.sd_keys"])
    assert repr(net) == str(G6["convnet.repr"])
    _, imgs, _ = eval_inputs()
    net.eval()
    with torch.no_grad():
        out = net(imgs).numpy()
    # same seed + same construction order = same initial weights as the reference's class: the forward agrees to fp32 round-off
    np.testing.assert_allclose(out, G6["convnet.out"], rtol=1e-5, atol=1e-6)


def test_convnet_unknown_options_exit_like_reference():
    from dd3d_amd.models.MNIST import ConvNet
    for bad in (dict(net_act="gelu"), dict(net_norm="rmsnorm"), dict(net_pooling="l2pooling")):
        kw = dict(net_act="relu", net_norm="instancenorm", net_pooling="avgpooling")
        kw.update(bad)
        with pytest.raises(SystemExit):
            ConvNet(1, 10, 16, 2, kw["net_act"], kw["net_norm"], kw["net_pooling"])


def test_oracle_populate_img_matches_reference():
    from oracle.ref_model import populate_img
    clouds, _, _ = eval_inputs()
    img = populate_img(clouds, torch.zeros(clouds.shape[0], 1, 28, 28))
    assert np.array_equal(img.numpy(), G6["populate.img"])


def test_augment_matches_reference_draw_for_draw():
    from dd3d_amd.evaluation import augment
    _, imgs, _ = eval_inputs()
    np.random.seed(5)
    torch.manual_seed(5)
    got = augment(imgs.clone(), torch.device("cpu"))
    np.testing.assert_allclose(got.numpy(), G6["augment.out"], rtol=0, atol=1e-6)


def test_epoch_matches_reference_on_cpu():
    from dd3d_amd import evaluation
    from dd3d_amd.configs import settings
    from dd3d_amd.utils.Mnist2D import TensorDataset
    _, imgs, labels = eval_inputs()
    old = settings.device
    settings.device = torch.device("cpu")
    try:
        np.random.seed(7)
        torch.manual_seed(7)
        net = _convnet()
        opt = torch.optim.SGD(net.parameters(), lr=0.01, momentum=0.9, weight_decay=0.0005)
        crit = torch.nn.CrossEntropyLoss()
        loader = torch.utils.data.DataLoader(TensorDataset(imgs.clone(), labels), batch_size=256, shuffle=True, num_workers=0)
        rec = [evaluation.epoch("train", loader, net, opt, crit, aug=False) for _ in range(3)]
        rec.append(evaluation.epoch("test", loader, net, opt, crit, aug=False))
    finally:
        settings.device = old
    np.testing.assert_allclose(np.array(rec), G6["epoch.loss_acc"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose([p.detach().norm().item() for p in net.parameters()], G6["epoch.w_norms"], rtol=1e-4)


def test_ply_layout_and_round_trip(tmp_path):
    from dd3d_amd import export
    pts = det_tensor((3, 7, 3), "ply").double()
    export.save_cad(pts * 0.3, str(tmp_path), 400, normalize=True)
    assert sorted(os.listdir(tmp_path)) == ["0_400.ply", "1_400.ply", "2_400.ply"]  # configs/settings.py:214-220
    raw = open(tmp_path / "1_400.ply", "rb").read()
    head = (b"ply\nformat binary_little_endian 1.0\ncomment Created by Open3D\nelement vertex 7\n"
            b"property double x\nproperty double y\nproperty double z\nend_header\n")
    assert raw.startswith(head) and len(raw) == len(head) + 7 * 3 * 8
    np.testing.assert_array_equal(export.read_ply(str(tmp_path / "1_400.ply")), np.clip((pts[1] * 0.3).numpy(), -1, 1))
    export.save_cad(pts * 5, str(tmp_path), 1, normalize=False)
    np.testing.assert_array_equal(export.read_ply(str(tmp_path / "2_1.ply")), (pts[2] * 5).numpy())
    img = torch.zeros(1, 28, 28)  # images_all[i] is [1,28,28] (mnist_distill.py:214): the points come out as (0, y, x)
    img[0, 3, 4] = img[0, 10, 20] = 1
    export.export_mnist(img, 7, str(tmp_path), 12, custom_name="2pt")
    got = export.read_ply(str(tmp_path / "original-2pt-7_12.ply"))  # mnist_distill.py:34-43: argwhere of the batch-of-one array
    np.testing.assert_array_equal(got, [[0, 3, 4], [0, 10, 20]])
    export.export_pcd(pts, 3, str(tmp_path), 0)
    np.testing.assert_array_equal(export.read_ply(str(tmp_path / "original-orig-3_0.ply")), pts[0].numpy())


def test_checkpoint_keys_and_cross_loading(tmp_path):
    from dd3d_amd import export
    from dd3d_amd.models.MEConv import MEConvExp
    net = MEConvExp(in_channel=3, out_channel=10)
    syn = det_tensor((20, 250, 3), "ck").requires_grad_()
    lab = torch.arange(10).repeat_interleave(2)
    opt = torch.optim.SGD([syn], lr=0.1, momentum=0.5)
    syn.sum().backward()
    opt.step()
    path = str(tmp_path / "x.pth")
    export.save_checkpoint(path, 99, 1.5, net, opt, syn, lab)
    state = torch.load(path, weights_only=True)
    assert list(state.keys()) == ["epoch", "least_loss", "model_state_dict", "optimizer_state_dict", "synthetic_xyz", "synthetic_labels"]  # :390-397
    assert state["epoch"] == 100 and state["least_loss"] == 1.5
    assert list(state["model_state_dict"].keys()) == ["features.0.kernel", "features.1.weight", "features.1.bias", "features.4.kernel",
                                                       "features.5.weight", "features.5.bias", "classifier.weight", "classifier.bias"]
    assert tuple(state["model_state_dict"]["features.4.kernel"].shape) == (27, 128, 128)
    net2 = MEConvExp(in_channel=3, out_channel=10)
    opt2 = torch.optim.SGD([syn.detach().clone().requires_grad_()], lr=0.1, momentum=0.5)
    got = export.load_checkpoint(path, net2, opt2)
    assert torch.equal(net2.classifier.weight, net.classifier.weight) and torch.equal(got["synthetic_xyz"], syn)
    assert torch.equal(opt2.state_dict()["state"][0]["momentum_buffer"], opt.state_dict()["state"][0]["momentum_buffer"])


def test_image_grid(tmp_path):
    from dd3d_amd import export
    imgs = torch.rand(7, 1, 28, 28)
    grid = export.make_grid(imgs, nrow=3)
    assert tuple(grid.shape) == (3, 3 * 30 + 2, 3 * 30 + 2)  # torchvision: padding 2 around and between
    assert torch.equal(grid[1, 2:30, 32:60], imgs[1, 0]) and float(grid[:, :2].abs().sum()) == 0
    export.save_image(imgs, str(tmp_path / "vis_iter0.png"), nrow=3)
    from PIL import Image
    assert Image.open(tmp_path / "vis_iter0.png").size == (92, 92)


@pytest.mark.gpu
def test_populate_img_kernel_matches_reference_and_oracle():
    from dd3d_amd.evaluation import populate_img
    from oracle.ref_model import populate_img as oracle_populate
    dev = torch.device("cuda:0")
    clouds, _, _ = eval_inputs()
    img = populate_img(clouds.to(dev), torch.zeros(clouds.shape[0], 1, 28, 28, device=dev))
    assert np.array_equal(img.cpu().numpy(), G6["populate.img"])
    # bench-sized set (100 clouds x 250 points), with points at the borders and negative (wrapping) coordinates
    big = torch.rand(100, 250, 3) * 28
    big[:, :5, 1:] = -big[:, :5, 1:]
    big[0, 5] = torch.tensor([0.0, -0.9, 27.999])
    big[0, 6] = torch.tensor([0.0, -28.0, 0.0])
    want = oracle_populate(big, torch.zeros(100, 1, 28, 28))
    got = populate_img(big.to(dev), torch.zeros(100, 1, 28, 28, device=dev))
    assert torch.equal(got.cpu(), want)
    bad = big.clone()
    bad[3, 3, 2] = 28.0
    with pytest.raises(IndexError):
        populate_img(bad.to(dev), torch.zeros(100, 1, 28, 28, device=dev))
    with pytest.raises(IndexError):
        oracle_populate(bad, torch.zeros(100, 1, 28, 28))
    with pytest.raises(RuntimeError):
        populate_img(big, torch.zeros(100, 1, 28, 28))  # host tensors: no CPU path


@pytest.mark.gpu
def test_evaluate_synset_runs_on_gpu(tmp_path):
    """A short evaluation (Epoch=20 instead of 1000) on stroke digits: the ConvNet must fit the 20 rasterised synthetic clouds."""
    from dd3d_amd import evaluation
    from dd3d_amd.configs import settings
    from dd3d_amd.synthetic import mnist3d_like
    from dd3d_amd.utils.Mnist2D import TensorDataset
    dev = torch.device("cuda:0")
    old = settings.device, settings.cad_per_class
    settings.device, settings.cad_per_class = dev, 2
    try:
        images, labels, _ = mnist3d_like(10, 12, seed=3)
        imgs = images.reshape(-1, 1, 28, 28).float()
        # "synthetic" clouds = the on-pixels of two real digits per class, padded by repeating the first point
        clouds, labs = [], []
        for c in range(10):
            for j in range(2):
                ys, xs = torch.nonzero(imgs[(labels == c).nonzero()[j, 0], 0] == 1, as_tuple=True)
                p = torch.stack([torch.zeros_like(ys), ys, xs], 1).float()[:250]
                clouds.append(torch.cat([p, p[:1].expand(250 - p.shape[0], 3)]))
                labs.append(c)
        testloader = torch.utils.data.DataLoader(TensorDataset(imgs, labels), batch_size=256, shuffle=False)
        np.random.seed(0)
        torch.manual_seed(0)
        net = _convnet()
        net, acc_train, acc_test = evaluation.evaluate_synset(0, net, torch.stack(clouds), torch.tensor(labs), testloader, 0, Epoch=20,
                                                              export_cad_dir=str(tmp_path))
        assert os.path.exists(tmp_path / "vis_iter0.png")
        assert 0.0 <= acc_test <= 1.0 and acc_train >= 0.5
    finally:
        settings.device, settings.cad_per_class = old


@pytest.mark.gpu
def test_main_runs_the_fused_graph_path_with_eval_and_export(tmp_path):
    """mnist_distill.main() end to end for a few iterations: MEConvExp goes through FusedDistiller (the HIP-graph step bench.py times),
    the evaluation block and the .ply export run, and the checkpoint holds the reference's six keys with a torch-SGD optimizer state that
    loads into torch.optim.SGD (mnist_distill.py:390-398)."""
    from dd3d_amd import export, mnist_distill
    from dd3d_amd.fused_step import FusedDistiller
    ini = tmp_path / "t.ini"
    ini.write_text("[DEFAULT]\nlogging_parent = %s\nexperiment_name = t\n[MODEL]\nbatch_size = 32\ndist_opt = modmse\ndist_lr = 0.1\n[SHAPENET]\nnum_points = 250\n"
                   "[DISTILLATION]\ncad_per_class = 2\ntotal_iterations = 5\nsave_cad_every = 2\nexport_folder_name = output/\n"
                   "distillation_checkpoint_dir = %s\neval_epochs = 2\nnum_eval = 1\n" % (tmp_path / "logs", tmp_path / "ckpt"))
    d = mnist_distill.main(["--no-log"], str(ini))
    assert isinstance(d, FusedDistiller) and d._graph is not None and d.steps_done == 5
    d.check()
    ck = list((tmp_path / "ckpt").glob("*.pth"))
    assert len(ck) == 1
    state = torch.load(str(ck[0]), weights_only=True)
    assert list(state.keys()) == ["epoch", "least_loss", "model_state_dict", "optimizer_state_dict", "synthetic_xyz", "synthetic_labels"]
    assert tuple(state["synthetic_xyz"].shape) == (20, 250, 3) and state["synthetic_labels"].tolist() == [c for c in range(10) for _ in range(2)]
    p = state["synthetic_xyz"].clone().requires_grad_()
    opt = torch.optim.SGD([p], lr=0.5, momentum=0.9)
    opt.load_state_dict(state["optimizer_state_dict"])      # torch's own loader accepts it
    assert opt.param_groups[0]["lr"] == 0.1 and opt.param_groups[0]["momentum"] == 0.5
    assert tuple(opt.state_dict()["state"][0]["momentum_buffer"].shape) == (20, 250, 3)
    plys = list((tmp_path / "logs").rglob("*.ply"))
    assert len(plys) >= 20 and export.read_ply(str(plys[0])).shape == (250, 3)
