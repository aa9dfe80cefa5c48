"""Evaluation block on the GPU next to the reference's way of doing it on the host:
  * populate_img: dd3d_populate_img (one launch) vs the reference's per-point Python assignments, 100 x 250 points
  * one training epoch of the evaluation ConvNet on 100 rasterised clouds (augment + forward + backward + SGD), GPU vs the same code on
    the host cores
usage: python tools/evalbench.py"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dd3d_amd import evaluation
from dd3d_amd.configs import settings
from dd3d_amd.models.MNIST import ConvNet
from dd3d_amd.utils.Mnist2D import TensorDataset

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
clouds = torch.rand((100, 250, 3), generator=g) * 27.99
clouds[:, :, 0] = 0
labels = torch.arange(10).repeat_interleave(10)

t0 = time.perf_counter()
ref = torch.zeros(100, 1, 28, 28)
for idx, digit in enumerate(clouds):          # the reference's way (mnist_distill.py:163-168): one tensor assignment per point
    for pt in digit:
        q = pt.type(torch.LongTensor)
        ref[idx, 0, q[1], q[2]] = 1
t_ref = time.perf_counter() - t0
cd, canvas = clouds.to(dev), torch.zeros(100, 1, 28, 28, device=dev)
evaluation.populate_img(cd, canvas)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    evaluation.populate_img(cd, canvas)
e1.record(); torch.cuda.synchronize()
assert torch.equal(canvas.cpu(), ref)
print(f"populate_img 100 x 250 points: GPU {e0.elapsed_time(e1) / 50 * 1e3:.1f} us per call (incl. the error-flag read-back), host loop {t_ref * 1e3:.1f} ms")

def epochs(device, n):
    settings.device = device
    np.random.seed(0); torch.manual_seed(0)
    net = ConvNet(1, 10, 128, 3, "relu", "instancenorm", "avgpooling").to(device)
    opt = torch.optim.SGD(net.parameters(), lr=0.01, momentum=0.9, weight_decay=0.0005)
    crit = torch.nn.CrossEntropyLoss()
    loader = torch.utils.data.DataLoader(TensorDataset(ref.to(device), labels.to(device)), batch_size=256, shuffle=True)
    evaluation.epoch("train", loader, net, opt, crit, aug=False)   # warm-up
    if device.type == "cuda": torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        loss, acc = evaluation.epoch("train", loader, net, opt, crit, aug=False)
    if device.type == "cuda": torch.cuda.synchronize()
    return (time.perf_counter() - t) / n, loss

tg, lg = epochs(dev, 30)
tc, lc = epochs(torch.device("cpu"), 5)
print(f"evaluation epoch (100 images, augment + ConvNet step): GPU {tg * 1e3:.1f} ms, host ({torch.get_num_threads()} threads) {tc * 1e3:.1f} ms"
      f" -> 1001 epochs x 5 nets per evaluation: {tg * 5005:.0f} s vs {tc * 5005:.0f} s")
