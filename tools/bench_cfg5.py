"""BASELINE configs[4] (cfg5) at FULL size on one GPU: 55 classes x 256 real clouds x 2048 points (ShapeNet-shaped ellipsoid shells in
[-1,1]^3), voxel size 0.02, ipc 10, MEConvExp(num_classes=55, dense_extent=26^3) -- a 2.25 M-column classifier whose per-group gradient
list alone is 54 GB for all 110 groups.  The classes are independent in the matching step (mnist_distill.py:333-369: the loop over c), so
the step is run over CHUNKS of classes, one FusedDistiller per chunk (the class-sharding mechanism of dd3d_amd/sharding.py, here in
sequence on one device), all chunks updating the same synthetic-point tensor; eager launches (the chunks share the allocator's memory).
usage: python tools/bench_cfg5.py [--chunk 11] [--steps 3] [--warmup 1] [--classes 55]   -> one JSON line"""
import argparse, json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dd3d_amd.fused_step import FusedDistiller
from dd3d_amd.models.MEConv import MEConvExp
from dd3d_amd.synthetic import shell_clouds

ap = argparse.ArgumentParser()
ap.add_argument("--chunk", type=int, default=11)
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--warmup", type=int, default=1)
ap.add_argument("--classes", type=int, default=55)
ap.add_argument("--metric-name", default="modmse")
args = ap.parse_args()
dev = torch.device("cuda:0")
torch.manual_seed(0)
n_classes, n_points, ipc, per, pool = args.classes, 2048, 10, 256, 320
net = MEConvExp(in_channel=3, out_channel=n_classes, num_classes=n_classes, dense_extent=(26, 26, 26)).to(dev).train()
g = torch.Generator().manual_seed(0)
syn = ((torch.rand((n_classes * ipc, n_points, 3), generator=g) * 1.8 - 0.9).float()).to(dev)
clouds = shell_clouds(n_classes * pool, n_points, seed=0)
idx = [list(range(c * pool, (c + 1) * pool)) for c in range(n_classes)]
chunks = [list(range(a, min(a + args.chunk, n_classes))) for a in range(0, n_classes, args.chunk)]
ds = []
for ch in chunks:
    d = FusedDistiller(None, idx, net, ipc=ipc, num_classes=n_classes, num_points=n_points, batch_size=per, dist_opt=args.metric_name, device=dev,
                       seed=1234, image_syn=syn, classes=ch, real_clouds=clouds, quantization_size=0.02, clamp_range=(-0.95, 0.95),
                       zero_first_axis=False, dist_lr=1e-6, graph=False)
    d.image_syn = syn   # every chunk updates its own rows of the SAME tensor
    ds.append(d)
work = {}

def step():
    total = torch.zeros((), device=dev)
    for d in ds:
        total += d.outer_step()
        rc = d.row_counts()[0]
        for k, v in zip(("M0", "M1", "M2"), rc):
            work[k] = work.get(k, 0) + v
        d.last = {}       # the chunk's gradient lists go back to the allocator before the next chunk runs
    return total

for _ in range(args.warmup):
    step()
torch.cuda.synchronize()
work.clear()
t0 = time.perf_counter()
for _ in range(args.steps):
    loss = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / args.steps
print(json.dumps({"metric": "matching_steps_per_sec", "value": 1.0 / dt, "unit": "steps/s", "ms_per_step": dt * 1e3, "n_gpus": 1, "steps": args.steps,
                  "warmup": args.warmup, "dtype": "f32", "data": "synthetic",
                  "config": {"workload": f"cfg5: ShapeNet-shaped ellipsoid shells, {n_classes} classes x {per} real clouds x {n_points} pts, voxel 0.02, ipc={ipc}, "
                                         f"MEConvExp(num_classes={n_classes}, dense_extent=26^3), dist_opt={args.metric_name}",
                             "path": f"class-batched step, eager launches, {len(chunks)} chunks of <= {args.chunk} classes in sequence"},
                  "loss": float(loss.item()), "rows_per_step": {k: v // args.steps for k, v in work.items()},
                  "peak_hbm_GB": torch.cuda.max_memory_allocated() / 1e9}))
