"""Micro-benchmark of the 128-channel sparse convolution on the cfg2 real batch (10 x 256 stroke digits) and the synthetic batch.
usage: python tools_convbench.py   (set DD3D_CONV128_DENSE=1 to time the dense-tile kernel instead)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dd3d_amd import _lib as L
from dd3d_amd.sparse import CoordinateManager
from dd3d_amd.synthetic import mnist3d_like, init_synthetic
from dd3d_amd.utils.Mnist2D import batch_from_images

dev = torch.device("cuda:0")
images, labels, _ = mnist3d_like(10, 256, seed=0)
b = batch_from_images(images.to(dev), labels.to(dev), 250)
syn = init_synthetic(10, 10, 250, seed=0).to(dev).clamp_(0, 27)
syn[:, :, 0] = 0
sc = torch.cat([torch.arange(100, device=dev, dtype=torch.float32).repeat_interleave(250)[:, None], syn.reshape(-1, 3)], 1)
rc = b["coordinates"].clone()
rc[:, 0] += 100
for name, coords, B in (("real", b["coordinates"], 2560), ("syn", sc, 100), ("merged", torch.cat([sc, rc]), 2660)):
    cm = CoordinateManager(coords, B)
    nbr, ld = cm.kernel_map(2, 2, 3)
    M = cm.num_rows(2)
    pairs = int((nbr[:, :M] >= 0).sum())
    X = torch.randn(M, 128, device=dev)
    assert int(nbr[:, :M].max()) < X.shape[0] and ld >= M  # every gathered row exists
    W = torch.randn(27, 128, 128, device=dev) * 0.05
    Y = torch.empty(M, 128, device=dev)
    def run():
        L.call("dd3d_conv_fwd", L.ptr(X), M, 128, L.ptr(W), 128, L.ptr(nbr), ld, 27, M, None, 0, L.ptr(Y), L.stream())
    for _ in range(5): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 50
    e0.record()
    for _ in range(n): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    # scheduled variant (dd3d_conv_schedule: pattern-sorted rows + balanced tile placement); must be bit-identical
    nblk = int(L.lib.dd3d_conv_schedule_blocks(M))
    perm = torch.empty(M, dtype=torch.int32, device=dev)
    nbr_s = torch.empty_like(nbr); tmask = torch.empty((M + 31) // 32, dtype=torch.int32, device=dev)
    tmap = torch.empty(nblk, dtype=torch.int32, device=dev)
    ws = torch.empty(int(L.lib.dd3d_conv_schedule_ws_bytes(M)), dtype=torch.uint8, device=dev)
    Y2 = torch.empty_like(Y)
    def sched():
        L.call("dd3d_conv_schedule", L.ptr(nbr), ld, 27, M, None, L.ptr(perm), L.ptr(tmap), L.ptr(nbr_s), L.ptr(tmask), L.ptr(ws), ws.numel(), L.stream())
    def run2():
        L.call("dd3d_conv128_ordered", L.ptr(X), M, L.ptr(W), L.ptr(nbr), ld, 27, M, None, 0, L.ptr(perm), L.ptr(tmap), L.ptr(nbr_s), L.ptr(tmask), L.ptr(Y2), L.stream())
    sched(); run2(); torch.cuda.synchronize()
    assert sorted(perm.tolist()) == list(range(M)), "perm is not a permutation"
    tl = [t for t in tmap.tolist() if t >= 0]
    TR = int(L.lib.dd3d_conv_schedule_tile_rows(M))
    assert sorted(tl) == list(range((M + TR - 1) // TR)), "tile_map does not cover every tile once"
    assert torch.equal(Y, Y2), "scheduled conv differs"
    present = nbr[:, :M] >= 0
    pm = present[:, perm.long()]
    pm = torch.cat([pm, pm.new_zeros(27, (-M) % TR)], 1).reshape(27, -1, TR).any(2).sum(0).cpu() * 4
    loads = torch.zeros(256)
    for bi, t in enumerate(tmap.tolist()):
        if t >= 0: loads[bi % 256] += pm[t]
    print(f"   schedule: tile-steps {int(pm.sum())} (cloud order {int(torch.cat([present, present.new_zeros(27, (-M) % TR)], 1).reshape(27, -1, TR).any(2).sum()) * 4}), most / least loaded CU {int(loads.max())} / {int(loads.min())}")
    for f, tag in ((sched, "dd3d_conv_schedule"), (run2, "scheduled conv")):
        e0.record()
        for _ in range(n): f()
        e1.record(); torch.cuda.synchronize()
        us2 = e0.elapsed_time(e1) / n * 1e3
        print(f"   {tag}: {us2:8.1f} us" + (f"  {2*pairs*128*128/us2/1e6:6.1f} TFLOP/s (alg)" if f is run2 else ""))
    act = int((nbr[:, :M] >= 0).any(1).sum())
    print(f"{name}: M={M} pairs={pairs} active offsets={act} density={pairs/(M*act):.2f}  {us:8.1f} us  {2*pairs*128*128/us/1e6:6.1f} TFLOP/s (alg)")
