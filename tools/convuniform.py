"""k_conv128_rows on a perfectly balanced launch: 768 tiles of 128 rows, every row with the same 9 offsets present (random neighbours), so
each CU runs 3 workgroups x 36 steps and the time per CU-step can be read off the launch time without instrumentation.
usage: python tools/convuniform.py [tiles_per_cu ...]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dd3d_amd import _lib as L
dev = torch.device("cuda:0")
for per_cu in [int(a) for a in sys.argv[1:]] or [1, 2, 3, 6]:
    M = 256 * per_cu * 128
    g = torch.Generator(device="cpu").manual_seed(0)
    nbr = torch.full((27, M), -1, dtype=torch.int32)
    for k in (1, 4, 7, 10, 13, 16, 19, 22, 25):
        nbr[k] = torch.randint(0, M, (M,), generator=g, dtype=torch.int32)
    nbr = nbr.to(dev)
    X = torch.randn(M, 128, device=dev); W = torch.randn(27, 128, 128, device=dev) * 0.05; Y = torch.empty(M, 128, device=dev)
    nblk = int(L.lib.dd3d_conv_schedule_blocks(M))
    perm = torch.arange(M, dtype=torch.int32, device=dev)
    nbr_s = nbr.clone(); tmask = torch.full((M // 128,), sum(1 << k for k in (1, 4, 7, 10, 13, 16, 19, 22, 25)), dtype=torch.int32, device=dev)
    tmap = torch.full((nblk,), -1, dtype=torch.int32, device=dev); tmap[:M // 128] = torch.arange(M // 128, dtype=torch.int32, device=dev)
    def run(): L.call("dd3d_conv128_ordered", L.ptr(X), M, L.ptr(W), L.ptr(nbr), M, 27, M, None, 0, L.ptr(perm), L.ptr(tmap), L.ptr(nbr_s), L.ptr(tmask), L.ptr(Y), L.stream())
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    print(f"{per_cu} tiles/CU: {us:8.1f} us  {us / (36 * per_cu):.3f} us per CU-step  {2 * 9 * M * 128 * 128 / us / 1e6:6.1f} TFLOP/s")
