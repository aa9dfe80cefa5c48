"""Feasibility probe: can torch.cuda.graph capture our ctypes-launched kernels on two forked streams, and what does replay cost?"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dd3d_amd import _lib as L

dev = torch.device("cuda:0")
n = 1 << 14
x = torch.randn(n, device=dev)
m = torch.zeros(n, device=dev)
g = torch.randn(n, device=dev)
s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)

def body(k):
    main = torch.cuda.current_stream()
    s1.wait_stream(main); s2.wait_stream(main)
    for s in (s1, s2):
        with torch.cuda.stream(s):
            for _ in range(k):
                t = torch.empty(n, device=dev)
                L.call("dd3d_sgd_momentum_clamp", L.ptr(x), L.ptr(g), L.ptr(m), n, 0.0, 0.5, 0, 0, 0.0, 27.0, L.stream())
    main.wait_stream(s1); main.wait_stream(s2)

side = torch.cuda.Stream(dev)
with torch.cuda.stream(side):
    body(4)
torch.cuda.synchronize()
for k in (50, 200):
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        body(k)
    torch.cuda.synchronize()
    for _ in range(3):
        gr.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        gr.replay()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"graph with 2x{k} kernel nodes: host {(t1-t0)/20*1e6:.1f} us/replay, total {(t2-t0)/20*1e6:.1f} us/replay = {(t2-t0)/20*1e6/(2*k):.2f} us/node")
    t0 = time.perf_counter()
    for _ in range(20):
        body(k)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"eager 2x{k}: {(t2-t0)/20*1e6:.1f} us = {(t2-t0)/20*1e6/(2*k):.2f} us/launch")
