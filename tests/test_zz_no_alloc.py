"""dd3d_arp_inorm_bwd_static -- the one entry point that allocates nothing (its per-cloud partial sums live in a file-scope device
array) -- against dd3d_arp_inorm_bwd with a caller workspace: same kernels, so every output must be bit-identical; two calls in a
row on the same GPU, the second with a smaller input (the array keeps the first call's contents); sizes beyond the static maximum
are rejected.  Through the C ABI (ctypes), like every other GPU test."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def _case(B, G, max_rows, seed):
    g = torch.Generator().manual_seed(seed)
    n_in = torch.randint(0, max_rows + 1, (B,), generator=g)
    n_in[B // 2] = 0                                       # an empty cloud
    n_out = torch.clamp((n_in + 2) // 3, min=0)
    seg_in = torch.cat([torch.zeros(1, dtype=torch.long), torch.cumsum(n_in, 0)])
    seg_out = torch.cat([torch.zeros(1, dtype=torch.long), torch.cumsum(n_out, 0)])
    M_in, M_out = int(seg_in[-1]), int(seg_out[-1])
    cloud = torch.repeat_interleave(torch.arange(B), n_in)
    local = torch.arange(M_in) - seg_in[cloud]
    parent = (seg_out[cloud] + local // 3).to(torch.int32)  # every input row has one parent inside its cloud's pooled rows
    # arg[o][c] = one of the (up to 3) children of pooled row o, per channel
    ocloud = torch.repeat_interleave(torch.arange(B), n_out)
    olocal = torch.arange(M_out) - seg_out[ocloud]
    nkid = torch.minimum(torch.full_like(olocal, 3), n_in[ocloud] - olocal * 3)
    pick = (torch.rand(M_out, 128, generator=g) * nkid[:, None]).long()
    arg = (seg_in[ocloud][:, None] + olocal[:, None] * 3 + pick).to(torch.int32)
    gco = torch.linspace(0, B, G + 1).long()
    cg = (torch.bucketize(torch.arange(B), gco[1:], right=True)).to(torch.int32)
    t = dict(g_pool=torch.randn(M_out, 128, generator=g), arg=arg, parent=parent, xhat=torch.randn(M_in, 128, generator=g),
             inv_std=torch.rand(B, 128, generator=g) + 0.5, w=1 + 0.2 * torch.randn(128, generator=g), b=0.2 * torch.randn(128, generator=g),
             seg=seg_in.to(torch.int32), cg=cg, gco=gco.to(torch.int32), ga_prev=torch.randn(M_in, 128, generator=g),
             gw_extra=torch.randn(G, 128, generator=g))
    return {k: v.to(DEV).contiguous() for k, v in t.items()}, M_in


def _run(static, t, M_in, B, G, second_order):
    from dd3d_amd import _lib as L
    dY = torch.full((M_in, 128), 7.0, device=DEV)
    ga = torch.full((M_in, 128), 7.0, device=DEV)
    gw_w, gw_b = torch.full((G, 128), 7.0, device=DEV), torch.full((G, 128), 7.0, device=DEV)
    head = [L.ptr(t["g_pool"]), L.ptr(t["arg"]), L.ptr(t["parent"]), L.ptr(t["xhat"]), L.ptr(t["inv_std"]), L.ptr(t["w"]), L.ptr(t["b"]), 128,
            L.ptr(t["seg"]), B, L.ptr(t["cg"]), L.ptr(t["gco"]), G, L.ptr(dY), L.ptr(ga), B // 2,
            L.ptr(t["ga_prev"]) if second_order else None, L.ptr(t["gw_extra"]) if second_order else None, L.ptr(gw_w), L.ptr(gw_b)]
    if static:
        L.call("dd3d_arp_inorm_bwd_static", *head, L.stream())
    else:
        ws = torch.empty(int(L.lib.dd3d_arp_inorm_bwd_ws_bytes(B)), dtype=torch.uint8, device=DEV)
        L.call("dd3d_arp_inorm_bwd", *head, L.ptr(ws), ws.numel(), L.stream())
    torch.cuda.synchronize()
    return dY, ga, gw_w, gw_b


def test_no_alloc_entry_point_equals_workspace_entry_point():
    from dd3d_amd import _lib as L
    # two calls in a row on one GPU, the second smaller: the static array still holds the first call's partial sums
    for (B, G, rows, seed, so) in ((300, 6, 120, 1, False), (17, 2, 40, 2, True)):
        t, M_in = _case(B, G, rows, seed)
        want = _run(False, t, M_in, B, G, so)
        got = _run(True, t, M_in, B, G, so)
        for a, b in zip(got, want):
            assert torch.equal(a, b)
        assert bool(torch.isfinite(got[0]).all()) and float(got[2].abs().max()) > 0
    # beyond the static maximum: a clear error, nothing launched
    mx = int(L.lib.dd3d_arp_inorm_bwd_static_max_clouds())
    assert mx == 8192
    rc = L.lib.dd3d_arp_inorm_bwd_static(*([None] * 7), 128, None, mx + 1, None, None, 1, None, None, 0, None, None, None, None, None)
    assert rc == -2


_CAPTURE_CHILD = r"""
import sys, torch
sys.path.insert(0, {root!r})
from tests.test_zz_no_alloc import _case, _run, DEV
from dd3d_amd import _lib as L
B, G = 40, 3
t, M_in = _case(B, G, 60, 5)
w1 = torch.ones(1, device=DEV)
# set-up call OUTSIDE capture: B = 0 launches nothing and takes the array's address on this device (include/dd3d.h)
assert L.lib.dd3d_arp_inorm_bwd_static(None, None, None, None, None, L.ptr(w1), L.ptr(w1), 128, None, 0, None, None, 1, None, None, 0, None, None, None, None, L.stream()) == 0
dY, ga = torch.full((M_in, 128), 7.0, device=DEV), torch.full((M_in, 128), 7.0, device=DEV)
gw_w, gw_b = torch.full((G, 128), 7.0, device=DEV), torch.full((G, 128), 7.0, device=DEV)
free0 = torch.cuda.mem_get_info()[0]
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    L.call("dd3d_arp_inorm_bwd_static", L.ptr(t["g_pool"]), L.ptr(t["arg"]), L.ptr(t["parent"]), L.ptr(t["xhat"]), L.ptr(t["inv_std"]), L.ptr(t["w"]), L.ptr(t["b"]),
           128, L.ptr(t["seg"]), B, L.ptr(t["cg"]), L.ptr(t["gco"]), G, L.ptr(dY), L.ptr(ga), B // 2, None, None, L.ptr(gw_w), L.ptr(gw_b), L.stream())
g.replay()
torch.cuda.synchronize()
want = _run(False, t, M_in, B, G, False)
for a, b in zip((dY, ga, gw_w, gw_b), want):
    assert torch.equal(a, b)
print("captured-ok")
"""


def test_no_alloc_entry_point_under_graph_capture_after_a_set_up_call():
    """The entry point is meant for callers under HIP-graph capture: in a FRESH process, one set-up call outside capture (B = 0: nothing is
    launched, the device address of the static array is taken), then the real call captured into a graph and replayed; outputs
    bit-identical to the workspace entry point."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", _CAPTURE_CHILD.format(root=root)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "captured-ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
