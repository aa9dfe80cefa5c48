"""Two processes on the one GPU of the box, gloo between them (RCCL needs a GPU per rank): the second sharding level end to end --
sharding.split_layout, FusedDistiller(real_classes, real_part, exchange=all_reduce_in(group)) with eager launches, gather_synthetic --
against the single-process step on the same samples."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    try:
        _work(rank, world, port, q)
    except Exception:   # a rank that dies silently leaves the other one waiting in a collective until the timeout
        import traceback
        q.put((rank, "error", traceback.format_exc()))


def _work(rank, world, port, q):
    import datetime
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from detgen import det_state_dict
    from dd3d_amd.fused_step import FusedDistiller
    from dd3d_amd.models.MEConv import MEConvExp
    from dd3d_amd.sharding import all_reduce_in, gather_synthetic, reduce_loss, split_layout
    from dd3d_amd.synthetic import init_synthetic, mnist3d_like
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    net = MEConvExp(in_channel=3, out_channel=10)
    net.load_state_dict(det_state_dict(net, "net.MEConvExp"))
    images, _, idx = mnist3d_like(4, 30, seed=5)
    syn = init_synthetic(4, 3, 250, seed=2)
    lay = split_layout(4, rank, world, 2)                     # one shard of four classes, two parts: rank 0 owns classes 0, 2; rank 1 owns 1, 3
    group = dist.new_group(lay["group_ranks"])
    d = FusedDistiller(images, idx, net, ipc=3, num_classes=4, batch_size=24, device=dev, seed=11, image_syn=syn, dist_opt="ours", graph=False,
                       classes=lay["classes"], real_classes=lay["real_classes"], real_part=lay["real_part"], exchange=all_reduce_in(group))
    losses = []
    log = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", f"split_rank{rank}.log"), "w")
    for it in range(4):
        print("step", it, file=log, flush=True)
        losses.append(float(reduce_loss(d.outer_step().clone()).item()))
    print("steps done", file=log, flush=True)
    d.check()
    full = gather_synthetic(d.image_syn, 3, 4)
    ref = None
    if rank == 0:                                             # the unsplit step, same seed = same permutations
        w = FusedDistiller(images, idx, net, ipc=3, num_classes=4, batch_size=24, device=dev, seed=11, image_syn=syn, dist_opt="ours", graph=False)
        ref = ([float(w.outer_step().item()) for _ in range(4)], w.image_syn.cpu())
    q.put((rank, lay["classes"], losses, full.cpu(), ref, d._graph is None))
    dist.destroy_process_group()


def test_split_real_batch_two_ranks_on_one_gpu():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=110) for _ in range(world)), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=30)
    for r in res:
        assert r[1] != "error", r[2]
    assert all(p.exitcode == 0 for p in procs)
    assert res[0][1] == [0, 2] and res[1][1] == [1, 3] and res[0][5] and res[1][5]
    ref_losses, ref_syn = res[0][4]
    for r in res:
        assert np.allclose(r[2], ref_losses, rtol=2e-5), (r[2], ref_losses)      # every rank sees the summed loss of the whole step
        assert float((r[3] - ref_syn).abs().max() / ref_syn.abs().max()) < 1e-6   # and the gathered synthetic set is the unsplit one
    assert torch.equal(res[0][3], res[1][3])
