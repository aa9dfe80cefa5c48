"""world_size-2 gloo test (CPU) of the class-sharding helpers used by bench.py: ownership is a partition, the loss all-reduce sums
the per-rank partial losses, the timing reduction takes the slowest rank, the synthetic set can be re-assembled."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dd3d_amd.sharding import gather_synthetic, max_over_ranks, owned_classes, reduce_loss
    num_classes, ipc = 10, 2
    mine = owned_classes(num_classes, rank, world)
    per_class = torch.arange(1, num_classes + 1, dtype=torch.float32) ** 2      # stand-in for the per-class matching losses
    loss = reduce_loss(per_class[mine].sum().clone())
    t = max_over_ranks(1.0 + rank, "cpu")
    syn = torch.zeros(num_classes * ipc, 3, 3)
    for c in mine:
        syn[c * ipc:(c + 1) * ipc] = c + 1        # only the owner holds fresh rows
    full = gather_synthetic(syn, ipc, num_classes)
    q.put((rank, mine, float(loss), t, full[:, 0, 0].tolist()))
    dist.destroy_process_group()


def test_class_sharding_world2_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    owned = [set(r[1]) for r in res]
    assert owned[0] | owned[1] == set(range(10)) and not (owned[0] & owned[1])
    total = float((torch.arange(1, 11, dtype=torch.float32) ** 2).sum())
    assert all(r[2] == pytest.approx(total) for r in res)          # every rank sees the summed loss
    assert all(r[3] == 2.0 for r in res)                             # max over ranks of (1.0, 2.0)
    expect = [float(c + 1) for c in range(10) for _ in range(2)]
    assert all(r[4] == expect for r in res)


def test_owned_classes_partition_any_world():
    from dd3d_amd.sharding import owned_classes
    for world in (1, 2, 4, 8, 16):
        seen = sorted(c for r in range(world) for c in owned_classes(10, r, world))
        assert seen == list(range(10))
    with pytest.raises(ValueError):
        owned_classes(10, 3, 2)
