"""Class sharding of the matching step over the GPUs of one node.

Classes are independent inside a step (mnist_distill.py:335), so class c belongs to rank ``c mod world``; each rank keeps its
classes' synthetic clouds, gradients and SGD momentum.  The only exchange is the all-reduce of the scalar matching loss
(mnist_distill.py:361,373) -- RCCL over xGMI on the GPUs (backend "nccl"), gloo in the CPU tests."""
from __future__ import annotations

from typing import List

import torch
import torch.distributed as dist


def owned_classes(num_classes: int, rank: int, world: int) -> List[int]:
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    return [c for c in range(num_classes) if c % world == rank]


def is_distributed() -> bool:
    return dist.is_available() and dist.is_initialized()


def _all_reduce(t: torch.Tensor, op, group=None):
    """all_reduce in place.  RCCL reduces device tensors directly; under gloo (the CPU tests, and the two-ranks-on-one-GPU test) a device
    tensor is staged through the host."""
    if t.is_cuda and dist.get_backend(group) == "gloo":
        h = t.detach().cpu()
        dist.all_reduce(h, op=op, group=group)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=op, group=group)
    return t


def reduce_loss(loss: torch.Tensor) -> torch.Tensor:
    """Sum of the per-rank matching losses, in place; a no-op for a single process."""
    if is_distributed():
        _all_reduce(loss, dist.ReduceOp.SUM)
    return loss


def max_over_ranks(seconds: float, device) -> float:
    """The slowest rank's wall time (the benchmark contract)."""
    if not is_distributed():
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    _all_reduce(t, dist.ReduceOp.MAX)
    return float(t.item())


def gather_synthetic(image_syn: torch.Tensor, ipc: int, num_classes: int) -> torch.Tensor:
    """Assemble the full synthetic set from the rows each rank owns (checkpoint / evaluation time only: every rank contributes
    its classes' rows, the rest are zeroed, and one SUM all-reduce leaves the complete tensor everywhere)."""
    if not is_distributed():
        return image_syn
    rank, world = dist.get_rank(), dist.get_world_size()
    out = torch.zeros_like(image_syn)
    for c in owned_classes(num_classes, rank, world):
        out[c * ipc:(c + 1) * ipc] = image_syn[c * ipc:(c + 1) * ipc]
    _all_reduce(out, dist.ReduceOp.SUM)
    return out


# ---------------------------------------------------------------------------------------------------------------------
# Second level: the real batch of a class split over ranks
# ---------------------------------------------------------------------------------------------------------------------
def split_layout(num_classes: int, rank: int, world: int, real_parts: int = 0):
    """world = class_shards x real_parts.  Rank r = (shard r // P, part r % P): it computes the real-batch gradients of ALL classes of its
    shard (class c belongs to shard ``c mod class_shards``) over part ``r % P`` of every sampled batch, and owns the synthetic clouds,
    the matching loss and the update of every P-th class of the shard.  ``real_parts = 0``: the smallest P for which the classes divide
    evenly over the shards (10 classes: 8 ranks = 2 shards x 4 parts, 4 ranks = 2 x 2, 2 ranks = 2 x 1 -- every rank then holds the
    same number of real clouds, where class sharding alone leaves 2 classes on two of 8 ranks and 1 on the others).
    Returns dict(classes, real_classes, real_part, group_ranks); the FusedDistiller arguments of the same names."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    if real_parts <= 0:
        shards = max(d for d in range(1, world + 1) if world % d == 0 and num_classes % d == 0)
        real_parts = world // shards
    if world % real_parts:
        raise ValueError("real_parts must divide the world size")
    shards = world // real_parts
    cs, rp = rank // real_parts, rank % real_parts
    mine = [c for c in range(num_classes) if c % shards == cs]
    owned = [c for j, c in enumerate(mine) if j % real_parts == rp]
    if not owned:
        raise ValueError(f"{len(mine)} classes per shard cannot give each of {real_parts} ranks a synthetic class to own")
    return {"classes": owned, "real_classes": mine, "real_part": (rp, real_parts), "group_ranks": [cs * real_parts + i for i in range(real_parts)]}


def all_reduce_in(group):
    """exchange(flat) for FusedDistiller: sum the flat buffer of partial real-batch gradients over the ranks of ``group`` (the ranks that
    share a class shard), in place -- RCCL over xGMI on the GPUs, gloo in the tests.  Called outside graph capture."""
    def exchange(flat: torch.Tensor):
        _all_reduce(flat, dist.ReduceOp.SUM, group)
    return exchange


# ---------------------------------------------------------------------------------------------------------------------
# Inner-loop network update: gradients of the network parameters summed over the ranks
# ---------------------------------------------------------------------------------------------------------------------
def all_reduce_gradients(params, group=None):
    """``p.grad`` of every parameter summed over the ranks, in place: ONE flat fp32 buffer (MEConvExp: 1 456 778 floats = 5.83 MB), ONE
    all-reduce -- over seven 153 GB/s xGMI links a message of this size is latency-bound, so nothing is gained by bucketing it.  RCCL on the
    GPUs, gloo in the CPU tests; a no-op for a single process.  Parameters without a gradient contribute zeros."""
    params = [p for p in params]
    if not is_distributed() or not params:
        return
    flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in params])
    _all_reduce(flat, dist.ReduceOp.SUM, group)
    off = 0
    for p in params:
        n = p.numel()
        g = flat[off:off + n].view_as(p)
        if p.grad is None:
            p.grad = g.clone()
        else:
            p.grad.copy_(g)
        off += n
