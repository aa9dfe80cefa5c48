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


def reduce_loss(loss: torch.Tensor) -> torch.Tensor:
    """Sum of the per-rank matching losses, in place; a no-op for a single process."""
    if is_distributed():
        dist.all_reduce(loss, op=dist.ReduceOp.SUM)
    return loss


def max_over_ranks(seconds: float, device) -> float:
    """The slowest rank's wall time (the benchmark contract)."""
    if not is_distributed():
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
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
    dist.all_reduce(out, op=dist.ReduceOp.SUM)
    return out
