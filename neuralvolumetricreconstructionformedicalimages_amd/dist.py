"""Data parallelism for the NAF hot path: one process per GPU, rays sharded across ranks, gradients summed with one
RCCL all-reduce per parameter buffer (torch.distributed backend "nccl" is RCCL on ROCm; "gloo" for CPU tests).

The reference has no distributed code at all.  The path shards naturally: rays are independent, the
model (57 MB fp32 table at T=2^19 + 17 KB MLP) is replicated, and the only exchange is the gradient sum before the
optimiser.  The loss is defined as the GLOBAL masked mean: each rank weights its rays by
mask / (global number of masked rays), so summing the per-rank gradients gives exactly the single-process gradient of
the concatenated batch.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def init_from_env(device_type="cuda"):
    """Initialise the default process group from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun contract).
    Returns (rank, world_size, local_rank, group-or-None)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world == 1:
        return 0, 1, local_rank, None
    if not dist.is_initialized():
        if device_type == "cuda":
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")
    return rank, world, local_rank, dist.group.WORLD


def shard_range(n_items, rank, world):
    """Contiguous [begin, end) slice of `n_items` rays for `rank` (sizes differ by at most one)."""
    base, rem = divmod(n_items, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def global_mean_weights(mask, group=None):
    """w_r = mask_r / (number of masked rays on ALL ranks): the data-parallel form of the masked MSE (loss.py:37)."""
    m = mask.float()
    total = m.sum().reshape(1)
    if group is not None:
        dist.all_reduce(total, group=group)
    return m / total.clamp(min=1.0)


def all_reduce_sum_(tensors, group=None):
    """In-place sum of each tensor over the group (no-op without a group)."""
    if group is None:
        return
    for t in tensors:
        dist.all_reduce(t, group=group)


def broadcast_parameters(tensors, group=None, src=0):
    """Make every rank start from rank `src`'s parameters."""
    if group is None:
        return
    for t in tensors:
        dist.broadcast(t, src=src, group=group)
