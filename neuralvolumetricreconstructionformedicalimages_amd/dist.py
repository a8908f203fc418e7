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


def local_device_index():
    """GPU of this rank: LOCAL_RANK (one process per GPU).  NAF_DIST_SHARE_GPU=1 is the rehearsal hook of a one-GPU box:
    every rank then uses device 0 (together with NAF_DIST_BACKEND=gloo -- RCCL refuses two ranks on one device)."""
    if os.environ.get("NAF_DIST_SHARE_GPU") == "1":
        return 0
    return int(os.environ.get("LOCAL_RANK", "0"))


def init_from_env(device_type="cuda"):
    """Initialise the default process group from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun contract).
    Returns (rank, world_size, local_rank, group-or-None).  Backend: RCCL ("nccl") on GPUs, gloo on the CPU;
    NAF_DIST_BACKEND overrides it (rehearsals on a one-GPU box)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world == 1:
        return 0, 1, local_rank, None
    if not dist.is_initialized():
        backend = os.environ.get("NAF_DIST_BACKEND", "nccl" if device_type == "cuda" else "gloo")
        if device_type == "cuda":
            torch.cuda.set_device(local_device_index())
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_device_index()))
        else:
            dist.init_process_group(backend)
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


def default_bucket_levels(num_levels):
    """Level ranges in the order the gradient scatter finishes them: the fine half, the next quarter, the coarse quarter.
    Every reducer launch still runs in full rounds of 256 workgroups (64 row buckets x 8 levels = 512; the two launches of
    four levels split each bucket's tiles four ways, 4 x 256, and finish their rows with one atomic each),
    each bucket's all-reduce hides behind the reduction of the buckets after it, and the exchange left exposed at the end is
    the coarse quarter, whose dense levels are small (6.7 of 57 MB at L = 16, T = 2^19; 33.6 and 16.8 MB before it).
    Measured on one GPU (bench.py --force-dp --buckets ...): 9.08-9.18 ms per step against 9.13-9.15 ms for two halves -- the
    reducer itself is faster when the coarse levels, whose runs are uneven, do not share a launch with four hashed levels
    (2.05 against 2.14 ms) -- with a last bucket a third the size."""
    L = int(num_levels)
    if L >= 4:
        return [(L // 2, L), (L // 4, L // 2), (0, L // 4)]
    return [(L // 2, L), (0, L // 2)] if L >= 2 else [(0, L)]


def grad_bucket_slices(offsets, level_dim, bucket_levels):
    """Element ranges [begin, end) of the flat table-gradient buffer for each bucket of levels (`offsets` = row offsets
    per level, [L+1]; a level's rows are contiguous, hashgrid.py:92-102).  Raises unless the buckets are disjoint level
    ranges that cover every level exactly once."""
    offs = [int(v) for v in offsets]
    L = len(offs) - 1
    buckets = [(int(a), int(b)) for a, b in bucket_levels]
    covered = sorted(l for a, b in buckets for l in range(a, b))
    if covered != list(range(L)) or any(a >= b for a, b in buckets):
        raise ValueError(f"bucket_levels {buckets} must be disjoint, non-empty level ranges covering 0..{L}")
    return [(offs[a] * level_dim, offs[b] * level_dim) for a, b in buckets]


def aligned_update_slices(slices, multiple=4):
    """Element ranges for the per-bucket optimiser pass, given the exchange slices in the order their sums arrive.

    The Adam kernel works on 16-byte groups, so a boundary between two buckets that is not a multiple of `multiple`
    elements is moved to the next / previous multiple in favour of the bucket whose sum arrives LATER: collectives complete
    in issue order, so when the later bucket is ready the ragged elements of its neighbour are too.  The ranges still tile
    the same elements exactly once."""
    order = {rng: i for i, rng in enumerate(slices)}
    tiled = sorted(slices)
    for (a0, b0), (a1, b1) in zip(tiled, tiled[1:]):
        if b0 != a1:
            raise ValueError("exchange slices must be contiguous")
    bounds = [tiled[0][0]]
    for left, right in zip(tiled, tiled[1:]):
        x = left[1]
        if x % multiple:
            x = x + (-x) % multiple if order[left] > order[right] else x - x % multiple
        bounds.append(x)
    bounds.append(tiled[-1][1])
    moved = {rng: (bounds[i], bounds[i + 1]) for i, rng in enumerate(tiled)}
    return [moved[rng] for rng in slices]


def all_reduce_buckets_(flat, slices, group=None):
    """Sum each [begin, end) slice of the flat buffer over the group, one collective per slice, in list order (every rank
    must use the same list).  The engine issues exactly these collectives, each as soon as its bucket's event has fired."""
    if group is None:
        return
    for a, b in slices:
        dist.all_reduce(flat[a:b], group=group)


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
