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
        # The training group keeps the backend's default timeout (NAF_DIST_TIMEOUT_MIN overrides it): a dead rank or mismatched
        # collectives in a training step surface in minutes.  What must outlast an evaluation is only the wait of the other ranks while
        # rank 0 evaluates alone (a 1024^2 projection + a 512^3 .. 1024^3 volume query): `wait_for_rank0` below, on a group of its own.
        import datetime
        kw = {}
        if "NAF_DIST_TIMEOUT_MIN" in os.environ:
            kw["timeout"] = datetime.timedelta(minutes=float(os.environ["NAF_DIST_TIMEOUT_MIN"]))
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_device_index()), **kw)
        else:
            dist.init_process_group(backend, **kw)
    return rank, world, local_rank, dist.group.WORLD


_EVAL_GROUP = None


def wait_for_rank0(group=None):
    """Barrier for the ranks that wait while rank 0 evaluates or checkpoints alone: a gloo group of its own with a long timeout
    (NAF_DIST_EVAL_TIMEOUT_MIN, default 2 h), created at the first call by every rank -- the waiters block on the host instead of
    spinning in a device collective, and the training group's collectives keep their short timeout."""
    global _EVAL_GROUP
    if group is None or dist.get_world_size(group) == 1:
        return
    if _EVAL_GROUP is None:
        import datetime
        minutes = float(os.environ.get("NAF_DIST_EVAL_TIMEOUT_MIN", "120"))
        _EVAL_GROUP = dist.new_group(ranks=list(range(dist.get_world_size())), backend="gloo", timeout=datetime.timedelta(minutes=minutes))
    dist.barrier(group=_EVAL_GROUP)


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


# Below this many sample points per rank and step (the reference's 1 024 rays x 192 samples are 0.2 M) the scatter of half the
# levels lasts ~0.06 ms -- less than the launch latency of the collective it would hide -- so the table is exchanged as ONE range:
# three collectives per step (MLP all-reduce, reduce-scatter, all-gather) instead of five, one reducer launch instead of two.
SINGLE_BUCKET_BELOW_POINTS = 1 << 20


def pick_dp_mode(world, num_levels, level_dim, table_elements, points_per_step, feature_bytes=2, table_bytes=2):
    """'levels' or 'sharded' for a multi-GPU run, by the bytes a rank puts on the links per step (x (N-1)/N either way):
    level-parallel = two all-to-alls of points x L x C features / feature gradients; sharded data-parallel = reduce-scatter of the
    fp32 table gradient + all-gather of the table the kernels read.  At the reference's 1 024 rays x 192 samples (chest, bf16):
    25 MB against 85.5 MB -- and level-parallel ranks also run only 1/N of the optimiser pass, the largest kernel of a step that
    size; from ~3 300 rays per GPU and step on the gradient exchange is the smaller one.  Level-parallel needs a world size that
    divides the number of levels."""
    if world <= 1 or points_per_step is None or num_levels % world != 0:
        return "sharded"
    levels_bytes = 2 * int(points_per_step) * num_levels * level_dim * feature_bytes
    sharded_bytes = int(table_elements) * (4 + table_bytes)
    return "levels" if levels_bytes < sharded_bytes else "sharded"


def default_bucket_levels(num_levels, points_per_step=None):
    """Level ranges in the order the gradient scatter finishes them: the fine half, then the coarse half (one range for small steps).
    Two buckets of L/2 levels keep every reducer launch UNSPLIT (64 row buckets x 8 levels = 512 workgroups: a workgroup is the
    sole owner of its rows and adds its 64-bit fixed-point sums in a fixed order), so the table gradient a rank hands to the
    exchange is bit-reproducible from run to run.  Buckets of four levels or fewer (round 2's default had two of them) split each
    row bucket's tiles over several workgroups that finish with fp32 atomics -- 6.7 MB instead of 22.5 MB left exposed behind the
    last reduction, at the price of run-to-run noise in the last bits; they remain available through `bucket_levels`.  With the
    reduce-scatter / sharded-Adam / all-gather form of the exchange the tail that cannot overlap is a quarter of what the
    all-reduce left anyway.  Measured on one GPU (bench.py --force-dp --buckets ...): 9.13-9.15 ms per 65 536-ray step for the
    two halves against 9.08-9.18 ms for three buckets."""
    L = int(num_levels)
    if points_per_step is not None and int(points_per_step) < SINGLE_BUCKET_BELOW_POINTS:
        return [(0, L)]
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


def sharded_exchange_slices(slices, world, padded_end):
    """Exchange ranges for the reduce-scatter / all-gather form of the step: the bucket ranges `slices` (in the order their
    gradients become final) with every boundary moved to a multiple of world * 4 elements -- in favour of the bucket that
    finishes LATER, whose collective may touch the ragged elements because they are final by then -- and the table's end
    extended to `padded_end` (the flat buffers are padded to a multiple of lcm(64, world * 4) elements).  Every range then
    splits into `world` equal shards that start on 16-byte boundaries."""
    unit = 4 * int(world)
    if padded_end % unit:
        raise ValueError("padded_end must be a multiple of world * 4 elements")
    last = max(slices, key=lambda r: r[1])
    if padded_end < last[1]:
        raise ValueError("padded_end lies inside the table")
    stretched = [(a, padded_end) if (a, b) == last else (a, b) for a, b in slices]
    out = aligned_update_slices(stretched, multiple=unit)
    for a, b in out:
        if (b - a) % unit or b <= a:
            raise ValueError(f"bucket range [{a}, {b}) cannot be split into {world} aligned shards")
    return out


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
