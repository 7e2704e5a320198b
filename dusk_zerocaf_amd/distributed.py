"""One-process-per-GPU scale-out of the path.

Element-wise calls (field/scalar ops, scalar-mul, Ristretto round trip) are independent per
element: the batch is cut into contiguous per-rank ranges and NO data-path collective is
needed.  The one real exchange step is the MSM (not in the reference): every rank reduces
its shard to a single partial point, the 160-byte partials are all-gathered (RCCL over xGMI
with backend "nccl", or gloo on CPU) and folded IN RANK ORDER with the unified Edwards add,
so every rank holds the identical result.  Point addition is not an `ncclRedOp_t`, hence
all-gather + local fold instead of all-reduce; the payload is latency-bound by far.
"""
from __future__ import annotations

import numpy as np


def shard_bounds(n: int, rank: int, world: int):
    """Contiguous range [lo, hi) of rank `rank` -- same rule as the C ABI's in-context sharding."""
    per = (n + world - 1) // world
    lo = min(n, rank * per)
    return lo, min(n, lo + per)


def all_gather_rows(local_row: np.ndarray, group=None) -> np.ndarray:
    """all-gather one (1, w) uint64 row per rank -> (world, w), ordered by rank."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    t = torch.from_numpy(np.ascontiguousarray(local_row).view(np.int64).reshape(-1))
    backend = dist.get_backend(group)
    if backend == "nccl":
        t = t.cuda()
    parts = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(parts, t, group=group)
    return np.stack([p.cpu().numpy().view(np.uint64) for p in parts])


def fold_in_rank_order(rows: np.ndarray, add_fn) -> np.ndarray:
    """((r0 + r1) + r2) + ... with `add_fn` = batched Edwards add on (1, 20) rows."""
    acc = rows[0:1].copy()
    for i in range(1, rows.shape[0]):
        acc = add_fn(acc, rows[i:i + 1])
    return acc


def msm_sharded(points, scalars, local_msm, add_fn, group=None) -> np.ndarray:
    """sum_i k_i P_i over the ranks of `group`.  `points`/`scalars` are this rank's shard;
    `local_msm(points, scalars) -> (1, 20)` and `add_fn` come from the engine
    (Engine.msm / Engine.ed_add)."""
    import torch.distributed as dist
    partial = local_msm(points, scalars)
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return partial
    return fold_in_rank_order(all_gather_rows(partial, group), add_fn)
