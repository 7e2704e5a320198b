"""One-process-per-GPU scale-out of the path.

Element-wise calls (field/scalar ops, scalar-mul, Ristretto round trip) are independent per
element: the batch is cut into contiguous per-rank ranges and NO data-path collective is
needed.  The one real exchange step is the MSM (not in the reference): every rank reduces
its shard to a single partial point, the 160-byte partials are all-gathered over xGMI and
folded IN RANK ORDER with the unified Edwards add in ONE kernel on the device, so every rank
holds the identical result.  Point addition is not an `ncclRedOp_t`, hence all-gather +
ordered fold instead of all-reduce; the payload is latency-bound by far.

Two realisations of the same step:
  * `msm_sharded_rccl`  -- everything inside libzerocaf_hip.so (zc_msm_sharded): the library's own
    RCCL communicator (created here from a unique id broadcast over torch.distributed), the
    partial never leaves HBM before the fold.
  * `msm_sharded`       -- torch.distributed moves the partials (`nccl` = RCCL on device tensors,
    `gloo` on host arrays for the CPU test tier); the fold is the same one-launch device kernel
    (Engine.ed_fold_ordered) whenever an engine is given.
"""
from __future__ import annotations

import numpy as np


def shard_bounds(n: int, rank: int, world: int):
    """Contiguous range [lo, hi) of rank `rank` -- same rule as the C ABI's in-context sharding."""
    per = (n + world - 1) // world
    lo = min(n, rank * per)
    return lo, min(n, lo + per)


def all_gather_rows(local_row, group=None):
    """all-gather one (1, w) row per rank -> (world, w), ordered by rank.  A torch CUDA tensor
    stays on the device (RCCL, `all_gather_into_tensor`); a numpy row travels over gloo."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    if hasattr(local_row, "data_ptr"):                               # device tensor: RCCL, no host hop
        t = local_row.reshape(-1).contiguous()
        out = torch.empty((world, t.numel()), dtype=t.dtype, device=t.device)
        dist.all_gather_into_tensor(out.view(-1), t, group=group)
        return out
    t = torch.from_numpy(np.ascontiguousarray(local_row).view(np.int64).reshape(-1))
    if dist.get_backend(group) == "nccl":
        t = t.cuda()
    parts = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(parts, t, group=group)
    return np.stack([p.cpu().numpy().view(np.uint64) for p in parts])


def fold_in_rank_order(rows, add_fn):
    """((r0 + r1) + r2) + ... with `add_fn` = batched Edwards add on (1, 20) rows (host fallback of
    the CPU test tier; the product path folds with Engine.ed_fold_ordered)."""
    acc = rows[0:1].copy()
    for i in range(1, rows.shape[0]):
        acc = add_fn(acc, rows[i:i + 1])
    return acc


def msm_sharded(points, scalars, local_msm, add_fn=None, group=None, engine=None):
    """sum_i k_i P_i over the ranks of `group`.  `points`/`scalars` are this rank's shard.
    With `engine` (the product path): partial sum left on the device (Engine.msm_partial), RCCL
    all-gather of device rows, one-launch ordered fold on the device; returns a (1, 20) numpy row.
    Without: `local_msm(points, scalars) -> (1, 20)` and `add_fn` stand in (CPU tier, gloo)."""
    import torch.distributed as dist
    multi = dist.is_initialized() and dist.get_world_size(group) > 1
    if engine is not None:
        import torch
        if dist.is_initialized() and dist.get_backend(group) == "nccl":
            part = engine.msm_partial(points, scalars)               # (1, 20) on the device
            rows = all_gather_rows(part, group)
            res = engine.ed_fold_ordered(rows)
            torch.cuda.current_stream().synchronize()
            return res.cpu().numpy().view(np.uint64)
        partial = engine.msm(points, scalars)
        return engine.ed_fold_ordered(all_gather_rows(partial, group)) if multi else partial
    partial = local_msm(points, scalars)
    if not multi:
        return partial
    return fold_in_rank_order(all_gather_rows(partial, group), add_fn)


def init_library_comm(engine, group=None):
    """Give `engine`'s context its own RCCL communicator: rank 0 draws the unique id, torch.distributed
    carries the 128 bytes to the other ranks (host transport only), every rank joins."""
    import torch.distributed as dist
    if dist.is_initialized():
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        box = [engine.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0, group=group)
        uid = box[0]
    else:
        rank, world, uid = 0, 1, engine.comm_unique_id()
    engine.comm_init(uid, rank, world)


def msm_sharded_rccl(engine, points, scalars):
    """zc_msm_sharded: the whole exchange inside the library (init_library_comm first)."""
    return engine.msm_sharded(points, scalars)
