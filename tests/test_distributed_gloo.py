"""CPU tier, world_size 2, gloo: the N>1 path.  Element-wise work shards with no collective
(concatenated shard results == whole-batch result); the MSM exchange (all-gather of 160-byte
partials + fold in rank order) gives every rank the same point.  The group arithmetic in this
CPU test is done by the oracle (the checker) standing in for the GPU engine."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from dusk_zerocaf_amd import distributed as D
    from oracle import zc_ref
    from tests import vectors as V
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n = 37                                                   # ragged: shards of 19 and 18
        P = V.base_multiples(zc_ref, n, V.SEED + 100)
        K = V.rand_scalars_np(n, V.SEED + 101, bits=252)
        lo, hi = D.shard_bounds(n, rank, world)
        mine = zc_ref.ed_scalar_mul(P[lo:hi], K[lo:hi])          # element-wise: no collective
        res = D.msm_sharded(P[lo:hi], K[lo:hi], zc_ref.msm_naive, zc_ref.ed_add)
        q.put((rank, lo, hi, mine, res))
    finally:
        dist.destroy_process_group()


def test_shard_bounds_cover_exactly():
    from dusk_zerocaf_amd import distributed as D
    for n in (0, 1, 7, 8, 1 << 20, (1 << 20) + 3):
        for world in (1, 2, 3, 8):
            spans = [D.shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert all(0 <= hi - lo <= (n + world - 1) // world for lo, hi in spans)


def test_two_rank_sharding_and_msm_exchange(oracle):
    import torch.multiprocessing as mp
    from tests import vectors as V
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=180) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    n = 37
    P = V.base_multiples(oracle, n, V.SEED + 100)
    K = V.rand_scalars_np(n, V.SEED + 101, bits=252)
    whole = oracle.ed_scalar_mul(P, K)
    assert np.array_equal(np.concatenate([g[3] for g in got]), whole)       # shards concatenate to the batch
    assert np.array_equal(got[0][4], got[1][4])                             # every rank: identical limbs
    want = oracle.msm_naive(P, K)
    assert oracle.ed_eq(got[0][4], want)[0] == 1                            # same group element as the serial sum
    assert np.array_equal(oracle.ed_compress(got[0][4])[0], oracle.ed_compress(want)[0])
