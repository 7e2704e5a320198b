"""GPU tier, world_size 2: the N>1 path with the HIP engine on every rank.  One MI355X is enough:
both ranks open device 0 (RCCL refuses two ranks on one device, so the 160-byte partial sums travel
over gloo here; the RCCL call itself runs in test_gpu_parity.py::test_msm_exchange_inside_the_library).
Element-wise work shards with no collective; the sharded MSM = per-rank bucket method (zc_msm) ->
all-gather of the partials -> one-launch ordered fold on the device (zc_ed_fold_ordered): identical
limbs on both ranks, the oracle's sum as a group element."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N = 20000 + 7                                                    # ragged shards, bucket method on both (>= 4096 each)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _inputs(oracle):
    from tests import vectors as V
    base = V.base_multiples(oracle, 1024, V.SEED + 170)
    P = np.tile(base, (N // 1024 + 1, 1))[:N].copy()
    K = V.rand_scalars_np(N, V.SEED + 171, bits=252)
    K[5] = 0
    K[6] = [0, 0, 0, 0, 1 << 50]                                 # raw scalar >= 2^256: the early-stopping loop
    return P, K


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch  # noqa: F401  (first: one HIP runtime per process)
    import torch.distributed as dist
    import dusk_zerocaf_amd as z
    from dusk_zerocaf_amd import distributed as D
    from oracle import zc_ref
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        eng = z.Engine([0])
        P, K = _inputs(zc_ref)
        lo, hi = D.shard_bounds(N, rank, world)
        mine = eng.ed_scalar_mul(P[lo:hi], K[lo:hi])             # element-wise: no collective
        enc, ok = eng.ris_roundtrip_mul(eng.ris_compress(P[lo:hi]), K[lo:hi])
        res = D.msm_sharded(P[lo:hi], K[lo:hi], None, engine=eng)
        q.put((rank, mine, enc, ok, np.asarray(res)))
        eng.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_two_ranks_with_the_hip_engine(oracle):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=300) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    P, K = _inputs(oracle)
    assert np.array_equal(np.concatenate([g[1] for g in got]), oracle.mt(oracle.ed_scalar_mul, P, K))
    wenc, wok = oracle.mt(oracle.ris_roundtrip_mul, oracle.ris_compress(P), K)
    assert np.array_equal(np.concatenate([g[2] for g in got]), wenc) and np.array_equal(np.concatenate([g[3] for g in got]), wok)
    assert np.array_equal(got[0][4], got[1][4])                  # every rank: identical limbs
    want = oracle.msm_naive_mt(P, K)
    assert oracle.ed_eq(got[0][4].reshape(1, 20), want)[0] == 1
    assert np.array_equal(oracle.ed_compress(got[0][4].reshape(1, 20))[0], oracle.ed_compress(want)[0])
