"""GPU tier, REAL multi-rank RCCL: self-skips unless the box shows at least two devices, and needs no edit on an
8-GPU node -- it runs 2 and then every visible device (up to 8) as one rank per GPU, the library's own communicator
(zc_comm_init -> ncclAllGather over xGMI inside zc_msm_sharded) carrying the one exchange of the path:

  * the sharded strict scalar-mul (no collective) concatenates to the oracle's whole-batch result, limb for limb;
  * zc_msm_sharded ends with IDENTICAL limbs on every rank, the oracle's sum of `&P_i * &k_i` as a group element,
    and the rank-ordered fold (the reference's unified addition) of the ranks' partial sums limb for limb;
  * one rank handed a null pointer makes EVERY rank return an error (the poison record) -- nobody hangs;
  * `bench.py --workload msm --gpus N`, launched the way the driver launches it, passes its own
    `rccl_ranks == N` / `distinct_devices == N` guards.

On a one-GPU box the same code paths run at world size 1 (tests/test_gpu_parity.py::test_msm_exchange_inside_the_library)
and over gloo with two ranks on one device (tests/test_distributed_gpu.py)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N = (1 << 17) + 12345                                            # ragged shards; every shard runs the bucket method


def _device_count():
    try:
        import dusk_zerocaf_amd as z
        return z.load().zc_device_count()
    except Exception:
        return 0


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _inputs(eng):
    from tests import vectors as V
    P = eng.ed_mul_base(V.rand_scalars_np(N, V.SEED + 500, bits=249))
    K = V.rand_scalars_np(N, V.SEED + 501, bits=252)
    K[5] = 0
    K[6] = [0, 0, 0, 0, 1 << 50]                                 # raw scalar >= 2^256: the early-stopping loop
    K[N - 3] = [(1 << 52) - 1] * 5
    return P, K


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch
    import torch.distributed as dist
    import dusk_zerocaf_amd as z
    from dusk_zerocaf_amd import distributed as D
    torch.cuda.set_device(rank)
    dist.init_process_group("gloo", rank=rank, world_size=world)             # host transport for the 128-byte id and the barriers only
    try:
        eng = z.Engine([rank])
        P, K = _inputs(eng)                                                  # same seeds on every rank: the same global batch
        lo, hi = D.shard_bounds(N, rank, world)
        dP = torch.from_numpy(P[lo:hi].view(np.int64)).cuda()
        dK = torch.from_numpy(K[lo:hi].view(np.int64)).cuda()
        mine = eng.ed_scalar_mul(dP, dK).cpu().numpy().view(np.uint64)       # element-wise: no collective
        ident = [z.Engine.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(ident, src=0)
        eng.comm_init(ident[0], rank, world)
        ranks = eng.comm_size()
        total = np.asarray(eng.msm_sharded(dP, dK))                          # the exchange: ncclAllGather + ordered fold
        total_host = np.asarray(eng.msm_sharded(P[lo:hi], K[lo:hi]))         # host inputs take the same exchange
        part = eng.msm_partial(dP, dK).cpu().numpy().view(np.uint64)
        # the failure path: rank world-1 passes null points; every rank must come back with an error
        dist.barrier()
        import ctypes as C
        out = np.empty((1, 20), dtype=np.uint64)
        if rank == world - 1:
            rc = eng.lib.zc_msm_sharded(eng.ctx, None, C.c_void_p(dK.data_ptr()), hi - lo, C.c_void_p(out.ctypes.data))
        else:
            rc = eng.lib.zc_msm_sharded(eng.ctx, C.c_void_p(dP.data_ptr()), C.c_void_p(dK.data_ptr()), hi - lo, C.c_void_p(out.ctypes.data))
        err = eng.lib.zc_last_error().decode()
        again = np.asarray(eng.msm_sharded(dP, dK))                          # and the communicator still works afterwards
        uuid = str(torch.cuda.get_device_properties(rank).uuid) if hasattr(torch.cuda.get_device_properties(rank), "uuid") else str(rank)
        q.put((rank, mine, total, total_host, part, ranks, rc, err, again, uuid))
        eng.comm_destroy()
        eng.close()
    finally:
        dist.destroy_process_group()


def _run_world(world, oracle):
    import torch.multiprocessing as mp
    import dusk_zerocaf_amd as z
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=900) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    eng = z.Engine([0])
    try:
        P, K = _inputs(eng)
    finally:
        eng.close()
    assert [g[5] for g in got] == [world] * world                            # ncclCommCount on every rank
    assert len({g[9] for g in got}) == world                                 # one physical device per rank
    assert np.array_equal(np.concatenate([g[1] for g in got]), oracle.mt(oracle.ed_scalar_mul, P, K))
    want = oracle.msm_naive_mt(P, K)
    for g in got:
        assert np.array_equal(g[2], got[0][2]) and np.array_equal(g[3], got[0][2]) and np.array_equal(g[8], got[0][2])   # identical limbs, every rank, every call
    total = got[0][2].reshape(1, 20)
    assert oracle.ed_eq(total, want)[0] == 1
    assert np.array_equal(oracle.ed_compress(total)[0], oracle.ed_compress(want)[0])
    fold = got[0][4].reshape(1, 20)                                          # ((p_0 + p_1) + p_2) + ...: the reference's unified addition
    for g in got[1:]:
        fold = oracle.ed_add(fold, g[4].reshape(1, 20))
    assert np.array_equal(fold, total)
    assert all(g[6] != 0 for g in got), [g[6] for g in got]                  # the poisoned exchange failed on EVERY rank
    assert "null points" in got[world - 1][7] and all("failed its local part" in g[7] for g in got[:world - 1])


@pytest.mark.gpu
@pytest.mark.skipif(_device_count() < 2, reason="needs at least two GPUs: real multi-rank RCCL (one rank per device)")
def test_two_ranks_over_rccl(oracle):
    _run_world(2, oracle)


@pytest.mark.gpu
@pytest.mark.skipif(_device_count() < 3, reason="needs more than two GPUs")
def test_every_visible_device_over_rccl(oracle):
    _run_world(min(8, _device_count()), oracle)


@pytest.mark.gpu
@pytest.mark.skipif(_device_count() < 2, reason="needs at least two GPUs: real multi-rank RCCL (one rank per device)")
@pytest.mark.parametrize("workload,units", [("msm", 1 << 18), ("scalar_mul", 1 << 18)])
def test_bench_passes_its_own_multi_gpu_guards(workload, units):
    world = min(8, _device_count())
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "3", "--warmup", "1",
           "--workload", workload, "--units", str(units)]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=1200, cwd=ROOT, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == world and d["distinct_devices"] == world and d["parity_spot_check"] is True and d["scaling"] == "weak"
    assert abs(d["value"] - world * units / (d["ms_per_step"] * 1e-3)) < 0.01 * d["value"]
    if workload == "msm":
        assert d["rccl_ranks"] == world and d["msm_result_is_fold_of_shard_partials"] is True
