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


# ---------------------------------------------------------------------------------------------------------------
# ONE process, ONE context over several DISTINCT devices (zc_ctx_create(devices, ndev > 1), include/zerocaf_hip.h:73-79;
# what the Rust shim's HipBackend::new(&[0..7]) binds): host batches are split over the slots (per-device staging rings,
# comb tables, worker threads), zc_msm gathers the per-slot partial sums with hipMemcpyPeerAsync and folds them on slot 0,
# a device pointer is served by the slot that owns its device.  On a one-GPU box the same code runs as Engine([0, 0])
# (tests/test_gpu_parity.py); these run it on devices that really differ.  SURVEY 8(e): "one HIP stream + pinned staging per GPU".

def _ctx_inputs(eng, oracle):
    from tests import vectors as V
    n = 60001                                                    # ragged split over 2..8 slots
    P = eng.ed_mul_base(V.rand_scalars_np(n, V.SEED + 510, bits=249))
    K = V.rand_scalars_np(n, V.SEED + 511, bits=252)
    K[7] = 0
    K[8] = [0, 0, 0, 0, 1 << 50]
    K[n - 1] = [(1 << 52) - 1] * 5
    return n, P, K


@pytest.mark.gpu
@pytest.mark.skipif(_device_count() < 2, reason="needs at least two GPUs: one context over distinct devices")
@pytest.mark.parametrize("ndev", [2, 8])
def test_one_context_over_distinct_devices(oracle, ndev):
    have = _device_count()
    if ndev > 2 and have < 3:
        pytest.skip("needs more than two GPUs")
    _check_one_context(list(range(min(ndev, have, 8))), oracle)


@pytest.mark.gpu
def test_one_context_three_slots_on_one_device(oracle):
    """The same checks on any box: three slots of one context on device 0 (keeps the multi-device test's own code exercised)."""
    _check_one_context([0, 0, 0], oracle)


def _check_one_context(devs, oracle):
    import torch
    import dusk_zerocaf_amd as z
    eq = np.array_equal
    multi = z.Engine(devs)
    try:
        n, P, K = _ctx_inputs(multi, oracle)                     # the points themselves come from the multi-device comb tables
        from tests import vectors as V
        import bench                                             # BASEPOINT limbs (constants.rs:188-211)
        m = 4000                                                 # rows from the first and the last slot's ranges: k B as a group element
        R = V.rand_scalars_np(n, V.SEED + 510, bits=249)
        rows = np.r_[0:m // 2, n - m // 2:n]
        wb = oracle.mt(oracle.ed_scalar_mul, np.tile(np.array(bench.BASEPOINT_LIMBS, dtype=np.uint64), (m, 1)), R[rows])
        assert oracle.ed_eq(P[rows], wb).all() and eq(oracle.ris_compress(P[rows]), oracle.ris_compress(wb))
        # host pointers: contiguous ranges over all slots, results in the caller's output range, limb for limb the oracle's batch
        assert eq(multi.ed_scalar_mul(P, K), oracle.mt(oracle.ed_scalar_mul, P, K))
        enc = oracle.ris_compress(P)
        enc[::97, 31] |= 0x80                                    # undecodable rows: ok = 0, zero bytes
        gout, gok = multi.ris_roundtrip_mul(enc, K)
        wout, wok = oracle.mt(oracle.ris_roundtrip_mul, enc, K)
        assert eq(gout, wout) and eq(gok, wok)
        a, b = V.rand_fe_np(n, V.SEED + 512), V.rand_fe_np(n, V.SEED + 513)
        assert eq(multi.fe_mul(a, b), oracle.fe_mul(a, b))
        # zc_msm over the slots: per-slot bucket method, hipMemcpyPeerAsync gather, ordered fold on slot 0
        got = multi.msm(P, K)
        want = oracle.msm_naive_mt(P, K)
        assert oracle.ed_eq(got, want)[0] == 1 and eq(oracle.ed_compress(got)[0], oracle.ed_compress(want)[0])
        # ... and limb for limb the ordered fold (the reference's unified addition) of the per-slot partial sums,
        # each computed by a one-device context on that slot's own contiguous range
        from dusk_zerocaf_amd import distributed as D
        fold = None
        for slot, dev in enumerate(devs):
            lo, hi = D.shard_bounds(n, slot, len(devs))
            one = z.Engine([dev])
            try:
                with torch.cuda.device(dev):
                    dP = torch.from_numpy(P[lo:hi].view(np.int64)).cuda()
                    dK = torch.from_numpy(K[lo:hi].view(np.int64)).cuda()
                    part = one.msm_partial(dP, dK).cpu().numpy().view(np.uint64).reshape(1, 20)
            finally:
                one.close()
            fold = part if fold is None else oracle.ed_add(fold, part)
        assert eq(np.asarray(got).reshape(1, 20), fold)
        # a device pointer on device k is served by slot k (no staging, result on the same device)
        for dev in (devs[-1], devs[0]):
            with torch.cuda.device(dev):
                dP = torch.from_numpy(P[:5000].view(np.int64)).cuda()
                dK = torch.from_numpy(K[:5000].view(np.int64)).cuda()
                out = multi.ed_scalar_mul(dP, dK)
                torch.cuda.synchronize(dev)
                assert out.device.index == dev and eq(out.cpu().numpy().view(np.uint64), oracle.mt(oracle.ed_scalar_mul, P[:5000], K[:5000]))
                m = multi.msm(dP, dK)
                assert oracle.ed_eq(np.asarray(m).reshape(1, 20), oracle.msm_naive_mt(P[:5000], K[:5000]))[0] == 1
    finally:
        multi.close()


@pytest.mark.gpu
@pytest.mark.skipif(_device_count() < 2, reason="needs at least two GPUs: a pointer on a device outside the context")
def test_pointer_outside_the_context_is_refused(oracle):
    """include/zerocaf_hip.h ZC_ERR_MIXED_MEM: device buffers must belong to a device of the context (and to one device)."""
    import ctypes as C
    import torch
    import dusk_zerocaf_amd as z
    from tests import vectors as V
    one = z.Engine([0])
    try:
        n = 1000
        P = one.ed_mul_base(V.rand_scalars_np(n, V.SEED + 520, bits=249))
        K = V.rand_scalars_np(n, V.SEED + 521, bits=252)
        with torch.cuda.device(1):
            dP, dK = torch.from_numpy(P.view(np.int64)).cuda(), torch.from_numpy(K.view(np.int64)).cuda()
            out = torch.empty_like(dP)
        torch.cuda.synchronize(1)
        rc = one.lib.zc_ed_scalar_mul(one.ctx, C.c_void_p(dP.data_ptr()), C.c_void_p(dK.data_ptr()), C.c_void_p(out.data_ptr()), n, 0)
        assert rc == -5 and "do not belong to a device of this context" in one.lib.zc_last_error().decode()
        res = np.empty((1, 20), dtype=np.uint64)
        rc = one.lib.zc_msm(one.ctx, C.c_void_p(dP.data_ptr()), C.c_void_p(dK.data_ptr()), n, C.c_void_p(res.ctypes.data))
        assert rc == -5
        with pytest.raises(z.ZerocafHipError):                   # the Python mirror refuses before any launch
            one.ed_scalar_mul(dP, dK)
        # buffers of one call on two different devices
        both = z.Engine([0, 1])
        try:
            d0P = torch.from_numpy(P.view(np.int64)).to("cuda:0")
            out0 = torch.empty_like(d0P)
            rc = both.lib.zc_ed_scalar_mul(both.ctx, C.c_void_p(d0P.data_ptr()), C.c_void_p(dK.data_ptr()), C.c_void_p(out0.data_ptr()), n, 0)
            assert rc == -5 and "different devices" in both.lib.zc_last_error().decode()
            assert np.array_equal(both.ed_scalar_mul(P, K), oracle.mt(oracle.ed_scalar_mul, P, K))      # and the context still works
        finally:
            both.close()
    finally:
        one.close()
