#!/usr/bin/env python3
"""bench.py -- headline benchmark: batched variable-base Edwards scalar multiplication
(BASELINE.json configs[2]: 2^20 points x random 252-bit scalars per GPU) on N MI355X.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of zc_ed_scalar_mul (strict mode: the reference's formula sequence,
bit-identical (X:Y:Z:T) limbs) over the rank's 2^20 HBM-resident points and scalars.
Independent elements: the batch is sharded across ranks with no data-path collective
(weak scaling: 2^20 per GPU).  PyTorch supplies device memory, the stream and
torch.distributed; all arithmetic is in libzerocaf_hip.so.  The oracle (oracle/) is touched
only by the cpu_baseline leg, whose first slice of results also serves as the post-timing
parity spot check of the GPU output (`--cpu-sample 0` skips both).
Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import concurrent.futures as cf
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: 8 TB/s spec
BYTES_PER_UNIT = {"scalar_mul": 360, "fe_mul": 120, "ristretto": 104, "msm": 200}   # SURVEY 8(d) algorithmic bytes
# measured on MI355X with tools/ubench (profiles/r01_ubench.txt): independent v_mad_u64_u32
# chains, all CUs -- wave-instructions x 64 lanes per second


def log(*a):
    if int(os.environ.get("RANK", "0")) == 0:
        print(*a, file=sys.stderr, flush=True)


def make_inputs(eng, torch, n, seed, workload, scalar_bits=252):
    """Synthetic, seeded, generated on the GPU box: P_i = r_i * B (valid subgroup points in
    non-trivial extended coordinates, produced by the engine's fixed-base kernel) and S252 scalars."""
    rng = np.random.default_rng(seed)

    def scalars(bits):
        k = rng.integers(0, 1 << 52, size=(n, 5), dtype=np.uint64)
        k[:, 4] = rng.integers(0, 1 << (bits - 208), size=n, dtype=np.uint64)
        return k

    dev = torch.device("cuda", torch.cuda.current_device())
    to_dev = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).to(dev)
    if workload == "fe_mul":
        a, b = scalars(252), scalars(252)           # < 2^252 < p: canonical field elements
        return {"a": to_dev(a), "b": to_dev(b), "host": (a, b)}
    P = eng.ed_mul_base(to_dev(scalars(249)))
    torch.cuda.synchronize()
    K = scalars(scalar_bits)
    d = {"P": P, "K": to_dev(K), "host_K": K}
    if workload == "ristretto":
        d["enc"] = eng.ris_compress(P)
        torch.cuda.synchronize()
        d["enc"][::97, 31] |= 0x80                  # ~1 % undecodable encodings
    return d


def usable_cores():
    """Threads this process may really run at once: affinity mask capped by the cgroup quota."""
    try:
        c = len(os.sched_getaffinity(0))
    except AttributeError:
        c = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    c = min(c, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    c = min(c, max(1, q // per))
        except Exception:
            pass
    return max(1, c)


def cpu_baseline(workload, sample, host_inputs):
    """Oracle (reference-shaped C restatement) on the host cores, bounded sample: `sample`
    units in total, taken cyclically from the rank's own seeded inputs."""
    from oracle import zc_ref
    zc_ref.build()
    zc_ref.lib()
    cores = usable_cores()
    m = len(host_inputs[0])
    per_thread = max(1, sample // cores)
    fn1 = zc_ref.fe_mul if workload == "fe_mul" else zc_ref.ed_scalar_mul
    a, b = host_inputs

    first = {}

    def work(t):
        done, lo = 0, (t * per_thread) % m
        while done < per_thread:
            cnt = min(per_thread - done, m - lo, 1 << 16)
            r = fn1(a[lo:lo + cnt], b[lo:lo + cnt])
            if t == 0 and done == 0:
                first["out"] = r                    # oracle results for inputs [0, cnt): the parity spot check
            done += cnt
            lo = (lo + cnt) % m
        return done

    t0 = time.perf_counter()
    with cf.ThreadPoolExecutor(max_workers=cores) as ex:
        total = sum(ex.map(work, range(cores)))
    dt = time.perf_counter() - t0
    return total / dt, cores, dt, total, first["out"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--units", "--n", dest="n", type=int, default=1 << 20, help="units per GPU per step")
    ap.add_argument("--workload", default="scalar_mul", choices=["scalar_mul", "fe_mul", "ristretto", "msm"])
    ap.add_argument("--scalar-bits", type=int, default=252, choices=[249, 252],
                    help="252 = uniform raw 252-bit scalars (BASELINE wording, headline); 249 = the reference's Scalar::random domain")
    ap.add_argument("--mode", default="strict", choices=["strict", "fast"],
                    help="scalar_mul only: strict = reference formula sequence (bit-exact X:Y:Z:T limbs, the "
                         "headline); fast = windowed non-strict mode (same group element, labelled extra)")
    ap.add_argument("--cpu-sample", type=int, default=-1, help="units for the CPU baseline (0 disables)")
    ap.add_argument("--check", type=int, default=256, help="elements re-checked against the oracle after timing")
    args = ap.parse_args()

    import torch                                   # before the HIP library: one HIP runtime per process
    import torch.distributed as dist
    import dusk_zerocaf_amd as z

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    if args.gpus != world:
        log("note: --gpus %d but WORLD_SIZE %d; using WORLD_SIZE" % (args.gpus, world))

    eng = z.Engine([local])
    stream = torch.cuda.current_stream()
    eng.set_stream(stream.cuda_stream)
    n = args.n
    data = make_inputs(eng, torch, n, 0x5EED0003 + rank, args.workload, args.scalar_bits)

    if args.workload == "scalar_mul":
        out = torch.empty_like(data["P"])
        flags = z.FAST if args.mode == "fast" else z.STRICT
        step = lambda: eng.ed_scalar_mul(data["P"], data["K"], out=out, flags=flags)
    elif args.workload == "fe_mul":
        step = lambda: eng.fe_mul(data["a"], data["b"])
        out = None
    elif args.workload == "msm":
        # BASELINE configs[4] shape: every rank reduces its shard with the bucket method, the
        # 160-byte partials are all-gathered (RCCL) and folded in rank order
        from dusk_zerocaf_amd import distributed as D
        out = None
        msm_result = []

        def step():
            msm_result[:] = [D.msm_sharded(data["P"], data["K"], eng.msm, eng.ed_add)]
    else:
        out = torch.empty_like(data["enc"])
        step = lambda: eng.ris_roundtrip_mul(data["enc"], data["K"], out=out)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for i in range(args.steps):
        ev[i][0].record(stream)
        step()
        ev[i][1].record(stream)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    kern_ms = [a.elapsed_time(b) for a, b in ev]
    kern_avg_s = sum(kern_ms) / len(kern_ms) * 1e-3

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    units = n * world * args.steps
    value = units / dt
    unit_bytes = BYTES_PER_UNIT[args.workload]
    achieved = unit_bytes * n / kern_avg_s / 1e9
    roofline = {"bound": "hbm", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": None,
                "kernel": {"scalar_mul": "k_ed_scalar_mul", "fe_mul": "k_fe_mul", "ristretto": "k_ris_roundtrip_mul",
                           "msm": "k_msm_accumulate (+ rocPRIM radix sort, reduce, fold)"}[args.workload],
                "kernel_avg_ms": round(kern_avg_s * 1e3, 4), "algorithmic_bytes_per_unit": unit_bytes}
    # HBM bytes per launch from the committed rocprofv3 --pmc passes (FETCH_SIZE x2 on gfx950 +
    # WRITE_SIZE, profiles/r01_pmc_summary.md), scaled to this launch's unit count
    pmc = os.path.join(ROOT, "profiles", "r01_hbm_traffic.json")
    if os.path.exists(pmc):
        try:
            t = json.load(open(pmc))
            if args.workload == "scalar_mul":
                roofline["traffic"] = round(t["scalar_mul"] * n / (1 << 20)) if args.mode == "strict" else None
            elif args.workload == "fe_mul":
                roofline["traffic"] = round(t["fe_mul_per_unit_bytes"] * n)
        except Exception:
            pass
    ub = os.path.join(ROOT, "profiles", "r01_ubench.json")
    mix = os.path.join(ROOT, "profiles", "r01_isa_mix.json")
    if args.workload == "scalar_mul" and args.mode == "strict" and args.scalar_bits == 252 and os.path.exists(ub) and os.path.exists(mix):
        try:
            u, m = json.load(open(ub)), json.load(open(mix))
            mad_peak = float(u["v_mad_u64_u32_lane_ops_per_s"])          # lane-ops/s, all CUs, tools/ubench
            insts = float(json.load(open(pmc))["scalar_mul_valu_insts_per_launch"]) * n / (1 << 20)
            # The step loop's ISA (tools/isa_mix.py): 9 Montgomery multiplications x (135 v_mad_u64_u32 +
            # 9 v_mul_lo_u32 + 16 v_lshrrev_b64).  These run at the multiplier's rate (~5 cycles per
            # wave-instruction per SIMD); the 32-bit ALU ops in between were measured not to cost issue
            # time (removing ~100 of them per step changed the kernel time by 0.3 %).
            share = m["multiplier_rate_class_per_step"] / float(m["valu_per_step"])
            t_mult = insts * share * 64 / mad_peak
            roofline["valu"] = {
                "note": "integer-VALU bound, not HBM bound (0.2% of HBM peak is expected): multiplier_bound_ms = "
                        "PMC SQ_INSTS_VALU x share of multiplier-rate instructions in the step loop (ISA histogram) "
                        "/ microbenchmarked v_mad_u64_u32 throughput of the whole chip",
                "valu_wave_insts_per_launch": insts, "multiplier_rate_share": round(share, 4),
                "per_step": {"valu": m["valu_per_step"], "multiplier_rate": m["multiplier_rate_class_per_step"]},
                "v_mad_u64_u32_peak_lane_ops_per_s": mad_peak,
                "multiplier_bound_ms": round(t_mult * 1e3, 3),
                "frac_of_multiplier_peak": round(t_mult / kern_avg_s, 4)}
        except Exception:
            pass

    cpu, checked = None, None
    sample = args.cpu_sample
    if sample < 0:
        sample = {"scalar_mul": 1 << 13, "ristretto": 1 << 13, "fe_mul": 1 << 24, "msm": 1 << 13}[args.workload] * usable_cores()
    if sample:
        if args.workload == "fe_mul":
            hi = data["host"]
        else:
            m = min(sample, n)
            hi = (data["P"][:m].cpu().numpy().view(np.uint64), data["host_K"][:m])
        v, cores, secs, total, want = cpu_baseline("fe_mul" if args.workload == "fe_mul" else "scalar_mul", sample, hi)
        # the baseline's first slice doubles as the post-timing parity spot check of the GPU result
        if args.check and args.workload in ("scalar_mul", "fe_mul"):
            k = min(args.check, len(want))
            res = out if args.workload == "scalar_mul" else step()
            torch.cuda.synchronize()
            got = res[:k].cpu().numpy().view(np.uint64)
            if args.workload == "scalar_mul" and args.mode == "fast":    # same group element: compare encodings
                enc = lambda pts: eng.ed_compress(torch.from_numpy(np.ascontiguousarray(pts).view(np.int64)).cuda())[0].cpu().numpy()
                checked = bool(np.array_equal(enc(got), enc(want[:k])))
            else:
                checked = bool(np.array_equal(got, want[:k]))
            if not checked:
                raise SystemExit("PARITY FAILURE: GPU result differs from the oracle")
        cpu = {"value": round(v, 1), "unit": "scalar-muls/s" if args.workload != "fe_mul" else "field-muls/s",
               "cores": cores, "kind": "port",
               "sample": "%d units of the same seeded workload, %d threads, %.1f s wall (%.0f s of CPU work); C "
                         "restatement of zerocaf's u64 backend (oracle/zc_ref.c, gcc -O3), not the Rust binary"
                         % (total, cores, secs, secs * cores)}

    line = {
        "metric": ("252-bit Edwards variable-base scalar-muls/sec (batched, strict bit-exact mode)" if args.mode == "strict"
                   else "252-bit Edwards variable-base scalar-muls/sec (batched, FAST non-strict mode)")
        if args.workload == "scalar_mul" else
        ("MSM point-scalar pairs/sec (bucket method per GPU, all-gather + ordered fold across GPUs)" if args.workload == "msm"
         else {"ristretto": "Ristretto decompress -> scalar-mul -> compress round trips/sec (fused, bit-exact encodings)",
               "fe_mul": "FieldElement multiplications/sec (batched, bit-exact canonical limbs)"}[args.workload]),
        "value": round(value, 1),
        "unit": {"fe_mul": "field-muls/s", "msm": "pairs/s", "ristretto": "round-trips/s"}.get(args.workload, "scalar-muls/s"),
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u64 (nine 29-bit limbs in u32 registers, 64-bit multiply-accumulate columns)", "data": "synthetic",
        "config": {"workload": {"scalar_mul": "2^20 EdwardsPoint variable-base scalar-mul, random %d-bit scalars (BASELINE configs[2])" % args.scalar_bits,
                                "fe_mul": "2^20 FieldElement mul (BASELINE configs[1])",
                                "ristretto": "Ristretto decompress->scalar-mul->compress (BASELINE configs[3] shape)",
                                "msm": "Pippenger MSM, 249-bit scalars, one shard per GPU (BASELINE configs[4] shape)"}[args.workload],
                   "units_per_gpu_per_step": n,
                   "sharding": "contiguous ranges; all-gather of one 160-byte partial per rank + ordered fold" if args.workload == "msm"
                               else "contiguous ranges, no collective",
                   "mode": {"scalar_mul": "strict (reference formula sequence, identical X:Y:Z:T limbs)" if args.mode == "strict"
                                          else "FAST (non-strict extra: same group element / encodings, limbs differ by a projective factor)",
                            "fe_mul": "bit-exact canonical limbs",
                            "ristretto": "bit-exact 32-byte encodings and ok mask",
                            "msm": "result compared as a group element (canonical encoding)"}[args.workload]},
        "roofline": roofline,
        "cpu_baseline": cpu,
        "parity_spot_check": checked,
    }
    print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
