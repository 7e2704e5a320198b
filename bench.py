#!/usr/bin/env python3
"""bench.py -- headline benchmark: batched variable-base Edwards scalar multiplication
(BASELINE.json configs[2]: 2^20 points x random 252-bit scalars per GPU) on N MI355X.

    python bench.py --gpus 1 --steps K --warmup W          (defaults: K = 20, W = 5 -- the driver's own invocation)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of zc_ed_scalar_mul (strict mode: the reference's formula sequence,
bit-identical (X:Y:Z:T) limbs) over the rank's 2^20 HBM-resident points and scalars.
Independent elements: the batch is sharded across ranks with no data-path collective
(weak scaling: 2^20 per GPU).  `--workload msm` is the one path with an exchange step
(BASELINE configs[4]): zc_msm_sharded = bucket method per GPU, ncclAllGather of the 160-byte
partial sums inside the library, ordered fold on the device.  `--workload ecdh` is the reference's only
macro-benchmark (benchmarks/dusk_benchmarks.rs:544-620): two key generations and two shared secrets per unit.
PyTorch supplies device memory, the stream and torch.distributed; all arithmetic is in
libzerocaf_hip.so.  The oracle (oracle/) is touched only after the timed region: the
cpu_baseline leg (the same operation on the host cores) whose results double as the parity
spot check of the GPU output (`--cpu-sample 0` skips both).
Prints ONE JSON line on rank 0.

The default single-GPU run (the driver's) also carries `secondary`: the other BASELINE configs -- fe_mul 2^24 and
fe_invert 2^20 (configs[1]), the Ristretto round trip 2^22 (configs[3]), the MSM shard 2^21 (configs[4]) -- and the ECDH
exchange, a few steps each AFTER the headline's timed region and outside its ms_per_step, each with its own roofline
record and oracle spot check (`--no-secondary` skips them).

roofline (schema "useful-work/2", since round 3; rounds 1-2 reported issued work as `frac`): the scalar-mul / Ristretto /
MSM / ECDH kernels are bound by the integer multiplier pipe (v_mad_u64_u32 class), not by HBM (0.2 % of 8 TB/s), so
`bound` = "valu_int_mul":
  achieved = USEFUL v_mad_u64_u32 lane-operations per second: the multiplications the algorithm needs x 135 per
             Montgomery multiplication / kernel time (HIP events, live) -- no profile input at all.  Strict scalar-mul:
             sum over this run's scalars of (bitlen - 1 + popcount) evaluations of the reference's addition formula x 9
             multiplications; MSM: non-zero window digits x the multiplications of one bucket addition (7 with affine
             records, 8 without -- the library's own plan, zc_msm_plan); Ristretto round trip: the windowed core's fixed
             schedule + the two codecs; ECDH: two comb multiplications + two encodings + two round trips.
  peak     = one multiplier-class wave-instruction per 4 shader cycles per SIMD at the nominal clock (a hard
             roof: 39.3 T lane-ops/s); `measured_rate` = the saturated v_mad_u64_u32 rate of this board measured
             in this run (libzc_ubench.so), with the same fractions against it
  frac     = achieved / peak (useful work only).  `frac_issued` = every multiplier-class instruction the kernel
             issued (PMC SQ_INSTS_VALU per unit from profiles/roofline_inputs.json x the class's share of the
             loop's VALU instructions, tools/isa_mix.py): reduction bookkeeping, carry shifts and lost wave steps
             included.  The PMC-derived fields are dropped (null, with a note) when the profile was taken on other
             kernel sources than the loaded library was built from (sha256 of csrc/ + the header, in zc_version()).
The HBM view (algorithmic bytes / time vs 8 TB/s, PMC traffic) is kept under `hbm`.  fe_mul is HBM-bound:
`bound` = "hbm" (tagged `cache_resident` when the launch's arrays fit the 256 MB Infinity Cache: then the figure
is cache bandwidth, not HBM).  fe_invert is neither: `bound` = "latency/occupancy" with the resident waves per SIMD.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: 8 TB/s spec
MADS_PER_MUL = 135               # v_mad_u64_u32 per Montgomery multiplication (zc_arith.hip.h, column-ordered)
MADS_PER_SQR = 99
ROOFLINE_SCHEMA = "useful-work/2"
WORKLOADS = {
    # algorithmic bytes per unit: SURVEY 8(d)
    "scalar_mul": {"bytes": 360, "kernel": "k_ed_scalar_mul_pw (+ k_sm_cost_hist/scan/scatter)", "bound": "valu_int_mul", "unit": "scalar-muls/s"},
    "fe_mul": {"bytes": 120, "kernel": "k_fe_mul", "bound": "hbm", "unit": "field-muls/s"},
    "fe_invert": {"bytes": 80, "kernel": "k_fe_invert_chunked", "bound": "latency/occupancy", "unit": "field-inversions/s"},
    "ristretto": {"bytes": 104, "kernel": "k_ris_roundtrip_mul_fast", "bound": "valu_int_mul", "unit": "round-trips/s"},
    "msm": {"bytes": 200, "kernel": "k_msm_runs_affine (+ k_msm_digits, k_msm_sort_hist/scatter + k_scan_*, k_msm_prepare_affine, k_msm_runs_edges/segments/fold_groups/window_combine)",
            "bound": "valu_int_mul", "unit": "pairs/s"},
    # two secret scalars in, two public keys and two shared secrets out (32-byte Ristretto encodings)
    "ecdh": {"bytes": 80 + 128, "kernel": "2 x k_ris_mul_base_compress + 2 x k_ris_roundtrip_mul_fast", "bound": "valu_int_mul", "unit": "exchanges/s"},
}
INFINITY_CACHE_BYTES = 256 << 20
HBM_COPY_GBS = 6290.0            # MI355X_MICROARCH.md: what a device-to-device copy reaches (the achievable HBM rate)
WINDOW_MADS = 4 * (4 * MADS_PER_SQR + 3 * MADS_PER_MUL) + MADS_PER_MUL + 8 * MADS_PER_MUL     # one 4-bit window of the windowed core
CODEC_MADS = 250 * MADS_PER_SQR + 61 * MADS_PER_MUL          # one (p-5)/8 power (~250 squarings + ~36 multiplications) + ~25 multiplications of glue
ROUNDTRIP_MADS = 63 * WINDOW_MADS + 55 * MADS_PER_MUL + 2 * CODEC_MADS
KEYGEN_MADS = 33 * 7 * MADS_PER_MUL + CODEC_MADS             # radix-256 comb: 33 mixed additions, then one encoding
# BASEPOINT (src/backend/u64/constants.rs:188-211), X | Y | Z | T limbs: the operand of the reference's key generation
BASEPOINT_LIMBS = [276718085098056, 1646536057461434, 2704687245600312, 2630386667454967, 13476148227069,
                   1303868825475266, 3250718520537114, 2702159777242978, 2702159776422297, 10555311626649,
                   1, 0, 0, 0, 0,
                   3634527586288175, 2006028620404053, 3424252198034825, 2478951925947079, 4567251727358]


def size_label(n):
    """2^k for powers of two, the number otherwise -- so that a record names the size it was taken at."""
    return "2^%d" % (n.bit_length() - 1) if n > 0 and n & (n - 1) == 0 else str(n)


def workload_label(wl, n, bits, ecdh="wire"):
    s = size_label(n)
    return {"scalar_mul": "%s EdwardsPoint variable-base scalar-mul, random %d-bit scalars (BASELINE configs[2])" % (s, bits),
            "fe_mul": "%s FieldElement mul (BASELINE configs[1])" % s,
            "fe_invert": "%s FieldElement invert (BASELINE configs[1]); division-step inversion shared by up to 32 elements per lane" % s,
            "ristretto": "%s Ristretto decompress->scalar-mul->compress, random %d-bit scalars, ~1%% undecodable inputs (BASELINE configs[3] shape)" % (s, bits),
            "msm": "%s-pair Pippenger MSM, %d-bit scalars, one shard per GPU (BASELINE configs[4] shape)" % (s, bits),
            "ecdh": "%s ECDH exchanges (benchmarks/dusk_benchmarks.rs:544-620): two key generations + two shared secrets per unit, %d-bit secrets, %s"
                    % (s, bits, "32-byte Ristretto encodings on the wire" if ecdh == "wire" else "the reference's double_and_add on extended points")}[wl]


def msm_nonzero_digits(K, c):
    """Number of non-zero signed c-bit window digits over all scalars (k_msm_digits' recoding): the bucket additions an
    MSM of these scalars cannot avoid."""
    n = len(K)
    W = -(-261 // c)
    carry = np.zeros(n, dtype=np.uint64)
    half, mask, total = np.uint64(1 << (c - 1)), np.uint64((1 << c) - 1), 0
    for w in range(W):
        bit = w * c
        idx, sh = bit // 52, bit % 52
        if idx >= 5:
            raw = carry.copy()
        else:
            x = K[:, idx] >> np.uint64(sh)
            if sh + c > 52 and idx + 1 < 5:
                x = x | (K[:, idx + 1] << np.uint64(52 - sh))
            raw = (x & mask) + carry
        carry = (raw > half).astype(np.uint64)
        total += int(np.count_nonzero(raw != carry * np.uint64(1 << c)))
    return total, W


def strict_formula_evaluations(K):
    """sum over the scalars of (bitlen - 1 + popcount): evaluations of the reference's addition formula in double_and_add."""
    n = len(K)
    bits = np.zeros(n, dtype=np.int64)
    pop = np.zeros(n, dtype=np.int64)
    for j in range(5):
        x = K[:, j]
        nz = x != 0
        bl = np.zeros(n, dtype=np.int64)
        bl[nz] = np.floor(np.log2(x[nz].astype(np.float64))).astype(np.int64) + 1
        bits = np.where(nz, 52 * j + bl, bits)
        pop += _popcount64(x)
    return int(np.sum(np.where(bits > 0, bits - 1 + pop, 0)))


def log(*a):
    if int(os.environ.get("RANK", "0")) == 0:
        print(*a, file=sys.stderr, flush=True)


class Env:
    """What every workload of one bench.py process shares."""
    def __init__(self, torch, z, eng, stream, rank, world, backend, dist):
        self.torch, self.z, self.eng, self.stream = torch, z, eng, stream
        self.rank, self.world, self.backend, self.dist = rank, world, backend, dist
        self.measured = None          # the saturated v_mad_u64_u32 rate of this board, measured once per process
        self.measured_done = False
        self.comm_ready = False

    def barrier(self):
        if self.world > 1:
            self.dist.barrier()
        self.torch.cuda.synchronize()


def make_inputs(env, n, seed, workload, scalar_bits=252, ecdh="wire"):
    """Synthetic, seeded, generated on the GPU box: P_i = r_i * B (valid subgroup points in
    non-trivial extended coordinates, produced by the engine's fixed-base kernel) and raw scalars."""
    torch, eng = env.torch, env.eng
    rng = np.random.default_rng(seed)

    def scalars(bits):
        k = rng.integers(0, 1 << 52, size=(n, 5), dtype=np.uint64)
        k[:, 4] = rng.integers(0, 1 << (bits - 208), size=n, dtype=np.uint64)
        return k

    dev = torch.device("cuda", torch.cuda.current_device())
    to_dev = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).to(dev)
    if workload in ("fe_mul", "fe_invert"):
        a, b = scalars(252), scalars(252)           # < 2^252 < p: canonical field elements
        if workload == "fe_invert":
            a[::1009] = 0                           # a few zeros: the reference's inverse() panics there (ok = 0, out = 0)
        return {"a": to_dev(a), "b": to_dev(b), "host": (a, b)}
    if workload == "ecdh":
        a, b = scalars(scalar_bits), scalars(scalar_bits)          # Alice's and Bob's secrets
        d = {"a": to_dev(a), "b": to_dev(b), "host": (a, b)}
        if ecdh == "reference":
            d["base"] = to_dev(np.tile(np.array(BASEPOINT_LIMBS, dtype=np.uint64), (n, 1)))
        return d
    P = eng.ed_mul_base(to_dev(scalars(249)))
    torch.cuda.synchronize()
    K = scalars(scalar_bits)
    d = {"P": P, "K": to_dev(K), "host_K": K}
    if workload == "ristretto":
        d["enc"] = eng.ris_compress(P)
        torch.cuda.synchronize()
        d["enc"][::97, 31] |= 0x80                  # ~1 % undecodable encodings
    return d


def make_run(env, wl, n, scalar_bits, mode="strict", ecdh="wire"):
    """The inputs of one workload on this rank and the function that performs ONE step of it."""
    torch, z, eng = env.torch, env.z, env.eng
    run = {"wl": wl, "n": n, "bits": scalar_bits, "mode": mode, "ecdh": ecdh, "st": {}}
    data = run["data"] = make_inputs(env, n, 0x5EED0003 + env.rank, wl, scalar_bits, ecdh)
    st = run["st"]
    if wl == "scalar_mul":
        st["out"] = torch.empty_like(data["P"])
        flags = z.FAST if mode == "fast" else z.STRICT
        run["step"] = lambda: eng.ed_scalar_mul(data["P"], data["K"], out=st["out"], flags=flags)
    elif wl == "fe_mul":
        run["step"] = lambda: eng.fe_mul(data["a"], data["b"])
    elif wl == "fe_invert":
        run["step"] = lambda: eng.fe_invert(data["a"])
    elif wl == "msm":
        # the whole exchange inside the library: local bucket method -> ncclAllGather of the 160-byte
        # partial sums on the library's own RCCL communicator -> ordered fold kernel -> host
        from dusk_zerocaf_amd import distributed as D
        if env.backend == "nccl" or env.world == 1:
            if not env.comm_ready:
                D.init_library_comm(eng)
                env.comm_ready = True

            def step():
                st["result"] = eng.msm_sharded(data["P"], data["K"])
        else:                                       # test hook (gloo, ranks sharing a device): partials over torch.distributed
            def step():
                st["result"] = D.msm_sharded(data["P"], data["K"], None, engine=eng)
        run["step"] = step
    elif wl == "ristretto":
        st["out"] = torch.empty_like(data["enc"])

        def step():
            st["ok"] = eng.ris_roundtrip_mul(data["enc"], data["K"], out=st["out"])[1]
        run["step"] = step
    elif ecdh == "wire":
        # A = (a B).compress(), B' = (b B).compress(), S = (a * B'.decompress()).compress(), S' = (b * A.decompress()).compress()
        def step():
            A = eng.ris_mul_base_compress(data["a"])
            Bp = eng.ris_mul_base_compress(data["b"])
            S, ok1 = eng.ris_roundtrip_mul(Bp, data["a"])
            Sp, ok2 = eng.ris_roundtrip_mul(A, data["b"])
            st.update(A=A, Bp=Bp, S=S, Sp=Sp, ok1=ok1, ok2=ok2)
        run["step"] = step
    else:
        # the reference's own ecdh_double_add: four double_and_add calls on extended points, limb-exact
        def step():
            A = eng.ed_scalar_mul(data["base"], data["a"])
            Bp = eng.ed_scalar_mul(data["base"], data["b"])
            st.update(A=A, Bp=Bp, S=eng.ed_scalar_mul(Bp, data["a"]), Sp=eng.ed_scalar_mul(A, data["b"]))
        run["step"] = step
    return run


def time_run(env, run, steps, warmup):
    """W untimed steps, then exactly K steps bracketed by barrier + synchronize; HIP events on the launch stream per step."""
    torch = env.torch
    for _ in range(warmup):
        run["step"]()
    env.barrier()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    t0 = time.perf_counter()
    for i in range(steps):
        ev[i][0].record(env.stream)
        run["step"]()
        ev[i][1].record(env.stream)
    env.barrier()
    dt = time.perf_counter() - t0
    if env.world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        env.dist.all_reduce(t, op=env.dist.ReduceOp.MAX)
        dt = float(t.item())
    kern_ms = [a.elapsed_time(b) for a, b in ev]
    return dt, sum(kern_ms) / len(kern_ms) * 1e-3


def roofline_inputs():
    """profiles/roofline_inputs.json: PMC per-unit figures, ISA shares and ubench peaks of one build."""
    path = os.path.join(ROOT, "profiles", "roofline_inputs.json")
    if not os.path.exists(path):
        return None, "profiles/roofline_inputs.json missing"
    inp = json.load(open(path))
    import dusk_zerocaf_amd as z
    built_from = z.load().zc_version().decode().rsplit("src:", 1)[-1]      # hash of the sources the loaded library was built from
    if inp.get("kernel_sources_sha256") != built_from:
        return inp, "profile taken on other kernel sources (sha256 %s...) than the loaded library was built from (%s...): PMC-derived fields dropped" % (
            str(inp.get("kernel_sources_sha256"))[:12], built_from[:12])
    return inp, None


def measured_rate(env, inp):
    """The saturated v_mad_u64_u32 rate of THIS board in THIS run (libzc_ubench.so), once per process."""
    if env.measured_done:
        return dict(env.measured) if env.measured else None
    env.measured_done = True
    measured = None
    try:
        import ctypes
        ub = ctypes.CDLL(os.path.join(os.path.dirname(env.z.LIB_PATH), "libzc_ubench.so"))
        ub.zc_ubench_mad_u64_u32.restype = ctypes.c_double
        ub.zc_ubench_mad_u64_u32.argtypes = [ctypes.c_double, ctypes.POINTER(ctypes.c_double)]
        ghz = ctypes.c_double(0.0)
        env.torch.cuda.synchronize()
        rate = ub.zc_ubench_mad_u64_u32(20.0, ctypes.byref(ghz))
        if rate > 0:
            measured = {"v_mad_u64_u32_T_lane_ops_per_s": round(rate, 2), "shader_clock_ghz": round(ghz.value, 3),
                        "source": "libzc_ubench.so in this run: 8 waves per SIMD of independent multiply-accumulate chains, 20 ms"}
    except OSError:
        pass
    if measured is None and (inp or {}).get("ubench", {}).get("v_mad_u64_u32_T_lane_ops_per_s"):
        measured = {"v_mad_u64_u32_T_lane_ops_per_s": inp["ubench"]["v_mad_u64_u32_T_lane_ops_per_s"],
                    "source": str(inp["ubench"].get("source")) + " (another run / board: libzc_ubench.so not built)"}
    env.measured = measured
    return dict(measured) if measured else None


def roofline_for(env, run, kern_avg_s):
    torch, eng = env.torch, env.eng
    wl, n, mode, data = run["wl"], run["n"], run["mode"], run["data"]
    W = WORKLOADS[wl]
    hbm_achieved = W["bytes"] * n / kern_avg_s / 1e9
    hbm = {"achieved": round(hbm_achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(hbm_achieved / HBM_PEAK_GBS, 6),
           "algorithmic_bytes_per_unit": W["bytes"], "traffic": None}
    inp, stale = roofline_inputs()
    kkey = wl if not (wl == "scalar_mul" and mode == "fast") else "scalar_mul_fast"
    kin = (inp or {}).get("kernels", {}).get(kkey, {}) if not stale else {}
    if wl == "scalar_mul" and run["bits"] != 252:
        kin = {}                                    # the PMC profile was taken on 252-bit scalars: no issued-work fields for another domain
    if kin.get("hbm_bytes_per_unit") is not None and kin.get("units_per_call") in (None, n):
        hbm["traffic"] = round(kin["hbm_bytes_per_unit"] * n)
        hbm["traffic_source"] = inp.get("source")
    kernel = W["kernel"] if kkey != "scalar_mul_fast" else "k_ed_scalar_mul_fast"
    if wl == "ecdh" and run["ecdh"] == "reference":
        kernel = "4 x k_ed_scalar_mul_pw"
    roofline = {"schema": ROOFLINE_SCHEMA, "bound": W["bound"], "kernel": kernel, "kernel_avg_ms": round(kern_avg_s * 1e3, 4)}
    props = torch.cuda.get_device_properties(torch.cuda.current_device())
    if W["bound"] == "hbm":
        roofline.update({k: hbm[k] for k in ("achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_unit")})
        # the arrays of one launch against the 256 MB Infinity Cache: below it the launches of a benchmark loop find
        # their inputs in the cache and `achieved` is cache bandwidth, not HBM
        roofline["cache_resident"] = bool(W["bytes"] * n <= INFINITY_CACHE_BYTES)
        if roofline["cache_resident"]:
            roofline["note"] = "%d MB per launch fit the 256 MB Infinity Cache: cache bandwidth, not an HBM figure (the HBM line is --units 16777216)" % (W["bytes"] * n >> 20)
        return roofline
    if W["bound"] == "latency/occupancy":
        # fe_invert: one division-step inversion per lane shared by c elements (Montgomery's trick); at 2^20 elements
        # only half a wave per SIMD is resident, so neither HBM nor the multiplier is the limit
        c = min(32, n // 65536)
        lanes = n if c < 2 else -(-n // c)
        roofline.update({"achieved": hbm["achieved"], "peak": hbm["peak"], "unit": hbm["unit"], "frac": hbm["frac"], "traffic": hbm["traffic"],
                         "algorithmic_bytes_per_unit": W["bytes"], "elements_per_lane": max(1, c), "lanes": lanes,
                         "resident_waves_per_simd": round(lanes / 64 / (props.multi_processor_count * 4), 3),
                         "note": "latency / occupancy bound: the figures are the HBM view for reference only; see DESIGN 4.2"})
        return roofline
    # peak: a v_mad_u64_u32-class wave-instruction cannot issue faster than once per 4 shader cycles per
    # SIMD (the best ever measured on this chip is 4.4, tools/ubench/occupancy.hip), priced at the
    # nominal clock: CUs x 4 SIMDs x 64 lanes x f_max / 4.  A hard roof; the board never holds f_max
    # under this load, so the saturated rate MEASURED in this very run (libzc_ubench.so, same board, same
    # thermal state, right after the headline's timed region) is reported beside it with its own fraction.
    f_max = (getattr(props, "clock_rate", 0) or 2400000) * 1e3
    peak = round(props.multi_processor_count * 4 * 64 * f_max / 4 / 1e12, 2)
    measured = measured_rate(env, inp)
    roofline.update({"achieved": None, "peak": peak, "unit": "T lane-ops/s (v_mad_u64_u32, 64 lanes per wave-instruction)",
                     "frac": None, "traffic": hbm["traffic"], "hbm": hbm,
                     "peak_basis": "one multiplier-class wave-instruction per 4 shader cycles per SIMD at the nominal %.1f GHz" % (f_max / 1e9),
                     "measured_rate": measured})
    # ---- useful work: the multiplications the algorithm needs, from this run's own inputs (no profile)
    useful, basis, digits, plan = None, None, None, None
    if wl == "scalar_mul" and mode == "strict":
        evals = strict_formula_evaluations(data["host_K"])
        useful = evals * 9 * MADS_PER_MUL
        basis = {"formula_evaluations_per_unit": round(evals / n, 2), "multiplications_per_evaluation": 9}
    elif wl == "scalar_mul":
        # windowed core: 63 four-bit windows of (3 x (4S + 3M) + (4S + 4M) doublings + one 8M addition) + the 55M table
        useful = n * (63 * WINDOW_MADS + 55 * MADS_PER_MUL)
        basis = {"windows": 63, "v_mad_u64_u32_per_window": WINDOW_MADS}
    elif wl == "ristretto":
        # the same core between one decompression and one compression
        useful = n * ROUNDTRIP_MADS
        basis = {"windows": 63, "v_mad_u64_u32_per_window": WINDOW_MADS, "v_mad_u64_u32_per_codec": CODEC_MADS}
    elif wl == "ecdh" and run["ecdh"] == "wire":
        useful = n * 2 * (KEYGEN_MADS + ROUNDTRIP_MADS)
        basis = {"per_unit": "2 key generations (33 mixed additions of 7 multiplications + one encoding) + 2 round trips (decode, windowed core, encode)",
                 "v_mad_u64_u32_per_key_generation": KEYGEN_MADS, "v_mad_u64_u32_per_round_trip": ROUNDTRIP_MADS}
    elif wl == "ecdh":
        a, b = data["host"]
        evals = 2 * (strict_formula_evaluations(a) + strict_formula_evaluations(b))
        useful = evals * 9 * MADS_PER_MUL
        basis = {"formula_evaluations_per_unit": round(evals / n, 2), "multiplications_per_evaluation": 9,
                 "per_unit": "4 double_and_add calls (a B, b B, a (b B), b (a B))"}
    elif wl == "msm":
        # what the library itself plans for this shard (zc_msm_plan): window width, record form, multiplications per bucket addition
        plan = eng.msm_plan(n, points_aligned16=data["P"].data_ptr() % 16 == 0)
        if plan["window_bits"]:
            digits, nwin = msm_nonzero_digits(data["host_K"], plan["window_bits"])
            mpa = 7 if plan["affine"] else 8
            useful = digits * mpa * MADS_PER_MUL            # one mixed / cached addition per non-zero digit
            basis = {"window_bits": plan["window_bits"], "windows": nwin, "nonzero_digits_per_pair": round(digits / n, 3),
                     "multiplications_per_bucket_addition": mpa, "plan": plan,
                     "note": "bucket reduction, window combination and the affine normalisation are overhead, not counted"}
    if useful is not None:
        roofline["achieved"] = round(useful / kern_avg_s / 1e12, 3)
        roofline["frac"] = round(useful / kern_avg_s / 1e12 / peak, 4)
        roofline["useful"] = dict(basis, v_mad_u64_u32_lane_ops=useful)
        if measured:
            measured["frac"] = round(useful / kern_avg_s / 1e12 / measured["v_mad_u64_u32_T_lane_ops_per_s"], 4)
    # ---- issued work: every multiplier-class instruction (PMC x ISA share), only with a profile of THIS build at THIS size
    if peak and kin.get("valu_insts_per_unit") and kin.get("multiplier_rate_share") and kin.get("units_per_call") in (None, n):
        lane_ops = kin["valu_insts_per_unit"] * n * kin["multiplier_rate_share"] * 64
        roofline["issued"] = {"achieved": round(lane_ops / kern_avg_s / 1e12, 3), "source": inp.get("source"),
                              "valu_wave_insts_per_unit": kin["valu_insts_per_unit"], "multiplier_rate_share": kin["multiplier_rate_share"],
                              "kernels_counted": kin.get("kernels_counted")}
        roofline["frac_issued"] = round(roofline["issued"]["achieved"] / peak, 4)
        if measured:
            measured["frac_issued"] = round(roofline["issued"]["achieved"] / measured["v_mad_u64_u32_T_lane_ops_per_s"], 4)
    if wl == "msm" and digits is not None:
        # the gathers of the bucket sums: one cached record per non-zero digit, random over the record array
        rec = plan["record_bytes"]                   # the payload a gather requests (112 affine: 27 limb words in seven 16-byte pieces / 128 projective); the stride is what it touches
        g = {"record_bytes": rec, "record_stride_bytes": plan["record_stride"], "records": digits, "unit": "GB/s", "peak": HBM_COPY_GBS,
             "peak_basis": "what a device copy reaches (MI355X_MICROARCH.md); random %d-byte gathers, one per %d-byte slot" % (rec, plan["record_stride"])}
        runs = (kin.get("runs") or {}).get(str(n))
        if runs:
            # the bucket-sum kernel alone (its duration per step in this build's kernel trace): gathered bytes against the
            # copy rate, and its multiplications against the multiplier roof -- the higher fraction names the bound
            t_runs = runs["avg_ms"] * 1e-3
            g["achieved"] = round(rec * digits / t_runs / 1e9, 1)
            g["frac"] = round(g["achieved"] / HBM_COPY_GBS, 4)
            g["k_msm_runs_avg_ms"] = runs["avg_ms"]
            g["k_msm_runs_time_share"] = runs["time_share"]
            g["k_msm_runs_multiplier_frac"] = round(useful / t_runs / 1e12 / peak, 4)
            g["source"] = runs["source"]
            roofline["bound"] = "gather" if g["frac"] > g["k_msm_runs_multiplier_frac"] else "valu_int_mul"
        roofline["gather"] = g
    if stale:
        roofline["profile_note"] = stale
    return roofline


SAMPLE_SEED = 0x5A3D


def spread_sample(n, m, seed=SAMPLE_SEED):
    """Which m of a batch's n units the oracle recomputes: the first m/4, the last m/4 (the ragged last workgroup, the tail
    of a persistent kernel's schedule) and a seeded, jittered stride through everything between them (one unit out of every
    `step` consecutive ones, its place inside the stretch drawn at random: every residue modulo the wave, workgroup and tile
    sizes occurs).  Sorted, no duplicates; all n when m >= n."""
    if m >= n:
        return np.arange(n, dtype=np.int64)
    h = m // 4
    mid = m - 2 * h
    step = (n - 2 * h) // mid
    rng = np.random.default_rng(seed ^ n)
    middle = h + step * np.arange(mid, dtype=np.int64) + rng.integers(0, step, size=mid)
    return np.concatenate([np.arange(h, dtype=np.int64), middle, np.arange(n - h, n, dtype=np.int64)])


def take_rows(env, t, idx):
    """rows idx of a device tensor, on the host"""
    return t.index_select(0, env.torch.from_numpy(idx).to(t.device)).cpu().numpy()


def cpu_leg(env, run, sample):
    """The same operation on the host cores with the oracle (reference-shaped C restatement), on a
    bounded sample of the rank's own inputs spread over the whole batch (spread_sample: head, tail and a seeded
    stride through the middle).  Returns (rate, cores, seconds, units, oracle results for the sampled units,
    their indices) -- the results are the parity spot check."""
    from oracle import zc_ref
    zc_ref.build()
    zc_ref.lib()
    wl, n, data = run["wl"], run["n"], run["data"]
    cores = zc_ref.host_threads()
    m = min(sample, n)
    idx = spread_sample(n, m)
    host = lambda t: np.ascontiguousarray(take_rows(env, t, idx)).view(np.uint64)
    pick = lambda a: np.ascontiguousarray(a[idx])
    t0 = time.perf_counter()
    if wl == "fe_mul":
        a, b = pick(data["host"][0]), pick(data["host"][1])
        t0 = time.perf_counter()
        reps = max(1, sample // m)
        want = None
        for _ in range(reps):
            want = zc_ref.mt(zc_ref.fe_mul, a, b)
        m *= reps
    elif wl == "fe_invert":
        a = pick(data["host"][0])
        t0 = time.perf_counter()
        want = zc_ref.mt(zc_ref.fe_invert, a)
    elif wl == "scalar_mul":
        want = zc_ref.mt(zc_ref.ed_scalar_mul, host(data["P"]), pick(data["host_K"]))
    elif wl == "ristretto":
        want = zc_ref.mt(zc_ref.ris_roundtrip_mul, np.ascontiguousarray(take_rows(env, data["enc"], idx)), pick(data["host_K"]))
    elif wl == "ecdh":
        # the reference's ecdh_double_add: key pairs with double_and_add on the basepoint, then the two shared secrets
        a, b = pick(data["host"][0]), pick(data["host"][1])
        base = np.tile(np.array(BASEPOINT_LIMBS, dtype=np.uint64), (m, 1))
        pa = zc_ref.mt(zc_ref.ed_scalar_mul, base, a)
        pb = zc_ref.mt(zc_ref.ed_scalar_mul, base, b)
        s1 = zc_ref.mt(zc_ref.ed_scalar_mul, pb, a)
        s2 = zc_ref.mt(zc_ref.ed_scalar_mul, pa, b)
        want = (pa, pb, s1, s2)
    else:                                           # msm: the reference's own sum of Mul<Scalar> over the sampled pairs
        want = zc_ref.msm_naive_mt(host(data["P"]), pick(data["host_K"]))
    dt = time.perf_counter() - t0
    return m / dt, cores, dt, m, want, idx


def check_and_baseline(env, run, sample, baseline_leg, msm_fold_ok=None):
    """Oracle results for a sample spread over the whole batch (spread_sample) against the GPU's outputs at the same places
    (a mismatch aborts the run); the CPU rate beside it."""
    from oracle import zc_ref
    torch, eng = env.torch, env.eng
    wl, n, data, st, mode = run["wl"], run["n"], run["data"], run["st"], run["mode"]
    v, cores, secs, total, want, idx = cpu_leg(env, run, sample)
    torch.cuda.synchronize()
    rows = lambda t: take_rows(env, t, idx)                  # the GPU's outputs for the sampled units
    if wl == "scalar_mul":
        got = rows(st["out"]).view(np.uint64)
        if mode == "fast":                           # same group element: compare encodings
            enc = lambda pts: eng.ed_compress(torch.from_numpy(np.ascontiguousarray(pts).view(np.int64)).cuda())[0].cpu().numpy()
            checked = bool(np.array_equal(enc(got), enc(want)))
        else:
            checked = bool(np.array_equal(got, want))
    elif wl == "fe_mul":
        got = run["step"]()
        torch.cuda.synchronize()
        checked = bool(np.array_equal(rows(got).view(np.uint64), want))
    elif wl == "fe_invert":
        got, gok = run["step"]()
        torch.cuda.synchronize()
        checked = bool(np.array_equal(rows(got).view(np.uint64), want[0]) and np.array_equal(rows(gok), want[1]))
    elif wl == "ristretto":
        wout, wok = want
        checked = bool(np.array_equal(rows(st["out"]), wout) and np.array_equal(rows(st["ok"]), wok))
    elif wl == "ecdh":
        pa, pb, s1, s2 = want
        if run["ecdh"] == "wire":
            # the bytes on the wire are the reference's: compress() of its own key pairs and shared secrets; and EVERY exchange agrees
            checked = bool(np.array_equal(rows(st["A"]), zc_ref.ris_compress(pa)) and np.array_equal(rows(st["Bp"]), zc_ref.ris_compress(pb))
                           and np.array_equal(rows(st["S"]), zc_ref.ris_compress(s1)) and np.array_equal(rows(st["Sp"]), zc_ref.ris_compress(s2))
                           and bool(torch.equal(st["S"], st["Sp"])) and bool(st["ok1"].all()) and bool(st["ok2"].all()))
        else:
            g = lambda key: rows(st[key]).view(np.uint64)
            checked = bool(np.array_equal(g("A"), pa) and np.array_equal(g("Bp"), pb) and np.array_equal(g("S"), s1) and np.array_equal(g("Sp"), s2)
                           and bool(eng.ris_eq(st["S"], st["Sp"]).all()))
    else:
        # MSM: the GPU sum over the sampled pairs (head, tail and the seeded stride, gathered) against the oracle's sum of
        # the same pairs, compared as canonical encodings (zc_msm contract: a group element)
        it = torch.from_numpy(idx).to(data["P"].device)
        sub = eng.msm(data["P"].index_select(0, it).contiguous(), data["K"].index_select(0, it).contiguous())
        checked = bool(np.array_equal(zc_ref.ed_compress(sub)[0], zc_ref.ed_compress(want)[0]) and zc_ref.ed_eq(sub, want)[0] == 1)
        if total == n and env.world == 1:
            checked = checked and bool(np.array_equal(zc_ref.ed_compress(st["result"])[0], zc_ref.ed_compress(want)[0]))
        # beyond the oracle's sample the timed result was checked as the ordered fold of the shard partials
        checked = checked and msm_fold_ok is True
    if not checked:
        raise SystemExit("PARITY FAILURE (%s, %d units): GPU result differs from the oracle" % (wl, n))
    if not baseline_leg:
        return checked, None
    what = {"scalar_mul": "zr_ed_scalar_mul (double_and_add)", "fe_mul": "zr_fe_mul", "fe_invert": "zr_fe_inverse (Savas-Koc, field.rs:854-925)",
            "ristretto": "zr_ris_roundtrip_mul (decompress, double_and_add, compress)",
            "msm": "zr_msm_naive (sum of double_and_add results with the unified add)",
            "ecdh": "4 x zr_ed_scalar_mul per unit (the reference's ecdh_double_add: two key pairs, two shared secrets)"}[wl]
    try:
        model = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except Exception:
        model = "unknown CPU"
    single = None
    if cores > 1:
        # the measured one-thread leg: the same operation on 1 / cores of the sample, one host thread
        m1 = min(n, max(1, total // cores))
        host1 = lambda t: np.ascontiguousarray(t[:m1].cpu().numpy()).view(np.uint64)
        t1 = time.perf_counter()
        if wl == "fe_mul":
            zc_ref.fe_mul(data["host"][0][:m1], data["host"][1][:m1])
        elif wl == "fe_invert":
            zc_ref.fe_invert(data["host"][0][:m1])
        elif wl == "scalar_mul":
            zc_ref.ed_scalar_mul(host1(data["P"]), data["host_K"][:m1])
        elif wl == "ristretto":
            zc_ref.ris_roundtrip_mul(data["enc"][:m1].cpu().numpy(), data["host_K"][:m1])
        elif wl == "ecdh":
            base1 = np.tile(np.array(BASEPOINT_LIMBS, dtype=np.uint64), (m1, 1))
            a1, b1 = data["host"][0][:m1], data["host"][1][:m1]
            pa1, pb1 = zc_ref.ed_scalar_mul(base1, a1), zc_ref.ed_scalar_mul(base1, b1)
            zc_ref.ed_scalar_mul(pb1, a1)
            zc_ref.ed_scalar_mul(pa1, b1)
        else:
            zc_ref.msm_naive(host1(data["P"]), data["host_K"][:m1])
        single = {"value": round(m1 / (time.perf_counter() - t1), 1), "units": m1}
    cpu = {"value": round(v, 1), "unit": WORKLOADS[wl]["unit"], "cores": cores, "kind": "port",
           "value_per_core": round(v / cores, 1), "value_single_core": single["value"] if single else round(v, 1),
           "single_core_sample_units": single["units"] if single else total, "cpu_model": model,
           "sample": "%d units of the same seeded workload spread over the whole batch (first quarter of the sample from the head, last quarter "
                     "from the tail, the rest a seeded jittered stride through the middle), %d threads, %.1f s wall (%.0f s of CPU work): %s; C "
                     "restatement of zerocaf's u64 backend (oracle/zc_ref.c, gcc %s, built on this host), not the Rust binary"
                     % (total, cores, secs, secs * cores, what, zc_ref.build_flags())}
    return checked, cpu


def default_sample(wl, world, cores):
    per_core = {"scalar_mul": 1 << 13, "ristretto": 1 << 12, "fe_mul": 1 << 24, "fe_invert": 1 << 16, "msm": 1 << 13, "ecdh": 1 << 11}[wl]
    return per_core * cores if world == 1 else {"fe_mul": 1 << 16, "fe_invert": 1 << 14}.get(wl, 1 << 11)   # N > 1: parity check only


def msm_fold_check(env, run):
    """config 5, fail closed: the timed result must be the ordered fold of the ranks' own partial sums (zc_msm_partial ->
    torch.distributed all-gather -> zc_ed_fold_ordered: another route than the in-library exchange), and each rank's
    partial must be the ordered fold of the partial sums of its eight contiguous sub-ranges."""
    torch, eng, dist = env.torch, env.eng, env.dist
    from dusk_zerocaf_amd import distributed as D
    data, n = run["data"], run["n"]
    part = eng.msm_partial(data["P"], data["K"])
    per = -(-n // 8)
    sub = torch.cat([eng.msm_partial(data["P"][lo:lo + per], data["K"][lo:lo + per]) for lo in range(0, n, per)])
    refold = eng.ed_fold_ordered(sub)
    torch.cuda.synchronize()
    same = lambda a, b: bool(eng.ed_eq(a, b).cpu().numpy()[0] == 1 and
                             np.array_equal(eng.ed_compress(a)[0].cpu().numpy(), eng.ed_compress(b)[0].cpu().numpy()))
    ok = same(part, refold)
    rows = part
    if env.world > 1:
        if env.backend == "nccl":
            rows = D.all_gather_rows(part)
        else:
            rows = torch.from_numpy(D.all_gather_rows(part.cpu().numpy().view(np.uint64)).view(np.int64)).cuda()
    total_pt = eng.ed_fold_ordered(rows)
    timed = torch.from_numpy(np.ascontiguousarray(run["st"]["result"]).view(np.int64)).cuda()
    ok = ok and same(total_pt, timed)
    if env.world > 1:
        flags = [None] * env.world
        dist.all_gather_object(flags, bool(ok))
        ok = all(flags)
    if not ok:
        raise SystemExit("PARITY FAILURE: the timed MSM result is not the ordered fold of the shard partials")
    return ok


def mode_label(wl, mode, ecdh):
    return {"scalar_mul": "strict (reference formula sequence, identical X:Y:Z:T limbs)" if mode == "strict"
                          else "FAST (non-strict extra: same group element / encodings, limbs differ by a projective factor)",
            "fe_mul": "bit-exact canonical limbs",
            "fe_invert": "bit-exact canonical limbs and ok mask (zero inputs)",
            "ristretto": "bit-exact 32-byte encodings and ok mask",
            "msm": "result compared as a group element (canonical encoding)",
            "ecdh": "bit-exact 32-byte encodings of both public keys and both shared secrets; S == S' on every element" if ecdh == "wire"
                    else "limb-exact public keys and shared secrets (the reference's double_and_add); S == S' on every element"}[wl]


# the other BASELINE configs + the reference's macro-benchmark, run after the headline in the default single-GPU line:
# (workload, units, scalar bits, timed steps, warm-up steps, oracle sample per host thread)
SECONDARY = [("fe_mul", 1 << 24, 252, 20, 30, 1 << 14), ("fe_invert", 1 << 20, 252, 20, 5, 1 << 10), ("ristretto", 1 << 22, 252, 3, 1, 128),
             ("msm", 1 << 21, 249, 10, 3, 256), ("ecdh", 1 << 20, 249, 3, 1, 48),
             # the headline once more on the reference's own Scalar::random domain (< 2^249, SURVEY 8d config 3: "report both")
             ("scalar_mul", 1 << 20, 249, 10, 2, 1 << 10)]


def secondary_name(wl, bits):
    return "scalar_mul_s%d" % bits if wl == "scalar_mul" else wl


def summary_line(name, rec):
    """One short line per config for the driver's 2000-character tail: rate, time per step, roofline fraction and bound, parity."""
    rf = rec["roofline"]
    return "secondary %s: %.4g %s, %.4f ms/step, frac %s of %s roof, parity %s" % (
        name, rec["value"], rec["unit"], rec["ms_per_step"], rf.get("frac"), rf.get("bound"), rec["parity_spot_check"])


def secondary_runs(env):
    out = []
    from oracle import zc_ref
    cores = zc_ref.host_threads()
    for wl, n, bits, steps, warmup, per_core in SECONDARY:
        t_all = time.perf_counter()
        run = make_run(env, wl, n, bits)
        dt, kern = time_run(env, run, steps, warmup)
        fold_ok = msm_fold_check(env, run) if wl == "msm" else None
        rf = roofline_for(env, run, kern)
        checked, cpu = check_and_baseline(env, run, per_core * cores, True, fold_ok)
        rec = {"workload": workload_label(wl, n, bits), "name": secondary_name(wl, bits), "units": n, "steps": steps, "warmup": warmup, "ms_per_step": round(dt / steps * 1e3, 4),
               "value": round(n * steps / dt, 1), "unit": WORKLOADS[wl]["unit"], "mode": mode_label(wl, "strict", "wire"), "roofline": rf,
               "parity_spot_check": checked, "parity_sample_units": per_core * cores if wl != "fe_mul" else min(n, per_core * cores),
               "cpu_value": cpu["value"] if cpu else None, "cpu_cores": cores, "wall_s": None}
        if wl == "msm":
            rec["msm_result_is_fold_of_shard_partials"] = fold_ok
            rec["rccl_ranks"] = env.eng.comm_size()
        del run
        env.torch.cuda.empty_cache()
        rec["wall_s"] = round(time.perf_counter() - t_all, 2)
        out.append(rec)
        log("secondary %s: %.4f ms per step, parity %s (%.1f s)" % (rec["name"], rec["ms_per_step"], checked, rec["wall_s"]))
    return out


def self_launch(args, json_fd):
    """`python bench.py --gpus N` with no launcher environment: start the N ranks here, one device each, the way
    torch.distributed.run would (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*, rendezvous on 127.0.0.1), pass rank 0's ONE
    JSON line through to stdout and fail when any rank fails.  Fail closed: fewer than N visible devices is an error
    (unless the one-device test hook ZC_BENCH_DEVICE is set), never a smaller measurement."""
    import socket
    import subprocess
    n = args.gpus
    if not os.environ.get("ZC_BENCH_DEVICE"):
        import torch
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < n:
            raise SystemExit("bench.py: --gpus %d but %d visible device%s: refusing to time fewer GPUs than asked for"
                             % (n, have, "" if have == 1 else "s"))
    # What the ranks inherit can be overridden from the caller's environment (none of it has met a
    # second GPU yet): ZC_BENCH_MASTER_ADDR / ZC_BENCH_MASTER_PORT (default: 127.0.0.1 and a free port -- a MASTER_PORT left over
    # in the caller's environment is NOT trusted: it may be taken), HSA_ENABLE_IPC_MODE_LEGACY when set (default 0: this image's
    # host driver only supports dmabuf IPC, RCCL's device-memory exchange fails without it).
    addr = os.environ.get("ZC_BENCH_MASTER_ADDR", "127.0.0.1")
    port = os.environ.get("ZC_BENCH_MASTER_PORT")
    if not port:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR=addr,
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, cwd=ROOT,
                                      stdout=subprocess.PIPE if r == 0 else 2, stderr=2))
    import threading
    got = []
    reader = threading.Thread(target=lambda: got.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    deadline = time.time() + float(os.environ.get("ZC_BENCH_LAUNCH_TIMEOUT", "1800"))
    failed = None
    while any(p.poll() is None for p in procs):
        bad = [r for r, p in enumerate(procs) if p.poll() not in (None, 0)]
        if bad or time.time() > deadline:            # one rank down (or the time is up): the others would wait in a barrier for ever
            failed = "rank %s failed" % bad if bad else "timed out"
            for p in procs:
                if p.poll() is None:
                    p.kill()
            break
        time.sleep(0.05)
    rcs = [p.wait() for p in procs]
    reader.join(10)
    lines = [l for l in (got[0] if got else b"").decode(errors="replace").splitlines() if l.strip()]
    if failed or any(rcs) or len(lines) != 1:
        raise SystemExit("bench.py: self-launched ranks failed (%s; exit codes %s; %d stdout lines from rank 0)" % (failed, rcs, len(lines)))
    os.write(json_fd, (lines[0] + "\n").encode())


def main():
    # RCCL and the HIP runtime print banners on fd 1 ("RCCL version : ...") when a communicator is created;
    # the contract is ONE JSON line on stdout, so everything else written to fd 1 goes to stderr.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)          # the defaults are the driver's own invocation (--steps 20 --warmup 5):
    ap.add_argument("--warmup", type=int, default=5)         # five launches take a cold board past its clock / power transient
    ap.add_argument("--units", "--n", dest="n", type=int, default=1 << 20, help="units per GPU per step")
    ap.add_argument("--workload", default="scalar_mul", choices=list(WORKLOADS))
    ap.add_argument("--scalar-bits", type=int, default=None, choices=[249, 252],
                    help="252 = uniform raw 252-bit scalars (BASELINE wording, headline); 249 = the reference's Scalar::random domain "
                         "(default for msm and ecdh: SURVEY 8d config 5 / the reference's key generation)")
    ap.add_argument("--mode", default="strict", choices=["strict", "fast"],
                    help="scalar_mul only: strict = reference formula sequence (bit-exact X:Y:Z:T limbs, the "
                         "headline); fast = windowed non-strict mode (same group element, labelled extra)")
    ap.add_argument("--ecdh", default="wire", choices=["wire", "reference"],
                    help="ecdh only: wire = public keys and shared secrets as 32-byte Ristretto encodings (comb key generation, fused "
                         "round trips); reference = the reference's ecdh_double_add literally (four strict double_and_add calls)")
    ap.add_argument("--cpu-sample", type=int, default=-1, help="units for the CPU baseline / parity check (0 disables)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the other configs' lines in the default single-GPU run")
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        return self_launch(args, json_fd)            # plain `python bench.py --gpus N`: launch the N ranks ourselves

    import torch                                   # before the HIP library: one HIP runtime per process
    import torch.distributed as dist
    import dusk_zerocaf_amd as z

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:                           # fail closed: never time another number of GPUs than asked for
        raise SystemExit("bench.py: --gpus %d but the launcher's WORLD_SIZE is %d" % (args.gpus, world))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    # test hooks for boxes with fewer GPUs than ranks (tests/test_bench_contract.py): every rank on one device,
    # gloo instead of RCCL (RCCL refuses two ranks on one device); never set by the driver
    backend = os.environ.get("ZC_BENCH_BACKEND", "nccl")
    if os.environ.get("ZC_BENCH_DEVICE"):
        local = int(os.environ["ZC_BENCH_DEVICE"])
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    eng = z.Engine([local])
    stream = torch.cuda.current_stream()
    eng.set_stream(stream.cuda_stream)             # kernels and the timing events share this stream
    env = Env(torch, z, eng, stream, rank, world, backend, dist)
    n, wl = args.n, args.workload
    if args.scalar_bits is None:
        args.scalar_bits = 249 if wl in ("msm", "ecdh") else 252
    headline_default = (wl == "scalar_mul" and n == 1 << 20 and args.mode == "strict" and args.scalar_bits == 252)

    run = make_run(env, wl, n, args.scalar_bits, args.mode, args.ecdh)
    dt, kern_avg_s = time_run(env, run, args.steps, args.warmup)

    # what makes a multi-GPU line self-proving: which physical device every rank ran on (N distinct ones), the rank
    # count RCCL itself reports for the library's communicator, and every rank's own kernel time (a straggler shows)
    pr = torch.cuda.get_device_properties(local)
    ident = {"rank": rank, "local_device": local, "name": pr.name, "uuid": str(getattr(pr, "uuid", "")),
             "pci": "%04x:%02x:%02x" % (getattr(pr, "pci_domain_id", 0), getattr(pr, "pci_bus_id", 0), getattr(pr, "pci_device_id", 0)),
             "kernel_avg_ms": round(kern_avg_s * 1e3, 4)}
    rccl_ranks = None
    if wl == "msm" and (backend == "nccl" or world == 1):
        rccl_ranks = eng.comm_size()
        ident["rccl_ranks"] = rccl_ranks
    idents = [ident]
    if world > 1:
        idents = [None] * world
        dist.all_gather_object(idents, ident)
    shared_device_hook = bool(os.environ.get("ZC_BENCH_DEVICE"))
    distinct = len({(d["uuid"], d["pci"]) for d in idents})
    if world > 1 and not shared_device_hook and distinct != world:
        raise SystemExit("bench.py: %d ranks but only %d distinct devices: %s" % (world, distinct, idents))
    if rccl_ranks is not None and backend == "nccl" and any(d.get("rccl_ranks") != world for d in idents):
        raise SystemExit("bench.py: the library's RCCL communicator does not span %d ranks: %s" % (world, idents))

    msm_fold_ok = None
    if wl == "msm" and args.cpu_sample != 0:                 # --cpu-sample 0 skips every parity check (profiling runs)
        msm_fold_ok = msm_fold_check(env, run)

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    W = WORKLOADS[wl]
    units = n * world * args.steps
    value = units / dt
    roofline = roofline_for(env, run, kern_avg_s)

    cpu, checked = None, None
    sample = args.cpu_sample
    baseline_leg = world == 1 or sample > 0         # the CPU baseline is reported at N = 1 (or when asked for)
    if sample < 0:
        from oracle import zc_ref as _z
        sample = default_sample(wl, world, _z.host_threads())
    if sample:
        checked, cpu = check_and_baseline(env, run, sample, baseline_leg, msm_fold_ok)

    secondary = None
    if headline_default and world == 1 and not args.no_secondary and args.cpu_sample != 0:
        del run
        torch.cuda.empty_cache()
        secondary = secondary_runs(env)

    metric = {"scalar_mul": "%d-bit Edwards variable-base scalar-muls/sec (batched, %s)" % (
                  args.scalar_bits, "strict bit-exact mode" if args.mode == "strict" else "FAST non-strict mode"),
              "msm": "MSM point-scalar pairs/sec (bucket method per GPU, in-library RCCL all-gather + ordered fold across GPUs)",
              "ristretto": "Ristretto decompress -> scalar-mul -> compress round trips/sec (fused, bit-exact encodings)",
              "fe_mul": "FieldElement multiplications/sec (batched, bit-exact canonical limbs)",
              "fe_invert": "FieldElement inversions/sec (batched, bit-exact canonical limbs)",
              "ecdh": "ECDH exchanges/sec (two key generations + two shared secrets each; the reference's macro-benchmark)"}[wl]
    line = {
        "metric": metric, "value": round(value, 1), "unit": W["unit"],
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u64 (nine 29-bit limbs in u32 registers, 64-bit multiply-accumulate columns)", "data": "synthetic",
        "config": {"workload": workload_label(wl, n, args.scalar_bits, args.ecdh),
                   "units_per_gpu_per_step": n,
                   "sharding": "contiguous ranges; ncclAllGather of one 160-byte partial sum per rank inside libzerocaf_hip.so + ordered fold kernel"
                               if wl == "msm" else "contiguous ranges, no collective",
                   "mode": mode_label(wl, args.mode, args.ecdh)},
        "secondary_summary": None,
        "roofline": roofline,
        "cpu_baseline": cpu,
        "parity_spot_check": checked,
        "devices": idents,
        "distinct_devices": distinct,
        "kernel_avg_ms_ranks": {"min": min(d["kernel_avg_ms"] for d in idents), "max": max(d["kernel_avg_ms"] for d in idents)},
    }
    if wl == "msm":
        line["rccl_ranks"] = rccl_ranks                      # ncclCommCount of the library's own communicator (None: gloo test hook)
        line["msm_result_is_fold_of_shard_partials"] = msm_fold_ok
    if secondary is not None:
        # every config's time and roofline fraction where the driver's record keeps them: a compact block in front of
        # `roofline`, the same block inside it, and (last thing on stderr) one short line per config
        line["secondary_summary"] = {x["name"]: [x["ms_per_step"], x["roofline"].get("frac"), x["roofline"].get("bound")] for x in secondary}
        roofline["secondary"] = {x["name"]: {"ms_per_step": x["ms_per_step"], "value": x["value"], "unit": x["unit"], "frac": x["roofline"].get("frac"),
                                             "bound": x["roofline"].get("bound"), "parity": x["parity_spot_check"]} for x in secondary}
        line["secondary"] = secondary
    else:
        del line["secondary_summary"]
    os.write(json_fd, (json.dumps(line) + "\n").encode())
    for x in secondary or []:
        log(summary_line(x["name"], x))
    if world > 1:
        dist.destroy_process_group()


def _popcount64(x):
    x = x.astype(np.uint64)
    m1, m2, m4 = np.uint64(0x5555555555555555), np.uint64(0x3333333333333333), np.uint64(0x0F0F0F0F0F0F0F0F)
    x = x - ((x >> np.uint64(1)) & m1)
    x = (x & m2) + ((x >> np.uint64(2)) & m2)
    x = (x + (x >> np.uint64(4))) & m4
    return ((x * np.uint64(0x0101010101010101)) >> np.uint64(56)).astype(np.int64)


if __name__ == "__main__":
    main()
