"""GPU tier: bench.py's output contract -- exactly ONE JSON line on stdout with the fields the driver
reads, `roofline` and `cpu_baseline` objects, the parity check passed -- for one rank, and for two ranks
launched the way the driver launches them (torch.distributed.run, 127.0.0.1).  A one-GPU box cannot give
each rank its own device, so the two-rank runs use bench.py's test hooks: both ranks on device 0 and
gloo for the barrier / MAX-reduce (RCCL refuses two ranks on one device)."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _one_json_line(out):
    lines = [l for l in out.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert all(k in d for k in REQUIRED), [k for k in REQUIRED if k not in d]
    return d


def test_spot_check_sample_covers_head_tail_and_a_stride_through_the_middle():
    """bench.py's oracle spot check (and CPU baseline sample) is spread over the whole batch: the first quarter of the
    sample from the head, the last quarter from the tail -- the ragged last workgroup, the end of a persistent kernel's
    schedule -- and a seeded jittered stride through everything between them.  Deterministic, sorted, no duplicates, the
    unit count asked for; the middle visits every residue modulo the wave / workgroup / tile sizes."""
    import numpy as np
    sys.path.insert(0, ROOT)
    import bench
    for n, m in ((1 << 20, 1 << 17), (1 << 24, 1 << 18), (1 << 22, 2048), ((1 << 21), 4096), (300000, 4096), (1000, 64), (7, 3)):
        idx = bench.spread_sample(n, m)
        assert len(idx) == m and idx.dtype == np.int64 and np.array_equal(idx, bench.spread_sample(n, m))
        assert np.all(np.diff(idx) > 0) and 0 <= idx[0] and idx[-1] < n and (m < 4 or (idx[0] == 0 and idx[-1] == n - 1))
        h = m // 4
        assert np.array_equal(idx[:h], np.arange(h)) and np.array_equal(idx[m - h:], np.arange(n - h, n))
        mid = idx[h:m - h]
        step = (n - 2 * h) // len(mid)
        assert np.all(mid >= h) and np.all(mid < n - h)
        assert np.array_equal((mid - h) // step, np.arange(len(mid)))         # one unit out of every `step` consecutive ones
        if len(mid) >= 4096 and step > 1:
            for mod in (64, 256, 4096):
                assert len(np.unique(mid % mod)) == mod, (n, m, mod)   # every residue occurs
        # the eighths of the batch are all visited
        if m >= 64:
            assert len(np.unique(idx * 8 // n)) == 8
    assert np.array_equal(bench.spread_sample(100, 100), np.arange(100)) and np.array_equal(bench.spread_sample(100, 1000), np.arange(100))


@pytest.mark.gpu
def test_single_rank_line():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--cpu-sample", "4096"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    d = _one_json_line(out.stdout)
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["parity_spot_check"] is True and d["vs_baseline"] is None
    r = d["roofline"]
    assert r["bound"] == "valu_int_mul" and {"achieved", "peak", "unit", "frac", "traffic"} <= set(r)
    assert 0 < r["frac"] < 1 and r["useful"]["formula_evaluations_per_unit"] > 300          # frac = useful multiplications only
    assert r["measured_rate"]["v_mad_u64_u32_T_lane_ops_per_s"] > 10 and 0 < r["measured_rate"]["frac"] < 1.05
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c and c["value_single_core"] > 0
    assert "spread over the whole batch" in c["sample"] and "head" in c["sample"] and "tail" in c["sample"] and "stride" in c["sample"]
    assert d["distinct_devices"] == 1 and len(d["devices"]) == 1 and d["devices"][0]["pci"]
    assert d["config"]["units_per_gpu_per_step"] == 1 << 20 and abs(d["value"] - (1 << 20) / (d["ms_per_step"] * 1e-3)) < 0.01 * d["value"]
    assert d["config"]["workload"].startswith("2^20 ") and r["schema"] == "useful-work/2" and "-march=" in c["sample"]
    # the default single-GPU line carries the other BASELINE configs (and the reference's ECDH macro-benchmark) as `secondary`:
    # driver-box numbers for every config, each with its own roofline record and oracle spot check, outside the headline's timing
    sec = {x["name"]: x for x in d["secondary"]}
    assert list(sec) == ["fe_mul", "fe_invert", "ristretto", "msm", "ecdh", "scalar_mul_s249"]
    assert [sec[k]["units"] for k in sec] == [1 << 24, 1 << 20, 1 << 22, 1 << 21, 1 << 20, 1 << 20]
    for k, x in sec.items():
        lg = x["units"].bit_length() - 1
        assert x["parity_spot_check"] is True and x["workload"].startswith("2^%d" % lg), k
        assert x["ms_per_step"] > 0 and abs(x["value"] - x["units"] / (x["ms_per_step"] * 1e-3)) < 0.01 * x["value"], k
        assert {"bound", "frac", "achieved", "peak", "kernel_avg_ms"} <= set(x["roofline"]) and 0 < x["roofline"]["frac"] < 1.2, k
    assert sec["fe_mul"]["roofline"]["bound"] == "hbm" and sec["fe_mul"]["roofline"]["cache_resident"] is False
    assert sec["msm"]["msm_result_is_fold_of_shard_partials"] is True and sec["msm"]["rccl_ranks"] == 1
    assert sec["msm"]["roofline"]["useful"]["plan"]["window_groups"] == 3 and sec["msm"]["roofline"]["useful"]["multiplications_per_bucket_addition"] == 7
    assert sec["ecdh"]["roofline"]["bound"] == "valu_int_mul"
    # the reference's own Scalar::random domain next to the 252-bit headline: fewer formula evaluations per unit, same kernel
    s249 = sec["scalar_mul_s249"]["roofline"]
    assert 365 < s249["useful"]["formula_evaluations_per_unit"] < r["useful"]["formula_evaluations_per_unit"] and "issued" not in s249
    # where the driver's record keeps the other configs: a compact block in FRONT of `roofline` (its fixed keys and a
    # 2000-character tail survive), the same numbers inside `roofline`, and one short stderr line per config, printed last
    keys = list(d)
    assert keys.index("secondary_summary") < keys.index("roofline") < keys.index("secondary")
    assert list(d["secondary_summary"]) == list(sec) == list(r["secondary"])
    for k, (ms, frac, bound) in d["secondary_summary"].items():
        assert ms == sec[k]["ms_per_step"] and frac == sec[k]["roofline"]["frac"] and bound == sec[k]["roofline"]["bound"], k
        assert r["secondary"][k]["frac"] == frac and r["secondary"][k]["parity"] is True
    # (RCCL's banner, buffered on fd 1, is flushed behind them at exit: five short lines)
    assert all("secondary" in l or "version" in l or l.startswith(("Hostname", "Librccl")) for l in out.stderr.splitlines()[-(len(sec) + 5):] if l.strip())
    tail = [l for l in out.stderr.splitlines() if l.startswith("secondary ") and "ms/step" in l]
    assert len(tail) == len(sec) and len("\n".join(tail)) < 1200 and len(out.stderr[out.stderr.index(tail[0]):]) < 1900, tail
    for k, l in zip(sec, tail):
        assert l.startswith("secondary %s: " % k) and "frac %s of %s roof" % (sec[k]["roofline"]["frac"], sec[k]["roofline"]["bound"]) in l and l.endswith("parity True"), l


@pytest.mark.gpu
def test_workload_string_names_the_size_it_ran_at():
    """`config.workload` is derived from --units (a record of a 2^18-unit run must not say 2^20); other sizes carry no `secondary`."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--units", str(1 << 18), "--cpu-sample", "2048"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    d = _one_json_line(out.stdout)
    assert d["config"]["workload"].startswith("2^18 ") and d["config"]["units_per_gpu_per_step"] == 1 << 18 and "secondary" not in d
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--units", "300000", "--workload", "fe_mul", "--cpu-sample", "4096"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    assert _one_json_line(out.stdout)["config"]["workload"].startswith("300000 ")


@pytest.mark.gpu
@pytest.mark.parametrize("variant", ["wire", "reference"])
def test_ecdh_workload(variant):
    """The reference's macro-benchmark (benchmarks/dusk_benchmarks.rs:544-620): two key generations and two shared secrets per
    unit.  `wire`: 32-byte Ristretto encodings, equal to compress() of the oracle's ecdh_double_add key pairs and secrets;
    `reference`: the four double_and_add calls literally, limb-exact.  S == S' on every element either way (bench.py aborts otherwise)."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "ecdh", "--ecdh", variant, "--units", str(1 << 16), "--steps", "2",
                          "--warmup", "1", "--cpu-sample", "1024"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    d = _one_json_line(out.stdout)
    assert d["parity_spot_check"] is True and d["unit"] == "exchanges/s" and d["config"]["workload"].startswith("2^16 ECDH")
    assert 0 < d["roofline"]["frac"] < 1 and d["cpu_baseline"]["value"] > 0


@pytest.mark.gpu
@pytest.mark.parametrize("workload,units", [("scalar_mul", 1 << 18), ("msm", 1 << 16)])
def test_two_ranks_launched_like_the_driver(workload, units):
    env = dict(os.environ, ZC_BENCH_BACKEND="gloo", ZC_BENCH_DEVICE="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--workload", workload, "--units", str(units)]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    d = _one_json_line(out.stdout)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["parity_spot_check"] is True
    assert d["cpu_baseline"] is None                                   # the CPU leg is reported at N = 1 only
    # whole-job aggregate: both ranks' units over the slowest rank's time
    assert abs(d["value"] - 2 * units / (d["ms_per_step"] * 1e-3)) < 0.01 * d["value"]
    # the self-proving fields of a scaling record: one identity per rank (both on the one device under the test hook),
    # every rank's own kernel time, and for the MSM the fold check and the communicator's rank count (None over gloo)
    assert [x["rank"] for x in d["devices"]] == [0, 1] and all(x["uuid"] or x["pci"] for x in d["devices"])
    assert d["distinct_devices"] == 1 and d["kernel_avg_ms_ranks"]["min"] <= d["kernel_avg_ms_ranks"]["max"]
    if workload == "msm":
        assert d["msm_result_is_fold_of_shard_partials"] is True and "rccl_ranks" in d


@pytest.mark.gpu
def test_plain_python_launch_starts_the_ranks_itself():
    """The driver's observed invocation is plain `python3 bench.py --gpus N ...` (BENCH_r04.json.cmd): with no launcher
    environment bench.py starts the N ranks itself and still prints ONE line with n_gpus = N (one-device test hooks here)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    # a rendezvous left over in the caller's environment is not trusted (an unroutable address, a privileged port): the ranks get
    # 127.0.0.1 and a free port unless ZC_BENCH_MASTER_ADDR / ZC_BENCH_MASTER_PORT say otherwise
    env.update(ZC_BENCH_BACKEND="gloo", ZC_BENCH_DEVICE="0", MASTER_ADDR="203.0.113.1", MASTER_PORT="1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--units", str(1 << 18)],
                         capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    d = _one_json_line(out.stdout)
    assert d["n_gpus"] == 2 and [x["rank"] for x in d["devices"]] == [0, 1] and d["parity_spot_check"] is True
    assert abs(d["value"] - 2 * (1 << 18) / (d["ms_per_step"] * 1e-3)) < 0.01 * d["value"]


@pytest.mark.gpu
def test_more_gpus_than_devices_fails_closed():
    """`--gpus N` on a box with fewer than N devices must not come back as a smaller measurement: non-zero exit, nothing on stdout."""
    import torch
    n = torch.cuda.device_count() + 1
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "ZC_BENCH_DEVICE", "ZC_BENCH_BACKEND")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode != 0 and out.stdout.strip() == "" and "visible device" in out.stderr, (out.returncode, out.stdout[-300:], out.stderr[-600:])
    # under a launcher: as many ranks as the launcher started, or an error -- never "using WORLD_SIZE"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT, env=dict(env, RANK="0", WORLD_SIZE="1"))
    assert out.returncode != 0 and out.stdout.strip() == "" and "WORLD_SIZE" in out.stderr


def test_gpus_flag_fails_closed_without_a_gpu():
    """CPU tier: the argument checks come before anything touches a device."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "ZC_BENCH_DEVICE", "ZC_BENCH_BACKEND")}
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        pytest.skip("two devices: --gpus 2 is a valid request here")
    run = lambda a, e: subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + a, capture_output=True, text=True, timeout=300, cwd=ROOT, env=e)
    out = run(["--gpus", "2"], env)
    assert out.returncode != 0 and out.stdout == "" and "visible device" in out.stderr, out.stderr[-500:]
    out = run(["--gpus", "1"], dict(env, RANK="0", WORLD_SIZE="2"))
    assert out.returncode != 0 and out.stdout == "" and "WORLD_SIZE is 2" in out.stderr, out.stderr[-500:]
    out = run(["--gpus", "0"], env)
    assert out.returncode != 0 and out.stdout == ""


@pytest.mark.gpu
def test_msm_line_reports_the_library_communicator():
    """One rank, the real in-library RCCL path: ncclCommCount of the context's communicator is on the line."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "msm", "--units", str(1 << 16), "--steps", "2",
                          "--warmup", "1", "--cpu-sample", "2048"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    d = _one_json_line(out.stdout)
    assert d["rccl_ranks"] == 1 and d["msm_result_is_fold_of_shard_partials"] is True and d["parity_spot_check"] is True
    # 2^16 pairs lie below the affine threshold: projective 128-byte records, 8 multiplications per bucket addition (the library's own plan)
    u = d["roofline"]["useful"]
    assert u["multiplications_per_bucket_addition"] == 8 and u["plan"]["affine"] is False and u["plan"]["record_bytes"] == 128 and 0 < d["roofline"]["frac"] < 1
