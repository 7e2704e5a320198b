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
    assert d["distinct_devices"] == 1 and len(d["devices"]) == 1 and d["devices"][0]["pci"]
    assert d["config"]["units_per_gpu_per_step"] == 1 << 20 and abs(d["value"] - (1 << 20) / (d["ms_per_step"] * 1e-3)) < 0.01 * d["value"]


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
def test_msm_line_reports_the_library_communicator():
    """One rank, the real in-library RCCL path: ncclCommCount of the context's communicator is on the line."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "msm", "--units", str(1 << 16), "--steps", "2",
                          "--warmup", "1", "--cpu-sample", "2048"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    d = _one_json_line(out.stdout)
    assert d["rccl_ranks"] == 1 and d["msm_result_is_fold_of_shard_partials"] is True and d["parity_spot_check"] is True
    assert d["roofline"]["useful"]["multiplications_per_bucket_addition"] == 7 and 0 < d["roofline"]["frac"] < 1
