"""bench.py's contract, on the GPU box: one JSON line with the required keys at N = 1, and the N > 1 code path (one
process per rank, barrier + max-over-ranks timing, per-step all-gather of poses) with two ranks sharing the one GPU
of the test box over gloo (the driver runs the real thing over RCCL)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
            "data", "config", "roofline"]


def last_json(out):
    lines = [l for l in out.strip().splitlines() if l.startswith("{")]
    assert lines, out[-2000:]
    return json.loads(lines[-1])


def test_single_gpu_line():
    out = subprocess.run([sys.executable, "bench.py", "--steps", "30", "--warmup", "10", "--cpu-seconds", "2"], cwd=ROOT, capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    d = last_json(out.stdout)
    for k in REQUIRED + ["cpu_baseline"]:
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 30 and d["warmup"] == 10 and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["dtype"] == "f32" and d["unit"] == "constraint-iters/s" and "workload" in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert abs(d["value"] - d["config"]["constraints"] * d["config"]["solve_sweeps_per_step"] * 30 / (d["ms_per_step"] * 30e-3)) / d["value"] < 1e-6
    c = d["cpu_baseline"]
    assert c["cores"] == 1 and c["kind"] in ("reference", "port") and c["value"] > 0


def test_two_ranks_on_one_device():
    env = dict(os.environ, S2AMD_BENCH_BACKEND="gloo", S2AMD_BENCH_SINGLE_DEVICE="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port",
           "29517", "bench.py", "--gpus", "2", "--steps", "20", "--warmup", "5", "--no-cpu", "--base", "100"]
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    d = last_json(out.stdout)
    for k in REQUIRED:
        assert k in d, k
    assert d["n_gpus"] == 2 and d["scaling"] == "weak"
    # whole-job value: both ranks' constraints
    per_rank = d["config"]["constraints"] * d["config"]["solve_sweeps_per_step"] * 20 / (d["ms_per_step"] * 20e-3)
    assert abs(d["value"] - 2 * per_rank) / d["value"] < 1e-6


def test_one_rank_through_rccl():
    """The N > 1 loop with its real backend: one rank, backend nccl (= RCCL), so init_process_group, the asynchronous
    pose export, all_gather_into_tensor on device tensors and the event hand-shakes all run on this single-GPU box."""
    env = dict(os.environ, S2AMD_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port",
           "29519", "bench.py", "--gpus", "1", "--steps", "50", "--warmup", "10", "--no-cpu"]
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    d = last_json(out.stdout)
    assert d["n_gpus"] == 1 and d["config"]["kernel_launches_per_step"] == 3
    # the collective and the hand-shakes must not serialise the steps: within 25 % of the plain single-process rate
    assert d["ms_per_step"] < 0.40, d["ms_per_step"]
