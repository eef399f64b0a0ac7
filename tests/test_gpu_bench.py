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
    # the other BASELINE configurations ride in the same driver-run line
    w = d["whole_step"]
    assert w["whole_step_ms"] > w["solver_device_ms"] > 0 and w["mean_active_constraints"] > 59000
    # (two separately timed loops: the one with the pair query is the slower by ~0.08 ms, run-to-run noise is ~0.01 ms)
    assert w["whole_step_with_pair_query_ms"] > 0.9 * w["whole_step_ms"] and w["pair_queries"] > 0
    isl = d["island_sharded"]
    assert isl["scaling"] == "strong" and isl["n_gpus"] == 1 and isl["config"]["constraints"] == 1218560
    r5 = isl["roofline"]
    # (`frac` is over the launch's MINIMUM traffic -- 184 B per constraint + 124 B per body, once per step -- and so at most 1; the contract's
    # per-sweep byte model is kept beside it)
    assert r5["bound"] == "hbm" and abs(r5["algorithmic_bytes_per_launch"] - (184.0 * 1218560 + 124.0 * 420352)) < 1 and 0 < r5["frac"] <= 1.0
    assert abs(r5["contract_model_8d"]["bytes_per_launch"] - 136.0 * 1218560 * 16) < 1
    assert d["configs"]["4_joint_grid"]["unit"] == "joint-iters/s" and d["configs"]["4_joint_grid"]["value"] > 0
    assert d["configs"]["3_tumbler"]["unit"] == "constraint-iters/s" and d["configs"]["3_tumbler"]["value"] > 0
    assert d["configs"]["3_tumbler"]["whole_loop_ms_per_step_tgs_soft"] > 0
    # SURVEY.md 8f row 4 as a number: the headline world while balls plough through it (joining bodies placed: nearly every step on the persistent kernel)
    ch = d["churn"]
    assert ch["steps"] == 240 and ch["steps_with_created_or_destroyed_contacts"] > 60 and ch["steps_on_persistent_kernel"] >= 225, ch
    assert ch["steps_that_built_a_structure"] <= 10 and 0 < ch["churn_step_median_ms"] < 2.0, ch


def test_two_ranks_on_one_device():
    env = dict(os.environ, S2AMD_BENCH_BACKEND="gloo", S2AMD_BENCH_SINGLE_DEVICE="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port",
           "29517", "bench.py", "--gpus", "2", "--steps", "20", "--warmup", "5", "--no-cpu", "--base", "100", "--no-extras"]
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
           "29519", "bench.py", "--gpus", "1", "--steps", "50", "--warmup", "10", "--no-cpu", "--no-extras"]
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    d = last_json(out.stdout)
    assert d["n_gpus"] == 1 and d["config"]["kernel_launches_per_step"] in (1, 3)  # (1: the self-contained strip step)
    # the collective and the hand-shakes must not serialise the steps: within 25 % of the plain single-process rate
    assert d["ms_per_step"] < 0.40, d["ms_per_step"]


def _unsharded_poses(islands, base, steps):
    from solver2d_amd import hip, synthetic, wire
    world = synthetic.pyramid(base, count=islands)
    params = wire.StepParams.make("TGS_Soft", 1.0 / 60.0, 8, 4, True)
    # (no torch here: its bundled HIP runtime cannot initialise after libs2amd.so's has taken the device in this process)
    with hip.Solver(0) as s:
        s.upload(*world)
        for _ in range(steps):
            s.step_resident(params)
        buf = s.device_alloc(len(world[0]) * 32)
        s.export_bodies_async(buf, len(world[0]), 0)
        s.export_wait(0)
        out = s.device_read(buf, (len(world[0]), 8))
        s.device_free(buf)
    return out


@pytest.mark.parametrize("nranks,port,islands", [(2, 29523, 37), (3, 29525, 37), (3, 29529, 2)])
def test_island_sharded_config5_equals_the_unsharded_world(tmp_path, nranks, port, islands):
    """bench.py --config 5 (BASELINE configs[4]) with 2 and 3 ranks sharing this box's one GPU over gloo: every rank's shard
    resident, the per-island body arrays (position, rot, linear and angular velocity: SURVEY.md 8e) all-gathered every step.  What
    rank 0 assembles from the gathered records after warmup + steps steps must equal, bit for bit, the bodies of the same world
    stepped unsharded in one solver.  (Two islands on
    three ranks: one rank owns nothing and still takes part in every all-gather.)"""
    import numpy as np
    dump = str(tmp_path / "poses.npy")
    env = dict(os.environ, S2AMD_BENCH_BACKEND="gloo", S2AMD_BENCH_SINGLE_DEVICE="1", MASTER_ADDR="127.0.0.1", S2AMD_BENCH_DUMP=dump)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nranks), "--master-addr", "127.0.0.1", "--master-port",
           str(port), "bench.py", "--gpus", str(nranks), "--config", "5", "--islands", str(islands), "--island-base", "14", "--steps", "12", "--warmup", "3"]
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    d = last_json(out.stdout)
    for k in REQUIRED:
        assert k in d, k
    assert d["n_gpus"] == nranks and d["scaling"] == "strong" and d["config"]["constraints"] == islands * 287
    assert abs(d["value"] - islands * 287 * 16 * 12 / (d["ms_per_step"] * 12e-3)) / d["value"] < 1e-6
    got = np.load(dump)
    want = _unsharded_poses(islands, 14, 15)
    assert got.shape == want.shape and np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_island_sharded_one_rank_through_rccl():
    """The sharded driver's loop with its real backend (one rank over RCCL): device-tensor all-gather, event hand-shakes."""
    env = dict(os.environ, S2AMD_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port",
           "29527", "bench.py", "--gpus", "1", "--config", "5", "--steps", "30", "--warmup", "5"]
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    d = last_json(out.stdout)
    assert d["n_gpus"] == 1 and d["config"]["constraints"] == 1218560 and d["config"]["lds_groups_this_rank"] == 512
    assert d["ms_per_step"] < 1.5, d["ms_per_step"]
