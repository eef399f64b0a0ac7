#!/usr/bin/env python3
"""Headline benchmark: contact-constraints x iterations / sec, LargePyramid base-200, TGS_Soft
(8 sub-steps, relax on) -- BASELINE.json's metric on BASELINE.json's config (configs[1]).

    python bench.py --gpus N --steps K --warmup W

A "step" is one s2Solve_TGS_Soft (the hot path of one s2World_Step) over one resident snapshot:
the pyramid's step-0 solver input (solver2d_amd.synthetic.pyramid == the reference's own captured
input, tests/test_synthetic.py), body state restored from the snapshot before every step, contact
impulses carried from step to step (warm starting).  Inputs are in HBM before the timed region.
`value` = C x solve_sweeps x K x N / wall seconds, with solve sweeps counted as executed
(TGS_Soft 8/4: 16 per step, reference src/solve_tgs_soft.c:211-269).

N > 1: one process per GPU (torch.distributed / RCCL), each rank owns one independent pyramid
island (weak scaling); the only exchange is the per-step all-gather of per-island body poses.

Extra objects on the JSON line:
  roofline      dominant kernel: stripStepKernel<SOFT_TGS> (the whole step in one persistent launch) when the strip
                structure is in use, else solveContactsSoftKernel<SOFT_TGS> (one colour batch).  Algorithmic bytes
                per launch (232 B/constraint-sweep, SURVEY.md 8d x the constraint-sweeps one launch processes) /
                average duration of one launch in steady state (launches enqueued back to back in a hipGraph and
                bracketed by HIP events on the launch stream; see s2amd_measure_dominant); traffic: the PMC passes
                under profiles/.
  cpu_baseline  the reference's own s2Solve_TGS_Soft timed on this host (oracle/_ref, kind
                "reference") or, if that library is absent, the oracle port; 1 core.
  whole_step    (N = 1) the SURVEY.md 8d trajectory of config 2: the base-200 WORLD (shapes, pair states) resident, 60 settle
                steps then 240 timed steps of s2amd_world_step (stage 3 narrow phase -> s2Solve_TGS_Soft -> stage 4 refit):
                whole-step ms and the solver's share of it.
  configs       (N = 1) BASELINE.json configs[3] and configs[4] on this one GPU: JointGrid 100x100 / PGS_NGS and
                512 x base-40 / TGS_Soft, each with its own unit, ms per step and roofline object.
  island_sharded  BASELINE.json configs[4] as the N-GPU job it names: ONE world of 512 base-40 pyramids,
                its islands bin-packed onto the N ranks, every rank's shard resident in its own HBM, one all-gather of
                device-resident pose records per step (solver2d_amd/distributed.py: ResidentShardedWorld).  STRONG scaling:
                `value` there = the whole world's constraint-sweeps / max-over-ranks time.  `--config 5` makes this the
                line itself.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

from solver2d_amd import hip, synthetic, wire  # noqa: E402

ALGO_BYTES_PER_CONSTRAINT_SWEEP = 232.0
HBM_PEAK_GBS = 8000.0                    # MI355X_MICROARCH.md: 8.0 TB/s spec


def cpu_baseline(base, vel, pos, budget_s):
    """Reference (or port) solver-only throughput on this host, single thread, bounded sample."""
    C = None
    try:
        from tests import refbind
        if refbind.available():
            L = refbind.lib()
            with refbind.RefWorld("pyramid", "TGS_Soft", base, 0) as w:
                w.step(1.0 / 60.0, vel, pos, True)  # creates the contacts
                L.s2ref_set_mode(3)
                L.s2ref_solve_seconds(1)
                t0 = time.time()
                steps = 0
                while steps < 3 or (time.time() - t0 < budget_s and steps < 400):
                    w.step(1.0 / 60.0, vel, pos, True)
                    steps += 1
                secs = L.s2ref_solve_seconds(1)
                L.s2ref_set_mode(0)
                _b, c, _j = w.pack()
                C = int((c["pointCount"] > 0).sum())
            sweeps = wire.solve_sweeps_per_step("TGS_Soft", vel, pos)
            return {"value": C * sweeps * steps / secs, "unit": "constraint-iters/s", "cores": 1, "kind": "reference",
                    "sample": "%d s2World_Step of pyramid base-%d, time inside the reference's s2Solve_TGS_Soft only "
                              "(%.1f ms/solve); host has %d cores" % (steps, base, 1e3 * secs / steps, os.cpu_count())}
    except Exception as e:  # fall through to the port
        sys.stderr.write("cpu_baseline: reference unavailable (%r), timing the oracle port\n" % (e,))
    from tests import oraclebind
    pre = synthetic.pyramid(base)
    params = wire.StepParams.make("TGS_Soft", 1.0 / 60.0, vel, pos, True)
    C = int((pre[1]["pointCount"] > 0).sum())
    bodies0 = pre[0].copy()
    steps, t0, secs = 0, time.time(), 0.0
    while steps < 3 or (time.time() - t0 < budget_s and steps < 400):
        pre[0][:] = bodies0
        t1 = time.perf_counter()
        oraclebind.solve(params, *pre)
        secs += time.perf_counter() - t1
        steps += 1
    sweeps = wire.solve_sweeps_per_step("TGS_Soft", vel, pos)
    return {"value": C * sweeps * steps / secs, "unit": "constraint-iters/s", "cores": 1, "kind": "port",
            "sample": "%d oracle solves of the base-%d snapshot (%.1f ms/solve); host has %d cores" % (
                steps, base, 1e3 * secs / steps, os.cpu_count())}


def pmc_traffic_bytes(kernel_prefix):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC summaries (separate
    FETCH_SIZE / WRITE_SIZE passes of this same command, profiles/r01_persistent_pmc_*.txt): counters are in KiB;
    FETCH_SIZE is doubled per the gfx950 note in MI355X_MICROARCH.md (it reads half of a wide coalesced stream).
    None when the summaries are not there."""
    for rnd in ("r02", "r01"):
        total = 0.0
        try:
            for name, scale in (("fetch", 2.0), ("write", 1.0)):
                path = os.path.join(ROOT, "profiles", "%s_persistent_pmc_%s_size.txt" % (rnd, name))
                rows = [l.split() for l in open(path) if l.startswith(kernel_prefix) and ("FETCH_SIZE" in l or "WRITE_SIZE" in l)]
                total += scale * float(rows[0][3]) * 1024.0
            return total, "profiles/%s_persistent_pmc_{fetch,write}_size.txt (rocprofv3 --pmc, separate passes of this command; not measured in this run)" % rnd
        except (OSError, IndexError, ValueError):
            continue
    return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=60)
    ap.add_argument("--base", type=int, default=200, help="pyramid base count (200 = BASELINE config 2)")
    ap.add_argument("--vel-iters", type=int, default=8)
    ap.add_argument("--pos-iters", type=int, default=4)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="time budget of the CPU baseline sample")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--opt", action="append", default=[], metavar="KEY=VALUE", help="s2amd_set_option passthrough (experiments)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    # test hooks (tests/test_gpu_bench.py): run the N > 1 logic on a box with ONE GPU over gloo
    backend = os.environ.get("S2AMD_BENCH_BACKEND", "nccl")
    device_index = 0 if os.environ.get("S2AMD_BENCH_SINGLE_DEVICE") == "1" else local_rank
    # S2AMD_BENCH_FORCE_DIST=1 (tests): take the distributed path even with ONE rank, so that the RCCL calls of the N > 1
    # loop can be exercised on a box with a single GPU
    distributed = world > 1 or os.environ.get("S2AMD_BENCH_FORCE_DIST") == "1"
    if distributed:
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(device_index)
        dist.init_process_group(backend=backend)
    n_gpus = max(args.gpus, 1)
    if world > 1 and world != n_gpus:
        n_gpus = world

    pre = synthetic.pyramid(args.base)
    params = wire.StepParams.make("TGS_Soft", 1.0 / 60.0, args.vel_iters, args.pos_iters, True)
    sweeps = wire.solve_sweeps_per_step("TGS_Soft", args.vel_iters, args.pos_iters)

    gpu = hip.Solver(device_index if distributed else 0, graph=not args.no_graph)
    for kv in args.opt:
        key, _, val = kv.partition("=")
        gpu.set_option(key, int(val))
    gpu.set_option("async", 1)  # steps are enqueued back to back; sync() below waits for the solver's stream
    gpu.upload(*pre)
    gpu.save_bodies()

    pose = None
    gathered = None
    torch = None
    if distributed:
        import torch
        nb = len(pre[0])
        # two pose buffers: the all-gather of step s reads one while step s+1 exports into the other
        pose = [torch.zeros((nb, 4), dtype=torch.float32, device="cuda") for _ in range(2)]
        gdev = "cuda" if backend == "nccl" else "cpu"
        gathered = torch.zeros((world * nb, 4), dtype=torch.float32, device=gdev)
        gather_done = [None, None]

    def enqueue(step):
        """restore + s2Solve + pose export of `step` on the solver's stream; nothing here waits for the device
        except for the collective that last read this step's pose buffer (two steps ago)."""
        gpu.restore_bodies()
        gpu.step_resident(params)
        if distributed:
            b = step & 1
            if gather_done[b] is not None:
                gather_done[b].synchronize()
            gpu.export_poses_async(pose[b].data_ptr(), nb, b)

    def exchange(step):
        """per-step exchange of the per-island body poses (RCCL all-gather), once this step's export has landed"""
        b = step & 1
        gpu.export_wait(b)
        dist.all_gather_into_tensor(gathered, pose[b] if backend == "nccl" else pose[b].cpu())
        if backend == "nccl":
            gather_done[b] = torch.cuda.Event()
            gather_done[b].record()

    def run(count):
        """`count` steps; with more than one rank the host enqueues step s+1 BEFORE it waits for the poses of step s, so
        the device goes from one solve straight into the next while the collective of the previous step is in flight."""
        if not distributed:
            for s_ in range(count):
                enqueue(s_)
            return
        enqueue(0)
        for s_ in range(count):
            if s_ + 1 < count:
                enqueue(s_ + 1)
            exchange(s_)

    def sync():
        gpu.synchronize()
        if distributed:
            import torch
            dist.barrier()
            torch.cuda.synchronize()

    run(args.warmup)
    sync()
    t0 = time.perf_counter()
    run(args.steps)
    sync()
    elapsed = time.perf_counter() - t0
    gpu.set_option("async", 0)
    gpu.restore_bodies()
    gpu.step_resident(params)  # one synchronous step: device time and counters for the report
    st = gpu.stats()
    C = st["constraintCount"]

    if distributed:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # dominant-kernel timing, live, with HIP events on the solver's own stream: the solve sweep's
    # colour-batch launches enqueued back to back (hipGraph) and bracketed by one event pair
    avg_launch_us, launches_per_sweep, constraints_per_launch = gpu.measure_dominant(params, repeats=40)
    avg_launch_s = max(avg_launch_us * 1e-6, 1e-12)
    achieved = ALGO_BYTES_PER_CONSTRAINT_SWEEP * constraints_per_launch / avg_launch_s / 1e9
    persistent = bool(gpu.stats().get("persistent", 0))
    # per-launch event pairs (eager launches), kept as a cross-check against rocprofv3's per-kernel durations
    prof_steps = 3
    gpu.set_option("profile", 1)
    kernel_ms, launches, overhead_ms = 0.0, 0, 0.0
    for _ in range(prof_steps):
        gpu.restore_bodies()
        gpu.step_resident(params)
        s2 = gpu.stats()
        kernel_ms += s2["solveKernelMs"]
        launches += s2["solveLaunches"]
        overhead_ms = s2["eventPairOverheadMs"]
    gpu.set_option("profile", 0)

    if rank == 0:
        value = C * sweeps * args.steps * n_gpus / elapsed
        out = {
            "metric": "contact-constraints x iters/sec, LargePyramid base-%d TGS_Soft" % args.base,
            "value": value,
            "unit": "constraint-iters/s",
            "n_gpus": n_gpus,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "LargePyramid base-%d (%d bodies, %d two-point contact constraints), s2_solverTGS_Soft, "
                            "s2World_Step(dt=1/60, velIters=%d, posIters=%d, warmStart) => %d solve sweeps + %d warm-start "
                            "sweeps per step; one independent pyramid per GPU" % (
                                args.base, len(pre[0]), C, args.vel_iters, args.pos_iters, sweeps, args.vel_iters),
                "constraints": C, "solve_sweeps_per_step": sweeps, "contact_colors": st["contactColors"],
                "kernel_launches_per_step": st["kernelLaunches"], "graph_replay": bool(st["graphReplayed"]),
                "device_ms_per_step": st["deviceMs"],
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": pmc_traffic_bytes("_Z15stripStepKernel") if persistent and args.base == 200 else None,
                "kernel": "stripStepKernel<SOFT_TGS> (whole step, one persistent launch; constraints_per_launch counts "
                          "constraint-sweeps)" if persistent else "solveContactsSoftKernel<SOFT_TGS> / stripSoftKernel<SOFT_TGS>",
                "avg_launch_us": avg_launch_us, "launches_per_step": 1 if persistent else launches_per_sweep * sweeps,
                "constraints_per_launch": constraints_per_launch,
                "eager_event_pair_us_per_launch": 1e3 * kernel_ms / max(launches, 1), "empty_event_pair_us": overhead_ms * 1e3,
                "algorithmic_bytes_per_launch": ALGO_BYTES_PER_CONSTRAINT_SWEEP * constraints_per_launch,
            },
        }
        if not args.no_cpu and world == 1:  # the contract: on rank 0 at N = 1 only
            out["cpu_baseline"] = cpu_baseline(args.base, args.vel_iters, args.pos_iters, args.cpu_seconds)
        print(json.dumps(out))
    gpu.close()
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
