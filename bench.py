#!/usr/bin/env python3
"""Headline benchmark: contact-constraints x iterations / sec, LargePyramid base-200, TGS_Soft
(8 sub-steps, relax on) -- BASELINE.json's metric on BASELINE.json's config (configs[1]).

    python bench.py --gpus N --steps K --warmup W

A "step" is one s2Solve_TGS_Soft (the hot path of one s2World_Step) over the resident world: the
pyramid's step-0 solver input (solver2d_amd.synthetic.pyramid == the reference's own captured
input, tests/test_synthetic.py), then CONSECUTIVE resident steps -- bodies and contact impulses
carried from step to step, nothing restored or copied inside the timed region (`--restore` brings
back round 3's form: the step-0 bodies copied back before every step).  Inputs are in HBM before
the timed region.
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
  issue         what actually bounds the dominant kernel: VALU issue fraction and occupied CUs from the committed SQ counter
                pass (profiles/rNN_*_pmc_sq.txt, separate rocprofv3 --pmc runs of this command) -- the kernels of this path
                are instruction-issue / latency bound, not bandwidth bound, and the line says so next to `roofline`.
  ranks_seen / devices   torch.distributed's world size and the PCI bus ids of the ranks' GPUs (all-gathered): `n_gpus` is
                what was really there.  `--gpus N` without a launcher starts the N ranks itself (torch.distributed.run).
  value_fast    (N = 1) the same headline loop on the tolerance-mode build libs2amd_fast.so (FMA contraction on; SURVEY.md 7: "report
                both"): value, ms per step, its own dominant-kernel time and roofline fraction, and the distance from the bit-exact
                build after 30 steps.  `value` / `roofline` on the line itself are always the bit-exact build's.
  sharded_abi   (N = 1) configs[4] through the C-ABI's own single-process sharding (s2amd_sharded_*): 1 / 2 / 4 logical shards of this GPU
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
PROFILE_ROUNDS = ("r06", "r05", "r04", "r03", "r02", "r01")  # committed rocprofv3 summaries, newest first
# A committed counter pass describes the build it profiled.  It is quoted on the line only when the kernel it names is the kernel this
# run launched AND its average duration there agrees with the live one within this fraction (counter passes run a few per cent slower
# than plain ones); otherwise `traffic` / `issue` are null with the reason beside them.
PMC_DURATION_TOLERANCE = 0.03


def pmc_same_build(pass_us, live_us):
    return live_us is None or (pass_us is not None and abs(pass_us - live_us) <= PMC_DURATION_TOLERANCE * live_us)


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


ISLAND_KERNELS = ("_Z16wideIslandKernel", "_Z16islandStepKernel")  # wide_kernel.hip (TGS_Soft), strip_kernel.hip (the general form)


def pmc_issue(kernel_prefixes, stem="persistent", live_us=None):
    """VALU-issue picture of the dominant kernel from the committed SQ pass (profiles/rNN_<stem>_pmc_sq.txt, rocprofv3 --pmc, its
    own run): VALU instructions per wave and the share of a SIMD's cycles in which it issues one -- a wave64 VALU instruction
    occupies its SIMD's issue for 4 cycles (tools/valu_bench.hip), a CU has 4 SIMDs, and the waves a CU hosts during the launch
    are the launch's waves over the CUs it occupies.  None when the summary is absent."""
    for rnd in PROFILE_ROUNDS:
        path = os.path.join(ROOT, "profiles", "%s_%s_pmc_sq.txt" % (rnd, stem))
        try:
            vals, calls, avg_ns = {}, None, None
            for l in open(path):
                f = l.split()
                if len(f) >= 4 and any(f[0].startswith(p) for p in kernel_prefixes):
                    if f[1].startswith("SQ_"):
                        vals.setdefault(f[1], (float(f[2]), float(f[3])))  # samples, average per sample
                    elif calls is None and f[1].isdigit():
                        calls, avg_ns = int(f[1]), float(f[2])  # the kernel table: calls, average duration
            if "SQ_INSTS_VALU" in vals and "SQ_WAVES" in vals and calls:
                if not pmc_same_build(avg_ns * 1e-3, live_us):
                    return {"refused": "profiles/%s_%s_pmc_sq.txt is of another build: %.1f us per launch there, %.1f us live (> %d %% apart)" % (
                        rnd, stem, avg_ns * 1e-3, live_us, round(100 * PMC_DURATION_TOLERANCE))}
                instances = vals["SQ_WAVES"][0] / calls  # counter instances sampled per dispatch
                waves = vals["SQ_WAVES"][1] * instances
                per_wave = vals["SQ_INSTS_VALU"][1] / max(vals["SQ_WAVES"][1], 1e-9)
                cycles = avg_ns * 2.4  # 2.4 GHz
                out = {"valu_insts_per_wave": per_wave, "waves_per_launch": waves, "kernel_us_in_that_pass": avg_ns * 1e-3,
                       "source": "profiles/%s_%s_pmc_sq.txt (rocprofv3 --pmc SQ_*, separate passes of this command; not measured in this run)" % (rnd, stem)}
                out["_cycles"] = cycles
                return out
        except (OSError, ValueError):
            continue
    return None


def finish_issue(issue, workgroups):
    """CUs occupied and the per-SIMD VALU issue share, once the number of workgroups of the launch is known."""
    if issue is None or "refused" in issue:
        return issue
    cus = min(max(int(workgroups), 1), 256)
    cycles = issue.pop("_cycles")
    waves_per_cu = issue["waves_per_launch"] / cus
    issue["workgroups"] = int(workgroups)
    issue["cus_occupied_of_256"] = cus
    issue["valu_issue_frac_per_simd"] = waves_per_cu * issue["valu_insts_per_wave"] * 4.0 / (4.0 * max(cycles, 1.0))
    return issue


def pmc_traffic_bytes(kernel_prefix, stem="persistent", live_us=None):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC summaries (separate
    FETCH_SIZE / WRITE_SIZE passes of this same command, profiles/r01_persistent_pmc_*.txt): counters are in KiB;
    FETCH_SIZE is doubled per the gfx950 note in MI355X_MICROARCH.md (it reads half of a wide coalesced stream).
    None when the summaries are not there."""
    prefixes = kernel_prefix if isinstance(kernel_prefix, (tuple, list)) else (kernel_prefix,)
    for rnd in PROFILE_ROUNDS:
        total = 0.0
        try:
            for name, scale in (("fetch", 2.0), ("write", 1.0)):
                path = os.path.join(ROOT, "profiles", "%s_%s_pmc_%s_size.txt" % (rnd, stem, name))
                lines = [l.split() for l in open(path) if l.startswith(tuple(prefixes))]
                rows = [f for f in lines if len(f) > 3 and f[1] in ("FETCH_SIZE", "WRITE_SIZE")]
                table = [f for f in lines if len(f) > 3 and f[1].isdigit()]  # the kernel table of the same pass: calls, avg_ns
                pass_us = float(table[0][2]) * 1e-3 if table else None
                if not pmc_same_build(pass_us, live_us):
                    return None, "refused: %s is of another build (%.1f us per launch there, %.1f us live, > %d %% apart)" % (
                        os.path.relpath(path, ROOT), pass_us or 0.0, live_us, round(100 * PMC_DURATION_TOLERANCE))
                total += scale * float(rows[0][3]) * 1024.0
            return total, "profiles/%s_%s_pmc_{fetch,write}_size.txt (rocprofv3 --pmc, separate passes of this command; not measured in this run)" % (rnd, stem)
        except (OSError, IndexError, ValueError):
            continue
    return None, None


JOINT_BYTES_PER_ITER = 216.0  # a revolute joint sweep: frame, masses, pivot mass, soft coefficients, impulses, limits (120 B) + 2 x (36 B read, 12 B written)


def step_roofline(algorithmic_bytes_per_step, seconds_per_step, model):
    """Roofline object of a multi-launch step: algorithmic bytes of the whole step over its wall time (no single dominant launch)."""
    achieved = algorithmic_bytes_per_step / max(seconds_per_step, 1e-12) / 1e9
    return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
            "algorithmic_bytes_per_step": algorithmic_bytes_per_step, "byte_model": model, "per": "step (all launches)"}


ALGO_BYTES_LDS_PATH = 136.0  # per constraint-sweep when the body state is served from LDS (the group kernel)


class Ranks:
    """What the legs need to know about the job: torch.distributed handle (None at N = 1), backend, this rank."""

    def __init__(self, dist, torch, backend, rank, world, device_index):
        self.dist, self.torch, self.backend, self.rank, self.world, self.device_index = dist, torch, backend, rank, world, device_index

    def barrier_sync(self, solver):
        solver.synchronize()
        if self.dist is not None:
            self.dist.barrier()
            self.torch.cuda.synchronize()

    def max_over_ranks(self, seconds):
        if self.dist is None:
            return seconds
        t = self.torch.tensor([seconds], dtype=self.torch.float64, device="cuda" if self.backend == "nccl" else "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())


def island_sharded_leg(ranks, islands, base, vel, pos, steps, warmup, graph=True, dump=None, weak=False):
    """BASELINE.json configs[4]: ONE world of `islands` base-`base` pyramids, islands sharded over the ranks, every shard
    resident, one all-gather of pose records per step (ResidentShardedWorld).  Strong scaling.
    weak: `islands` pyramids PER RANK instead (a world of islands x N, every rank building and owning its own -- the partition an
    island-sharded world of that size would have): the curve that can be linear, beside the strong one that cannot (a rank with
    fewer islands than CUs runs at one island's latency)."""
    from solver2d_amd import distributed as dsh
    world = synthetic.pyramid(base, count=islands)
    params = wire.StepParams.make("TGS_Soft", 1.0 / 60.0, vel, pos, True)
    sweeps = wire.solve_sweeps_per_step("TGS_Soft", vel, pos)
    C_total = int((world[1]["pointCount"] > 0).sum()) * (ranks.world if weak else 1)
    sw = dsh.ShardedWorld(*world, rank=0 if weak else ranks.rank, world_size=1 if weak else ranks.world)
    gpu = hip.Solver(ranks.device_index, graph=graph)
    gpu.set_option("async", 1)
    rs = dsh.ResidentShardedWorld(sw, gpu, ranks.torch, dist=ranks.dist, backend=ranks.backend, exchange_ranks=ranks.world, exchange_rank=ranks.rank)
    rs.run(params, warmup)
    ranks.barrier_sync(gpu)
    t0 = time.perf_counter()
    rs.run(params, steps)
    ranks.barrier_sync(gpu)
    elapsed = ranks.max_over_ranks(time.perf_counter() - t0)
    if dump and ranks.rank == 0:
        np.save(dump, rs.world_bodies())
    gpu.set_option("async", 0)
    gpu.step_resident(params)
    st = gpu.stats()
    us, _launches, c_launch = gpu.measure_dominant(params, repeats=10)
    rs.close()
    gpu.close()
    mine = int((sw.mine.contacts["pointCount"] > 0).sum())
    # the group kernel runs the WHOLE step of its islands in one launch: constraint-sweeps per launch = mine * sweeps
    algo = ALGO_BYTES_LDS_PATH * mine * sweeps
    model_gbs = algo / max(us * 1e-6, 1e-12) / 1e9
    # what the launch has to move at the very least: every wire contact in once per STEP (152 B) and its impulses out (16 B per point),
    # every body of the rank in (88 B) and its solver fields out (36 B) -- the records live in registers / LDS in between
    bodies_mine = int((sw.mine.bodies["type"] >= 0).sum())
    min_bytes = mine * (152.0 + 32.0) + bodies_mine * (88.0 + 36.0)
    achieved = min_bytes / max(us * 1e-6, 1e-12) / 1e9
    out_line = {
        "metric": "contact-constraints x iters/sec, %d independent base-%d pyramids TGS_Soft, islands sharded over the GPUs" % (islands * (ranks.world if weak else 1), base),
        "value": C_total * sweeps * steps / elapsed, "unit": "constraint-iters/s", "n_gpus": ranks.world, "steps": steps, "warmup": warmup,
        "ms_per_step": 1e3 * elapsed / steps, "scaling": "weak" if weak else "strong",
        "config": {"workload": "one world of %d base-%d pyramids (%d bodies, %d two-point constraints, %d islands), s2_solverTGS_Soft %d/%d; "
                               "islands %s %d rank(s), shards resident, one all-gather of the per-island body arrays {position, rot, v, w} per step "
                               "(%d bytes per rank)" % (islands * (ranks.world if weak else 1), base, len(world[0]) * (ranks.world if weak else 1), C_total,
                                                        islands * (ranks.world if weak else 1), vel, pos,
                                                        "of every rank built and owned by it (%d per rank):" % islands if weak else "bin-packed onto", ranks.world, rs.record * 32),
                   "constraints": C_total, "constraints_this_rank": mine, "islands_this_rank": int((sw.shard_of_island == ranks.rank).sum()),
                   "solve_sweeps_per_step": sweeps, "kernel_launches_per_step": st["kernelLaunches"], "lds_groups_this_rank": st["groupCount"],
                   "device_ms_per_step": st["deviceMs"], "graph_replay": bool(st["graphReplayed"]), "trajectory": "consecutive resident steps (no restore)"},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     # (the committed PMC passes are of the whole world on one GPU)
                     "traffic": pmc_traffic_bytes(ISLAND_KERNELS, "config5", live_us=us)[0] if ranks.world == 1 and islands == 512 and base == 40 else None,
                     "traffic_source": pmc_traffic_bytes(ISLAND_KERNELS, "config5", live_us=us)[1] if ranks.world == 1 and islands == 512 and base == 40 else None,
                     "kernel": "wideIslandKernel / islandStepKernel (whole step of this rank's islands in one launch: constraints resident in registers, bodies in "
                               "LDS, records prepared from and impulses stored to the wire contacts by the kernel itself)", "avg_launch_us": us,
                     "algorithmic_bytes_per_launch": min_bytes,
                     "byte_model": "the launch's MINIMUM traffic: %d constraints x (152 B wire record in + 2 x 16 B impulses out) + %d bodies x (88 B in + 36 B "
                                   "out), once per step -- the kernel keeps the records in registers and the bodies in LDS for all %d sweeps" % (mine, bodies_mine, sweeps),
                     # SURVEY.md 8(d)'s per-sweep model, kept as a labelled extra: it counts a record per SWEEP while the kernel reads it per
                     # STEP, so it is not a bandwidth and passes the peak once the kernel is fast (r5 reported it as `frac`: 1.003)
                     "contract_model_8d": {"bytes_per_launch": algo, "bytes_per_constraint_sweep": ALGO_BYTES_LDS_PATH, "gbs_if_it_were_traffic": model_gbs,
                                           "over_peak": model_gbs / HBM_PEAK_GBS},
                     "note": "`frac` = minimum bytes / kernel time / 8 TB/s (<= 1 by construction); `traffic` (PMC) over `algorithmic_bytes_per_launch` says how "
                             "much more than the minimum the kernel moves; the kernel is VALU-issue bound -- see `issue`"},
    }
    line = out_line
    traffic = line["roofline"]["traffic"]
    if traffic:
        line["roofline"]["traffic_gbs"] = traffic / max(us * 1e-6, 1e-12) / 1e9
        line["roofline"]["traffic_frac_of_peak"] = line["roofline"]["traffic_gbs"] / HBM_PEAK_GBS
        line["roofline"]["traffic_over_minimum"] = traffic / min_bytes
    line["issue"] = finish_issue(pmc_issue(ISLAND_KERNELS, "config5", live_us=us), st["groupCount"]) if ranks.world == 1 and islands == 512 and base == 40 else None
    if line["issue"] is not None and "refused" not in line["issue"]:
        line["issue"]["note"] = "512 workgroups of 512 threads on 256 CUs (two passes): the kernel is VALU-issue bound, not bandwidth bound"
        # what bounds this kernel, as a fraction: the share of a SIMD's cycles in which it issues a VALU instruction
        line["roofline"]["issue_frac"] = line["issue"]["valu_issue_frac_per_simd"]
        line["roofline"]["bound_in_fact"] = "VALU issue (see `issue`)"
    return line


def joint_grid_leg(device_index, steps, warmup):
    """BASELINE.json configs[3]: JointGrid 100x100 (19,800 revolute joints), s2_solverPGS_NGS 4/2, one GPU; the unit of work is
    a joint-iteration: joints x (velocity + position sweeps)."""
    pre = synthetic.joint_grid(100)
    params = wire.StepParams.make("PGS_NGS", 1.0 / 60.0, 4, 2, True)
    with hip.Solver(device_index) as gpu:
        gpu.set_option("async", 1)
        gpu.upload(*pre)
        gpu.save_bodies()
        for _ in range(warmup):
            gpu.restore_bodies()
            gpu.step_resident(params)
        gpu.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            gpu.restore_bodies()
            gpu.step_resident(params)
        gpu.synchronize()
        elapsed = time.perf_counter() - t0
        gpu.set_option("async", 0)
        gpu.restore_bodies()
        gpu.step_resident(params)
        st = gpu.stats()
    J = st["jointCount"]
    return {"workload": "JointGrid 100x100: %d bodies, %d revolute joints, s2_solverPGS_NGS 4/2" % (len(pre[0]), J), "unit": "joint-iters/s",
            "value": J * 6 * steps / elapsed, "ms_per_step": 1e3 * elapsed / steps, "steps": steps, "joint_colors": st["jointColors"],
            "kernel_launches_per_step": st["kernelLaunches"], "device_ms_per_step": st["deviceMs"],
            "roofline": step_roofline(JOINT_BYTES_PER_ITER * J * 6, elapsed / steps, "%d B per joint-iteration (joint record 120 B + two bodies 36 B read, 12 B "
                                      "written each) x %d joints x 6 sweeps" % (JOINT_BYTES_PER_ITER, J)),
            "path": "op interpreter on persistent strips (generic_kernel.hip), %d strips" % st["stripCount"] if st["persistent"] else "colour batches",
            "roofline_note": ("one persistent launch per step (+ prologue / epilogue): %d strips of the grid's BFS levels, 7 joint sweeps x (interior colours, "
                              "forward hand-off, seam colours, return hand-off) at ~1 us per phase -- bound by the instruction latency of one "
                              "revolute-joint solve per lane and colour, not by bandwidth" % st["stripCount"]) if st["persistent"] else
                             ("%d dependent launches of ~20k threads per step: launch-latency bound (a dependent launch costs ~2-3 us)" % st["kernelLaunches"])}


def tumbler_leg(device_index, count, settle, steps):
    """BASELINE.json configs[2]: the Tumbler (a motor-driven hollow drum, `count` boxes), s2_solverJacobi 4/2.  The scene is built
    as a resident world WITHOUT contacts and settled by the product itself: `settle` steps of the whole loop -- device pair
    query, the caller's contact creation, s2amd_world_step under TGS_Soft (the reference's Jacobi diverges on piles) -- with
    contacts created and destroyed by the hundred every step.  Then s2Solve_Jacobi is timed on that resident snapshot the way
    the headline times TGS_Soft (bodies restored before every solve)."""
    from tools import churn_bench
    world = synthetic.tumbler_world(count, spare_slots_per_box=24)
    keys = ("bodies", "contacts", "joints", "shapes", "pairs", "origins")
    soft = wire.StepParams.make("TGS_Soft", 1.0 / 60.0, 8, 4, True)
    free = sorted(np.flatnonzero(world["pairs"]["shapeA"] < 0).tolist(), reverse=True)
    with hip.Solver(device_index) as gpu:
        gpu.world_upload(*[world[k] for k in keys])
        moved, created, destroyed = 1, 0, 0
        t0 = time.perf_counter()
        for _ in range(settle):
            if moved > 0:
                new = gpu.world_find_pairs()
                if len(new):
                    gpu.world_set_contacts(*churn_bench.create_contacts(world, free, new))
                    created += len(new)
            info = gpu.world_step(soft)
            if info["separatedCount"] > 0:
                free.extend(gpu.world_separated(info["separatedCount"]).tolist())
                destroyed += info["separatedCount"]
            moved = info["movedCount"]
        loop_ms = 1e3 * (time.perf_counter() - t0) / settle
        res = gpu.world_download(*[world[k] for k in keys])
    bodies, contacts, joints = res[0], res[1], res[2]
    active = int((contacts["pointCount"] > 0).sum())
    jac = wire.StepParams.make("Jacobi", 1.0 / 60.0, 4, 2, True)
    with hip.Solver(device_index) as gpu:
        gpu.set_option("async", 1)
        gpu.upload(bodies, contacts, joints)
        gpu.save_bodies()
        for _ in range(10):
            gpu.restore_bodies()
            gpu.step_resident(jac)
        gpu.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            gpu.restore_bodies()
            gpu.step_resident(jac)
        gpu.synchronize()
        elapsed = time.perf_counter() - t0
        gpu.set_option("async", 0)
        gpu.restore_bodies()
        gpu.step_resident(jac)
        st = gpu.stats()
    return {"workload": "Tumbler, %d boxes: settled on the device by %d steps of the whole loop (TGS_Soft; %d contacts created, %d destroyed, %.2f ms per "
                        "loop step incl. pair query and the caller's Python pool), then s2_solverJacobi 4/2 on the resident snapshot: %d active constraints "
                        "of %d potential, the drum touches %d" % (count, settle, created, destroyed, loop_ms, active, st["potentialConstraints"],
                                                                 int(((contacts["bodyA"] == 1) | (contacts["bodyB"] == 1))[contacts["pointCount"] > 0].sum())),
            "unit": "constraint-iters/s", "value": active * 6 * steps / elapsed, "ms_per_step": 1e3 * elapsed / steps, "steps": steps,
            "kernel_launches_per_step": st["kernelLaunches"], "device_ms_per_step": st["deviceMs"], "whole_loop_ms_per_step_tgs_soft": loop_ms,
            "roofline": step_roofline(ALGO_BYTES_PER_CONSTRAINT_SWEEP * active * 6, elapsed / steps, "232 B per constraint-sweep (SURVEY.md 8d) x %d active "
                                      "constraints x 6 sweeps" % active),
            "roofline_note": "%d dependent launches per step over ~%d constraints: launch-latency bound" % (st["kernelLaunches"], active)}


def whole_step_leg(device_index, base, vel, pos, settle, steps):
    """SURVEY.md 8d, config 2 as a trajectory: the world resident (bodies, manifolds, shapes, pair states), `settle` steps,
    then `steps` timed s2amd_world_step calls = stage 3 (narrow phase on every pair) -> s2Solve_TGS_Soft -> stage 4 (refit).
    (`settle` = 200 since round 4: a new world's one-off search over strip widths -- seven more structure builds on a worker thread, asked
    for at step 32, judged at step 128 -- shares the device with the steps while it runs: 0.33 instead of 0.22 ms per step for those ~100
    steps, measured.  The timed region starts behind it.)"""
    world = synthetic.pyramid_world(base)
    params = wire.StepParams.make("TGS_Soft", 1.0 / 60.0, vel, pos, True)
    keys = ("bodies", "contacts", "joints", "shapes", "pairs", "origins")
    with hip.Solver(device_index) as gpu:
        gpu.world_upload(*[world[k] for k in keys])
        for _ in range(settle):
            gpu.world_step(params)
        t0 = time.perf_counter()
        solve_ms = contacts_ms = 0.0
        active = changed = moved = 0
        for _ in range(steps):
            info = gpu.world_step(params)
            solve_ms += info["solveMs"]
            contacts_ms += info["contactsMs"]
            active += info["activeContacts"]
            changed += info["graphChanged"]
            moved += info["movedCount"]
        elapsed = time.perf_counter() - t0
        st = gpu.stats()
        # ... and the same loop with stage 1 as well (src/world.c:125-126): the device pair query after every step that
        # re-inflated a fat box -- every step of this pile -- and s2CreateContact's part for whatever it finds (nothing, here)
        queries = found = 0
        t0 = time.perf_counter()
        for _ in range(steps):
            if info["movedCount"] > 0:
                found += len(gpu.world_find_pairs())
                queries += 1
            info = gpu.world_step(params)
        with_stage1 = time.perf_counter() - t0
    sweeps = wire.solve_sweeps_per_step("TGS_Soft", vel, pos)
    return {"workload": "LargePyramid base-%d as a resident world, %d settle + %d timed s2amd_world_step (update contacts -> s2Solve_TGS_Soft -> refit)" % (
                base, settle, steps),
            "whole_step_ms": 1e3 * elapsed / steps, "whole_step_with_pair_query_ms": 1e3 * with_stage1 / steps, "pair_queries": queries, "new_pairs_found": found,
            "solver_device_ms": solve_ms / steps, "update_contacts_ms_incl_readback": contacts_ms / steps,
            "mean_active_constraints": active / steps, "steps_with_graph_change": changed, "mean_moved_shapes": moved / steps,
            "value_whole_step": (active / steps) * sweeps * steps / elapsed, "unit": "constraint-iters/s", "persistent_strip_kernel": bool(st["persistent"]),
            "kernel_launches_per_solve": st["kernelLaunches"]}


def fast_leg(device_index, base, vel, pos, steps, warmup, graph, opts):
    """`value_fast`: the headline loop again on the tolerance-mode build (solver2d_amd/libs2amd_fast.so: the same sources, FMA
    contraction on in the device code) -- SURVEY.md 7: "-ffp-contract=off for parity builds, fast for perf builds, report both".
    Same world, same options, same timed region; its own dominant-kernel time; and how far 30 steps of it end from 30 steps of
    the bit-exact build (two floating-point evaluations of the same sweep order).  tests/test_gpu_fast.py states and checks the
    tolerances against the oracle."""
    try:
        hip.load(fast=True)
    except hip.S2AmdError as e:
        return {"error": str(e)}
    pre = synthetic.pyramid(base)
    params = wire.StepParams.make("TGS_Soft", 1.0 / 60.0, vel, pos, True)
    sweeps = wire.solve_sweeps_per_step("TGS_Soft", vel, pos)

    def make(fast):
        g = hip.Solver(device_index, graph=graph, fast=fast)
        for kv in opts:
            key, _, val = kv.partition("=")
            g.set_option(key, int(val))
        g.set_option("strip_patience", 0)
        g.upload(*pre)
        return g
    gpu = make(True)
    gpu.set_option("async", 1)
    for _ in range(warmup):
        gpu.step_resident(params)
    gpu.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        gpu.step_resident(params)
    gpu.synchronize()
    elapsed = time.perf_counter() - t0
    gpu.set_option("async", 0)
    gpu.step_resident(params)
    st = gpu.stats()
    C = st["constraintCount"]
    us, _launches, c_launch = gpu.measure_dominant(params, repeats=40)
    gpu.close()
    # one step of each build from the same state (30 steps into the bit-exact build's run)
    g = make(False)
    for _ in range(30):
        g.step_resident(params)
    start = tuple(x.copy() for x in pre)
    g.download(*start)
    g.close()
    ends = []
    for fast in (True, False):
        g = hip.Solver(device_index, graph=graph, fast=fast)
        g.set_option("strip_patience", 0)
        state = tuple(x.copy() for x in start)
        g.upload(*state)
        g.step_resident(params)
        g.download(*state)
        g.close()
        ends.append(state)
    dvel = float(np.abs(ends[0][0]["linearVelocity"] - ends[1][0]["linearVelocity"]).max())
    dw = float(np.abs(ends[0][0]["angularVelocity"] - ends[1][0]["angularVelocity"]).max())
    dimp = float(np.abs(ends[0][1]["points"]["normalImpulse"] - ends[1][1]["points"]["normalImpulse"]).max())
    vscale = max(float(np.abs(ends[1][0]["linearVelocity"]).max()), 1.0)
    achieved = ALGO_BYTES_PER_CONSTRAINT_SWEEP * c_launch / max(us * 1e-6, 1e-12) / 1e9
    return {"value": C * sweeps * steps / elapsed, "unit": "constraint-iters/s", "ms_per_step": 1e3 * elapsed / steps, "steps": steps, "warmup": warmup,
            "build": hip.load(fast=True).s2amd_build_flags().decode(), "library": os.path.relpath(hip.FAST_LIB_PATH, ROOT),
            "dtype": "f32 with contracted multiply-adds (one rounding where the reference has two)",
            "kernel_launches_per_step": st["kernelLaunches"], "persistent": bool(st["persistent"]),
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "avg_launch_us": us, "constraints_per_launch": c_launch, "algorithmic_bytes_per_launch": ALGO_BYTES_PER_CONSTRAINT_SWEEP * c_launch},
            "tolerance": "every s2Solve_* output within 1e-5 x sweeps of the oracle, norm-wise, on the golden inputs and at base 40 / 200 "
                         "(tests/test_gpu_fast.py, tools/fast_mode_error.py); NOT bit-equal to the reference -- `value` above is",
            "one_step_vs_bit_exact_build": {"from": "the bit-exact build's state after 30 steps, one s2Solve_TGS_Soft on each build", "max_abs_linear_velocity_m_s": dvel,
                                            "max_abs_angular_velocity_rad_s": dw, "max_abs_normal_impulse": dimp,
                                            "velocity_error_over_scale_per_sweep": dvel / vscale / (sweeps + vel)}}


def sharded_abi_leg(device_index, islands, base, vel, pos, steps, warmup, shard_counts=(1, 2, 4), device_lists=None):
    """BASELINE.json configs[4] through the C-ABI's own sharding (include/solver2d_amd.h: s2amd_sharded_*; csrc/sharded.hip): ONE
    process, the world's islands found on the device and bin-packed onto k shards -- here k LOGICAL shards on this one GPU, so the
    numbers say what the partition and the per-step exchange (one kernel per shard storing its rows into every shard's copy of the
    world's body records, on a stream of its own; between distinct GPUs one ncclAllGather per device instead) cost, not how k GPUs
    scale; the multi-process form over RCCL is `island_sharded`.  The k-shard results equal the unsharded world's bit for bit
    (tests/test_gpu_sharded.py)."""
    world = synthetic.pyramid(base, count=islands)
    params = wire.StepParams.make("TGS_Soft", 1.0 / 60.0, vel, pos, True)
    sweeps = wire.solve_sweeps_per_step("TGS_Soft", vel, pos)
    C = int((world[1]["pointCount"] > 0).sum())
    rows = []
    for devices_k in (device_lists if device_lists is not None else [[device_index] * k for k in shard_counts]):
        k = len(devices_k)
        with hip.ShardedSolver(devices_k) as sh:
            t0 = time.perf_counter()
            sh.upload(*world)
            upload_ms = 1e3 * (time.perf_counter() - t0)
            for _ in range(warmup):
                sh.step(params)
            # consecutive steps enqueued back to back, one host wait at the end -- as the unsharded line and `island_sharded` are timed
            t0 = time.perf_counter()
            for _ in range(steps):
                sh.step_async(params)
            sh.wait()
            ms = 1e3 * (time.perf_counter() - t0) / steps
            ops, _waits, form = sh.step_ops()
            t0 = time.perf_counter()
            for _ in range(steps):
                sh.step(params)
            ms_waited = 1e3 * (time.perf_counter() - t0) / steps
            owner, n_islands, _ = sh.partition()
            per_shard = np.bincount(owner[owner >= 0], minlength=k).tolist()
        rows.append({"shards": k, "devices": list(devices_k), "ms_per_step": ms, "ms_per_step_waiting_after_each": ms_waited, "value": C * sweeps / (ms * 1e-3),
                     "upload_and_partition_ms": upload_ms, "islands": n_islands, "bodies_per_shard": per_shard,
                     "exchange": ("stores", "rccl", "peer copies")[form], "stream_ops_per_step": ops,
                     "exchange_bytes_per_step": int(sum(per_shard) * 32 * max(k - 1, 0))})
    if device_lists is not None:
        return {"workload": "%d independent base-%d pyramids (%d constraints), s2_solverTGS_Soft %d/%d; ONE process, one shard per GPU behind the C-ABI "
                            "(s2amd_sharded_*): islands found on device 0, bin-packed, the per-step exchange of the owned body records over RCCL "
                            "(ncclAllGather, single process) where it loads, peer copies otherwise" % (islands, base, C, vel, pos),
                "unit": "constraint-iters/s", "steps": steps, "scaling": "strong", "by_device_count": rows}
    return {"workload": "%d independent base-%d pyramids (%d constraints), s2_solverTGS_Soft %d/%d; islands found on the device (s2amd_find_islands), "
                        "bin-packed onto k logical shards of ONE GPU by s2amd_sharded_upload; one exchange of the owned body records per step" % (
                            islands, base, C, vel, pos),
            "unit": "constraint-iters/s", "steps": steps, "by_shard_count": rows,
            "note": "logical shards of one GPU: what the partition and the exchange cost; not a scaling measurement"}


def sharded_abi_multi_gpu(world, islands, base, vel, pos):
    """configs[4] through the C-ABI's own sharding on 1, 2, ... `world` GPUs of this node, in a process of its own with a hard time limit
    (rank 0 of an N-rank job runs it after everything else: a single process that drives N devices, RCCL's single-process communicators
    included -- if it cannot run here, the line says why and the job's numbers stand)."""
    import subprocess
    lists = []
    k = 1
    while k <= world:
        lists.append(",".join(str(d) for d in range(k)))
        k *= 2
    cmd = [sys.executable, os.path.abspath(__file__), "--sharded-abi-devices", ";".join(lists), "--islands", str(islands), "--island-base", str(base),
           "--vel-iters", str(vel), "--pos-iters", str(pos)]
    env = dict(os.environ)
    for key in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GROUP_RANK", "ROLE_RANK", "LOCAL_WORLD_SIZE"):
        env.pop(key, None)
    try:
        out = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=240)
    except subprocess.TimeoutExpired:
        return {"skipped": "no result within 240 s"}
    except Exception as e:  # pragma: no cover
        return {"skipped": repr(e)}
    for line in reversed(out.stdout.decode(errors="replace").splitlines()):
        if line.startswith("{"):
            try:
                return json.loads(line)
            except ValueError:
                break
    return {"skipped": "exit code %d: %s" % (out.returncode, out.stderr.decode(errors="replace")[-400:])}


def churn_leg(device_index, base):
    """SURVEY.md 8f row 4 as a number: the headline world while its graph changes -- heavy balls shot into the base-200 pyramid, 240 steps
    of the whole loop a caller of the C-ABI runs (pair query -> s2CreateContact on the caller's pool -> s2amd_world_set_contacts ->
    s2amd_world_step), contacts created and destroyed in every other step.  tools/churn_bench.py is the long form of this object."""
    try:
        from tools import churn_bench
    except Exception as e:  # (tests/ is not importable: a stripped checkout)
        return {"skipped": repr(e)}
    d = churn_bench.run(churn_bench.Args(base=base, device=device_index))
    return {"workload": d["world"] + ", s2_solverTGS_Soft 8/4, %d steps of the whole loop" % d["steps"], "steps": d["steps"],
            "steps_with_created_or_destroyed_contacts": d["steps_with_created_or_destroyed_contacts"], "contacts_created": d["contacts_created"],
            "contacts_destroyed": d["contacts_destroyed"], "steps_on_persistent_kernel": d["steps_on_persistent_kernel"],
            "steps_that_built_a_structure": d["steps_that_rebuilt_the_structure"], "contacts_placed_without_rebuild": d["contacts_placed_without_rebuild"],
            "churn_step_median_ms": d["churn_steps_median"]["step_ms"], "quiet_step_median_ms": d["quiet_steps_median"]["step_ms"],
            "mean_step_ms": d["all_steps"]["step_ms"], "start_up_steps_ms": d["start_up_steps_ms"], "slowest_steps_ms": d["slowest_steps_ms"][:4],
            "steps_over_1ms_after_start_up": d["steps_over_1ms"], "steps_over_2ms_after_start_up": d["steps_over_2ms"],
            "overflow": {k: v for k, v in d["overflow"].items() if k != "note"}, "structure_builds_by_the_worker_thread": d["structure_builds_by_the_worker_thread"],
            "churn_step_median_parts_ms": {k: round(v, 4) for k, v in d["churn_steps_median"].items() if k != "launches"}}


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
    ap.add_argument("--config", type=int, default=2, choices=(2, 5),
                    help="2 = BASELINE's headline (LargePyramid base-200, one island per GPU); 5 = configs[4]: 512 x base-40, islands sharded over "
                         "the GPUs, as the line itself")
    ap.add_argument("--sharded-abi-devices", default=None, metavar="0;0,1;0,1,2,3",
                    help="only this: configs[4] through s2amd_sharded_* on each of the given device lists, one process (see sharded_abi_multi_gpu)")
    ap.add_argument("--islands", type=int, default=512)
    ap.add_argument("--weak", action="store_true", help="--config 5: --islands pyramids PER GPU (weak scaling) instead of in all (strong scaling)")
    ap.add_argument("--island-base", type=int, default=40)
    ap.add_argument("--no-fast", action="store_true", help="skip the value_fast object (the headline loop on the tolerance-mode build)")
    ap.add_argument("--no-extras", action="store_true", help="only the headline line (no whole_step / configs / island_sharded objects)")
    ap.add_argument("--restore", action="store_true", help="copy the step-0 bodies back before every step (round 3's headline loop) instead of consecutive steps")
    args = ap.parse_args()

    if args.sharded_abi_devices:
        lists = [[int(d) for d in part.split(",")] for part in args.sharded_abi_devices.split(";") if part]
        have = hip.load().s2amd_device_count()
        if any(d >= have for part in lists for d in part):
            print(json.dumps({"skipped": "this process sees %d device(s)" % have}))
            return
        print(json.dumps(sharded_abi_leg(0, args.islands, args.island_base, args.vel_iters, args.pos_iters, 30, 5, device_lists=lists)))
        return

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the N ranks ourselves (one process per GPU, RCCL) rather than
        # report N GPUs from one process.  Any failure to get N ranks on N devices ends non-zero (below).
        import socket
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        os.execv(sys.executable, [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
                                  "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:])
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world != max(args.gpus, 1):
        sys.stderr.write("bench.py: --gpus %d but the launcher started %d rank(s): refusing to report a GPU count that is not there\n" % (args.gpus, world))
        sys.exit(2)
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
    # what is really there: the collective's own world size and every rank's GPU (PCI bus id), gathered
    single_device_hook = os.environ.get("S2AMD_BENCH_SINGLE_DEVICE") == "1"
    my_device = hip.device_bus_id(device_index if distributed else 0)
    ranks_seen, devices = 1, [my_device]
    if distributed:
        ranks_seen = dist.get_world_size()
        devices = [None] * ranks_seen
        dist.all_gather_object(devices, my_device)
        if ranks_seen != world or (len(set(devices)) != ranks_seen and not single_device_hook):
            sys.stderr.write("bench.py: %d rank(s) on devices %s: not %d distinct GPUs\n" % (ranks_seen, devices, world))
            dist.destroy_process_group()
            sys.exit(3)
    n_gpus = ranks_seen
    ranks = Ranks(dist if distributed else None, __import__("torch") if distributed else None, backend, rank, world, device_index if distributed else 0)

    if args.config == 5:
        # the island-sharded job as the line itself (strong scaling); the contract's keys, its own roofline
        line = island_sharded_leg(ranks, args.islands, args.island_base, args.vel_iters, args.pos_iters, args.steps, args.warmup,
                                  graph=not args.no_graph, dump=os.environ.get("S2AMD_BENCH_DUMP"), weak=args.weak)
        line.update({"higher_is_better": True, "vs_baseline": None, "dtype": "f32", "data": "synthetic", "ranks_seen": ranks_seen, "devices": devices})
        if rank == 0:
            print(json.dumps(line))
        if distributed:
            dist.barrier()
            dist.destroy_process_group()
        return

    pre = synthetic.pyramid(args.base)
    params = wire.StepParams.make("TGS_Soft", 1.0 / 60.0, args.vel_iters, args.pos_iters, True)
    sweeps = wire.solve_sweeps_per_step("TGS_Soft", args.vel_iters, args.pos_iters)

    gpu = hip.Solver(device_index if distributed else 0, graph=not args.no_graph)
    for kv in args.opt:
        key, _, val = kv.partition("=")
        gpu.set_option(key, int(val))
    gpu.set_option("async", 1)  # steps are enqueued back to back; sync() below waits for the solver's stream
    # this world's constraint graph never changes: ask for the strip structure at once (by default it waits until the graph
    # has been unchanged for a step, and searches for a better strip partition only after 32 quiet steps)
    gpu.set_option("strip_patience", 0)
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
        if args.restore:
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
    if args.restore:
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
    which = gpu.stats().get("pairLanes", 0)
    kernel_prefix = {2: "_Z14wideStepKernel", 1: "_Z14pairStepKernel"}.get(which, "_Z15stripStepKernel")
    kernel_name = {2: "wideStepKernel (wide_kernel.hip: 512 threads per strip)", 1: "pairStepKernel (pair_kernel.hip: two lanes per constraint)"}.get(
        which, "stripStepKernel<SOFT_TGS> (strip_kernel.hip: 256 threads per strip)")
    # per-launch event pairs (eager launches), kept as a cross-check against rocprofv3's per-kernel durations
    prof_steps = 3
    gpu.set_option("profile", 1)
    kernel_ms, launches, overhead_ms = 0.0, 0, 0.0
    for _ in range(prof_steps):
        if args.restore:
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
                "trajectory": "step-0 bodies copied back before every step" if args.restore else "consecutive resident steps (no restore, no copy in the timed region)",
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": pmc_traffic_bytes(kernel_prefix, live_us=avg_launch_us)[0] if persistent and args.base == 200 else None,
                "traffic_source": pmc_traffic_bytes(kernel_prefix, live_us=avg_launch_us)[1] if persistent and args.base == 200 else None,
                "kernel": kernel_name + " -- whole step, one persistent launch; constraints_per_launch counts "
                          "constraint-sweeps" if persistent else "solveContactsSoftKernel<SOFT_TGS> / stripSoftKernel<SOFT_TGS>",
                "avg_launch_us": avg_launch_us, "launches_per_step": 1 if persistent else launches_per_sweep * sweeps,
                "constraints_per_launch": constraints_per_launch,
                "eager_event_pair_us_per_launch": 1e3 * kernel_ms / max(launches, 1), "empty_event_pair_us": overhead_ms * 1e3,
                "algorithmic_bytes_per_launch": ALGO_BYTES_PER_CONSTRAINT_SWEEP * constraints_per_launch,
            },
        }
        out["ranks_seen"], out["devices"] = ranks_seen, devices
        if persistent and args.base == 200:
            issue = pmc_issue((kernel_prefix,), live_us=avg_launch_us)
            issue = finish_issue(issue, st["stripCount"])
            if issue is not None and "refused" not in issue:
                issue["note"] = ("one island, %d strips = %d of 256 CUs hold a workgroup; a colour round is one wave's instruction stream per SIMD (4 cycles per "
                                 "instruction), %d dependent rounds per step: instruction-issue / latency bound, HBM idle" % (
                                     st["stripCount"], min(st["stripCount"], 256), 24 * 8))
            out["issue"] = issue
        if world == 1 and not args.no_fast:
            out["value_fast"] = fast_leg(0, args.base, args.vel_iters, args.pos_iters, args.steps, args.warmup, not args.no_graph, args.opt)
        if not args.no_cpu and world == 1:  # the contract: on rank 0 at N = 1 only
            out["cpu_baseline"] = cpu_baseline(args.base, args.vel_iters, args.pos_iters, args.cpu_seconds)
    gpu.close()
    # ---- beyond the headline (outside its timed region): the other BASELINE configurations, driver-run ----
    if not args.no_extras:
        extra_steps = max(10, min(args.steps, 60))
        sharded = island_sharded_leg(ranks, args.islands, args.island_base, args.vel_iters, args.pos_iters, extra_steps, 5, graph=not args.no_graph)
        # ... and with 512 islands PER GPU: the curve that can be linear (the strong one stops at one island's latency per rank)
        weak = island_sharded_leg(ranks, args.islands, args.island_base, args.vel_iters, args.pos_iters, extra_steps, 5, graph=not args.no_graph, weak=True) if world > 1 else None
        if rank == 0:
            out["island_sharded"] = sharded
            if weak is not None:
                out["island_sharded_weak"] = weak
            if world > 1:
                # the same partition in ONE process behind the C-ABI, RCCL from the library (no PyTorch): measured on the node's GPUs
                out["sharded_abi_multi_gpu"] = sharded_abi_multi_gpu(world, args.islands, args.island_base, args.vel_iters, args.pos_iters)
            if world == 1:
                out["whole_step"] = whole_step_leg(ranks.device_index, args.base, args.vel_iters, args.pos_iters, 200, 240)
                # SURVEY.md 8d's trajectory figure (settled world, stage 3 -> solve -> stage 4 every step): the honest whole-step number
                out["value_whole_step"] = out["whole_step"]["value_whole_step"]
                out["configs"] = {"3_tumbler": tumbler_leg(ranks.device_index, 10000, 120, 100),
                                  "4_joint_grid": joint_grid_leg(ranks.device_index, 100, 20),
                                  "5_one_gpu": {k: sharded[k] for k in ("value", "unit", "ms_per_step", "config", "roofline")}}
                out["churn"] = churn_leg(ranks.device_index, args.base)
                out["sharded_abi"] = sharded_abi_leg(ranks.device_index, args.islands, args.island_base, args.vel_iters, args.pos_iters, 30, 5)
    if rank == 0:
        print(json.dumps(out))
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
