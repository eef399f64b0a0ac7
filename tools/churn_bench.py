#!/usr/bin/env python3
"""A CHURNING world on the resident chain: heavy balls shot into a LargePyramid base-N (tests/world_chain.py: wreck_world
-- a generator, no oracle involved), the whole loop per step as a caller of the C-ABI runs it:

    s2amd_world_find_pairs (when the refit moved shapes) -> the caller's s2CreateContact for every new pair ->
    s2amd_world_set_contacts -> s2amd_world_step (update contacts -> s2Solve_* -> refit)

Contacts are created and destroyed in almost every step while the balls plough through the pile.  Reports per-step wall
time split into its parts, the host time spent on the constraint-graph structure (s2amdStepStats.hostPrepMs), and how many
steps rebuilt it.

    python tools/churn_bench.py [--base 200] [--steps 240] [--solver TGS_Soft] > gpurun_out/churn.json
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from solver2d_amd import hip, wire  # noqa: E402
from tests import common, world_chain  # noqa: E402


def create_contacts(world, free, new_pairs):
    """The caller's s2CreateContact (src/contact.c:137-203) on its own copy of the pool: first free slot, mixed friction,
    empty manifold."""
    n = len(new_pairs)
    slots = np.array([free.pop() for _ in range(n)], dtype=np.int32)
    contacts = np.zeros(n, dtype=wire.contact_dtype)
    pairs = np.zeros(n, dtype=wire.pair_state_dtype)
    contacts["bodyA"] = world["shapes"]["body"][new_pairs[:, 0]]
    contacts["bodyB"] = world["shapes"]["body"][new_pairs[:, 1]]
    contacts["friction"] = 0.6
    contacts["constraintIndex"] = -1
    pairs["shapeA"], pairs["shapeB"] = new_pairs[:, 0], new_pairs[:, 1]
    return slots, contacts, pairs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", default="wreck", choices=("wreck", "tumbler"), help="wreck: balls into a base-N pyramid; tumbler: BASELINE config 3's drum with --count boxes, from scratch")
    ap.add_argument("--count", type=int, default=10000)
    ap.add_argument("--base", type=int, default=200)
    ap.add_argument("--seed", type=int, default=3)
    ap.add_argument("--steps", type=int, default=240)
    ap.add_argument("--solver", default="TGS_Soft")
    ap.add_argument("--opt", action="append", default=[], metavar="KEY=VALUE")
    ap.add_argument("--trace", action="store_true", help="one line per step on stderr")
    print(json.dumps(run(ap.parse_args())))


class Args:
    """run()'s arguments for a caller that is not the command line (bench.py's `churn` object)."""

    def __init__(self, **kw):
        self.world, self.count, self.base, self.seed, self.steps, self.solver, self.opt, self.trace, self.device = "wreck", 10000, 200, 3, 240, "TGS_Soft", [], False, 0
        self.__dict__.update(kw)


def run(a):
    vel, pos = common.DEFAULT_ITERS[a.solver]
    params = wire.StepParams.make(a.solver, 1.0 / 60.0, vel, pos, True)
    if a.world == "tumbler":
        from solver2d_amd import synthetic
        world = synthetic.tumbler_world(a.count)
    else:
        world = world_chain.wreck_world(a.seed, a.base)
    free = sorted(np.flatnonzero(world["pairs"]["shapeA"] < 0).tolist(), reverse=True)
    rows = []
    with hip.Solver(getattr(a, "device", 0)) as s:
        s.set_option("prebuild_solver", wire.SOLVER_ID[a.solver])  # (a caller knows its world's solver: the structure is built with the upload)
        for kv in a.opt:
            k, v = kv.split("=")
            s.set_option(k, int(v))
        s.world_upload(*[world[k] for k in world_chain.WORLD_KEYS])
        moved = 1  # the balls were created "in the move buffer"
        status = np.zeros(len(world["contacts"]), dtype=np.int32)
        for step in range(a.steps):
            t0 = time.perf_counter()
            created = 0
            if moved > 0:
                new = s.world_find_pairs()
                t1 = time.perf_counter()
                if len(new):
                    slots, contacts, pairs = create_contacts(world, free, new)
                    t2 = time.perf_counter()
                    s.world_set_contacts(slots, contacts, pairs)
                    created = len(new)
                else:
                    t2 = t1
            else:
                t1 = t2 = t0
            t3 = time.perf_counter()
            info = s.world_step(params)
            t4 = time.perf_counter()
            st = s.stats()
            if info["separatedCount"] > 0:
                # the caller's s2DestroyContact: which pool slots are free again
                free.extend(s.world_separated(info["separatedCount"]).tolist())
            t5 = time.perf_counter()
            moved = info["movedCount"]
            if a.trace:
                sys.stderr.write("step %d: %.2f ms, created %d separated %d active %d potential %d launches %d hostPrep %.2f colours %d placed %d builds %d persistent %d fallbacks %d kernel %d overflow %d sliced %d async %d/%d\n" % (
                    step, 1e3 * (t5 - t0), created, info["separatedCount"], info["activeContacts"], st["potentialConstraints"], st["kernelLaunches"], st["hostPrepMs"],
                    st["contactColors"], st["placedContacts"], st["structureBuilds"], st["persistent"], st["persistFallbacks"], st["pairLanes"], st["overflowContacts"], st["slicedStep"],
                    st["asyncBuildsRequested"], st["asyncBuildsAdopted"]))
            rows.append({"step_ms": 1e3 * (t5 - t0), "pair_query_ms": 1e3 * (t1 - t0), "create_py_ms": 1e3 * (t2 - t1), "set_contacts_ms": 1e3 * (t3 - t2),
                         "world_step_ms": 1e3 * (t4 - t3), "destroy_py_ms": 1e3 * (t5 - t4), "host_structure_ms": st["hostPrepMs"], "solve_device_ms": info["solveMs"],
                         "created": created, "separated": info["separatedCount"], "flips": info["graphChanged"], "active": info["activeContacts"],
                         "launches": st["kernelLaunches"], "persistent": st["persistent"], "replayed": st["graphReplayed"], "strips": st["stripCount"], "sliced": st["slicedStep"], "overflow": st["overflowContacts"]})
    def mean(key, sel):
        v = [r[key] for r in sel]
        return sum(v) / max(len(v), 1)

    def median(key, sel):
        v = sorted(r[key] for r in sel)
        return v[len(v) // 2] if v else 0.0
    churn = [r for r in rows if r["created"] > 0 or r["separated"] > 0]
    quiet = [r for r in rows if not (r["created"] > 0 or r["separated"] > 0)]
    keys = ["step_ms", "pair_query_ms", "create_py_ms", "set_contacts_ms", "world_step_ms", "destroy_py_ms", "host_structure_ms", "solve_device_ms", "launches"]
    out = {"world": ("tumbler_world(%d): " % a.count if a.world == "tumbler" else "wreck_world(seed %d, base %d): " % (a.seed, a.base)) +
                    "%d bodies, %d contact slots" % (len(world["bodies"]), len(world["contacts"])),
           "solver": a.solver, "steps": a.steps, "steps_with_created_or_destroyed_contacts": len(churn),
           "contacts_created": sum(r["created"] for r in rows), "contacts_destroyed": sum(r["separated"] for r in rows),
           "steps_with_manifold_flips": sum(1 for r in rows if r["flips"]), "steps_that_rebuilt_the_structure": sum(1 for r in rows if r["host_structure_ms"] > 0), "contacts_placed_without_rebuild": st["placedContacts"],
           "steps_on_persistent_kernel": sum(r["persistent"] for r in rows), "steps_replayed_from_graph": sum(r["replayed"] for r in rows),
           "joined_without_rebuild": {"bodies_moved_to_the_strip_they_touched": st["bodiesAdopted"], "bodies_added_to_a_seam": st["seamBodiesAdded"], "spare_rounds_opened": st["roundsOpened"],
                                      "note": "counters of the structure in use at the last step (a rebuild starts them again)"},
           "overflow": {"steps_run_sliced": sum(1 for r in rows if r["sliced"] == 1), "steps_with_the_overflow_workgroup_in_the_launch": sum(1 for r in rows if r["sliced"] == 2), "most_contacts_waiting": max(r["overflow"] for r in rows),
                        "overflow_step_ms_median": median("step_ms", [r for r in rows if r["sliced"]]), "overflow_step_launches_median": median("launches", [r for r in rows if r["sliced"]]),
                        "note": "a contact that fits nowhere in the strips waits in an overflow position behind them while a worker thread builds the structure "
                                "that holds it; until that is adopted the persistent launch carries one more workgroup that sweeps the overflow contacts after every "
                                "sweep of the strips (option overflow_kernel 0, or after a time-out: the kernel is launched once per sweep -- sliced)"},
           "structure_builds_by_the_worker_thread": {"requested": st["asyncBuildsRequested"], "adopted": st["asyncBuildsAdopted"], "caller_waited_ms": st["asyncWaitMs"]},
           "steps_over_1ms": sum(1 for r in rows[2:] if r["step_ms"] > 1.0), "steps_over_2ms": sum(1 for r in rows[2:] if r["step_ms"] > 2.0),
           "all_steps": {k: mean(k, rows) for k in keys}, "churn_steps": {k: mean(k, churn) for k in keys}, "quiet_steps": {k: mean(k, quiet) for k in keys},
           "churn_steps_median": {k: median(k, churn) for k in keys}, "quiet_steps_median": {k: median(k, quiet) for k in keys},
           # (steps 0 and 1 build the world's first structures and load the kernels' code objects: start-up, not churn)
           "start_up_steps_ms": [round(r["step_ms"], 3) for r in rows[:2]],
           "start_up_steps_parts_ms": [{k: round(r[k], 3) for k in keys} for r in rows[:2]],
           "slowest_steps_ms": sorted((round(r["step_ms"], 3) for r in rows[2:]), reverse=True)[:8],
           "active_contacts_last": rows[-1]["active"]}
    return out


if __name__ == "__main__":
    main()
