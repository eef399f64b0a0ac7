#!/usr/bin/env python3
"""Whole-step timing of the resident world chain (s2amd_world_step: update contacts -> s2Solve_TGS_Soft -> refit) on the
LargePyramid base-N world, beside the UNMODIFIED reference's s2World_Step on the host (oracle/_ref, when it is there).

    python tools/world_bench.py [--base 200] [--steps 200] [--warmup 60] > gpurun_out/world.json
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from solver2d_amd import hip, synthetic, wire  # noqa: E402
from tests import world_chain  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--base", type=int, default=200)
    ap.add_argument("--count", type=int, default=1, help="independent pyramids in the world (BASELINE config 5: --base 40 --count 512)")
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=60)
    ap.add_argument("--ref-steps", type=int, default=20)
    ap.add_argument("--opt", action="append", default=[], metavar="KEY=VALUE")
    ap.add_argument("--churn", action="store_true", help="make the contact graph change every step (one manifold is told to be empty before each step)")
    ap.add_argument("--no-pairs", action="store_true", help="skip the stage-1 pair query in the steps where the refit moved shapes")
    a = ap.parse_args()
    params = wire.StepParams.make("TGS_Soft", 1.0 / 60.0, 8, 4, True)
    world = synthetic.pyramid_world(a.base, a.count)
    out = {"world": "%d x LargePyramid base-%d: %d bodies, %d shapes, %d contact slots" % (a.count, a.base, len(world["bodies"]), len(world["shapes"]), len(world["contacts"])),
           "solver": "TGS_Soft 8/4 warm start"}
    with hip.Solver(0) as s:
        for kv in a.opt:
            k, v = kv.split("=")
            s.set_option(k, int(v))
        s.world_upload(*[world[k] for k in world_chain.WORLD_KEYS])
        found = queries = 0
        pair_s = 0.0
        prep = []
        k = len(world["contacts"]) - 1
        churn_slot = np.array([k], dtype=np.int32)
        churn_contact = world["contacts"][k:k + 1].copy()
        churn_contact["pointCount"] = 0
        churn_pair = world["pairs"][k:k + 1].copy()

        def step():
            nonlocal found, queries, pair_s
            if a.churn:
                s.world_set_contacts(churn_slot, churn_contact, churn_pair)
            info = s.world_step(params)
            prep.append(s.stats()["hostPrepMs"])
            if info["movedCount"] > 0 and not a.no_pairs:
                t = time.perf_counter()
                found += len(s.world_find_pairs())  # stage 1 for the next step (a settled pyramid finds none)
                pair_s += time.perf_counter() - t
                queries += 1
            return info

        infos = [step() for _ in range(a.warmup)]
        found = queries = 0
        pair_s = 0.0
        t0 = time.perf_counter()
        infos = [step() for _ in range(a.steps)]
        ms = 1e3 * (time.perf_counter() - t0) / a.steps
        out.update({"host_structure_ms": sum(prep[-a.steps:]) / a.steps, "pair_queries": queries, "new_pairs_found": found, "pair_query_ms": 1e3 * pair_s / max(queries, 1)})
        st = s.stats()
        out.update({
            "gpu_world_step_ms": ms,
            "stage3_ms_incl_readback": sum(i["contactsMs"] for i in infos) / len(infos),
            "solve_device_ms": sum(i["solveMs"] for i in infos) / len(infos),
            "active_contacts": infos[-1]["activeContacts"],
            "graph_changes": sum(i["graphChanged"] for i in infos),
            "moved_shapes": sum(i["movedCount"] for i in infos),
            "separated": sum(i["separatedCount"] for i in infos),
            "persistent": st["persistent"], "kernel_launches_solve": st["kernelLaunches"],
        })
    try:
        from tests import refbind
        if refbind.available() and a.count == 1 and a.ref_steps > 0:
            with refbind.RefWorld("pyramid", "TGS_Soft", a.base, 0) as ref:
                for _ in range(5):
                    ref.step(1.0 / 60.0, 8, 4, True)
                t0 = time.perf_counter()
                for _ in range(a.ref_steps):
                    ref.step(1.0 / 60.0, 8, 4, True)
                out["reference_world_step_ms"] = 1e3 * (time.perf_counter() - t0) / a.ref_steps
                out["reference"] = "unmodified reference s2World_Step (all four stages), 1 host thread, %d steps" % a.ref_steps
                out["speedup"] = out["reference_world_step_ms"] / ms
    except Exception as e:  # the reference build is optional on the GPU box
        out["reference_error"] = repr(e)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
