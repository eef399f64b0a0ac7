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

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from solver2d_amd import hip, synthetic, wire  # noqa: E402
from tests import world_chain  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--base", type=int, default=200)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=60)
    ap.add_argument("--ref-steps", type=int, default=20)
    a = ap.parse_args()
    params = wire.StepParams.make("TGS_Soft", 1.0 / 60.0, 8, 4, True)
    world = synthetic.pyramid_world(a.base)
    out = {"world": "LargePyramid base-%d: %d bodies, %d shapes, %d contact slots" % (a.base, len(world["bodies"]), len(world["shapes"]), len(world["contacts"])),
           "solver": "TGS_Soft 8/4 warm start"}
    with hip.Solver(0) as s:
        s.world_upload(*[world[k] for k in world_chain.WORLD_KEYS])
        infos = [s.world_step(params) for _ in range(a.warmup)]
        t0 = time.perf_counter()
        infos = [s.world_step(params) for _ in range(a.steps)]
        ms = 1e3 * (time.perf_counter() - t0) / a.steps
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
        if refbind.available():
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
