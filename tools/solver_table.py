"""Every one of the ten solvers on the headline world (LargePyramid base N, resident, frozen snapshot like bench.py's
headline): ms per step, launches per step, which execution path the structure chose.  One JSON object per line.
    python tools/solver_table.py [--base 200] [--steps 50] > profiles/r02_solver_table.jsonl"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from solver2d_amd import hip, synthetic, wire  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--base", type=int, default=200)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--solvers", default=",".join(wire.SOLVER_NAMES))
    ap.add_argument("--world", default="pyramid", choices=("pyramid", "joint_grid"))
    ap.add_argument("--opt", action="append", default=[], metavar="KEY=VALUE")
    ap.add_argument("--fast", action="store_true", help="the tolerance-mode build (libs2amd_fast.so) instead of the bit-exact one")
    a = ap.parse_args()
    if a.world == "pyramid":
        bodies, contacts, joints = synthetic.pyramid(a.base)
    else:
        bodies, contacts, joints = synthetic.joint_grid(a.base)
    active = int((contacts["pointCount"] > 0).sum()) if len(contacts) else 0
    live_joints = int((joints["type"] >= 0).sum()) if len(joints) else 0
    for name in a.solvers.split(","):
        vel, pos = (8, 4) if name in ("TGS_Soft", "SoftStep") else (4, 2)
        params = wire.StepParams.make(name, 1.0 / 60.0, vel, pos, True)
        with hip.Solver(0, fast=a.fast) as gpu:
            for kv in a.opt:
                k, v = kv.split("=")
                gpu.set_option(k, int(v))
            gpu.set_option("strip_patience", 0)
            gpu.upload(bodies, contacts, joints)
            gpu.save_bodies()
            gpu.set_option("async", 1)
            for _ in range(8):
                gpu.restore_bodies()
                gpu.step_resident(params)
            gpu.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.steps):
                gpu.restore_bodies()
                gpu.step_resident(params)
            gpu.synchronize()
            ms = 1e3 * (time.perf_counter() - t0) / a.steps
            gpu.set_option("async", 0)
            gpu.restore_bodies()
            gpu.step_resident(params)
            st = gpu.stats()
        print(json.dumps({"world": "%s %d" % (a.world, a.base), "solver": name, "vel": vel, "pos": pos, "ms_per_step": round(ms, 4),
                          "launches": st["kernelLaunches"], "device_ms": round(st["deviceMs"], 4), "constraints": active, "joints": live_joints,
                          "colors": st["contactColors"], "joint_colors": st["jointColors"], "strips": st["stripCount"], "groups": st["groupCount"],
                          "persistent": st["persistent"]}), flush=True)


if __name__ == "__main__":
    main()
