#!/usr/bin/env python3
"""BASELINE config 3 (Tumbler, 10,000 boxes, Jacobi 4/2) resident on one GPU; the solver input is captured from the
reference world (oracle/_ref) after settling.  Meant for rocprofv3 --kernel-trace --stats."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from solver2d_amd import hip, wire  # noqa: E402
from tests import refbind  # noqa: E402


def main():
    solver = sys.argv[1] if len(sys.argv) > 1 else "Jacobi"
    vel, pos = (4, 2) if solver == "Jacobi" else (8, 4)
    with refbind.RefWorld("tumbler", "TGS_Soft", 10000, 0) as w:
        for _ in range(150):
            w.step(1.0 / 60.0, 8, 4, True)
        _params, pre, _post = w.step_captured(1.0 / 60.0, 8, 4, True)
    params = wire.StepParams.make(solver, 1.0 / 60.0, vel, pos, True)
    with hip.Solver(0) as gpu:
        gpu.upload(*pre)
        gpu.save_bodies()
        for _ in range(10):
            gpu.restore_bodies()
            gpu.step_resident(params)
        t0 = time.perf_counter()
        steps = 50
        for _ in range(steps):
            gpu.restore_bodies()
            gpu.step_resident(params)
        ms = 1e3 * (time.perf_counter() - t0) / steps
        st = gpu.stats()
    print(json.dumps({"config": "3: tumbler 10k, " + solver, "constraints": int((pre[1]["pointCount"] > 0).sum()), "ms_per_step": ms,
                      "device_ms": st["deviceMs"], "launches": st["kernelLaunches"], "colors": st["contactColors"]}))


if __name__ == "__main__":
    main()
