#!/usr/bin/env python3
"""PCIe-inclusive cost of the drop-in call s2amd_solve (host arrays in and out every step) on LargePyramid base-200
TGS_Soft 8/4, arrays in ordinary (pageable) memory."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from solver2d_amd import hip, synthetic, wire  # noqa: E402


def main():
    base = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    world = synthetic.pyramid(base)
    p = wire.StepParams.make("TGS_Soft", 1.0 / 60.0, 8, 4, True)
    out = {"config": "s2amd_solve, pyramid base-%d TGS_Soft 8/4" % base, "bytes_each_way": int(sum(a.nbytes for a in world))}
    for name, arrays in (("pageable", tuple(a.copy() for a in world)),):
        with hip.Solver(0) as gpu:
            for _ in range(10):
                gpu.solve(p, *arrays)
            t0 = time.perf_counter()
            steps = 40
            for _ in range(steps):
                gpu.solve(p, *arrays)
            out[name + "_ms"] = 1e3 * (time.perf_counter() - t0) / steps
            out[name + "_device_ms"] = gpu.stats()["deviceMs"]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
