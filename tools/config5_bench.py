#!/usr/bin/env python3
"""BASELINE config 5 (512 x pyramid base-40 in one world, TGS_Soft 8/4) resident on one GPU: the LDS group kernel in its
HBM-bound regime.  Meant to be run under rocprofv3 (kernel trace, FETCH_SIZE / WRITE_SIZE passes):

    python tools/config5_bench.py [--count 512] [--steps 30]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from solver2d_amd import hip, synthetic, wire  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--count", type=int, default=512)
    ap.add_argument("--steps", type=int, default=30)
    a = ap.parse_args()
    state = synthetic.pyramid(40, count=a.count)
    params = wire.StepParams.make("TGS_Soft", 1.0 / 60.0, 8, 4, True)
    with hip.Solver(0) as gpu:
        gpu.upload(*state)
        gpu.save_bodies()
        for _ in range(5):
            gpu.restore_bodies()
            gpu.step_resident(params)
        t0 = time.perf_counter()
        for _ in range(a.steps):
            gpu.restore_bodies()
            gpu.step_resident(params)
        ms = 1e3 * (time.perf_counter() - t0) / a.steps
        st = gpu.stats()
    C = len(state[1])
    print(json.dumps({"config": "5: %d x pyramid base-40" % a.count, "constraints": C, "bodies": len(state[0]), "ms_per_step": ms,
                      "device_ms": st["deviceMs"], "groups": st["groupCount"], "launches": st["kernelLaunches"],
                      "constraint_iters_per_s": C * 16 / (ms / 1e3)}))


if __name__ == "__main__":
    main()
