#!/usr/bin/env python3
"""BASELINE config 5's strong-scaling ceiling, measured on ONE GPU: what a rank of an N-GPU job would run -- 512 / N base-40
pyramids, TGS_Soft 8/4, resident -- for N = 1, 2, 4, 8, on the two paths the library has for small islands:

  islands : one 512-thread workgroup per island for the whole step (islandStepKernel): latency-bound per island, so the step
            time stops falling once every island has a CU to itself (N >= 2);
  strips  : the islands cut into strips of BFS levels, one workgroup per strip, seam bodies handed between neighbours
            (wideStepKernel): an island spreads over several CUs, which is what the idle CUs of N >= 4 are good for.

    python tools/config5_scaling.py > profiles/r03_config5_scaling.json
"""
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from solver2d_amd import hip, synthetic, wire  # noqa: E402


def run(islands, strips, steps=40):
    state = synthetic.pyramid(40, count=islands)
    params = wire.StepParams.make("TGS_Soft", 1.0 / 60.0, 8, 4, True)
    with hip.Solver(0) as gpu:
        gpu.set_option("async", 1)
        if strips:
            bodies = len(state[0])
            gpu.set_option("max_group_bodies", 256)  # a base-40 pyramid (821 bodies) fits no LDS group: strips
            gpu.set_option("strip_min_bodies", 0)
            gpu.set_option("strip_patience", 0)
            gpu.set_option("strip_retry", 0)
            gpu.set_option("strip_bodies", max(160, int(math.ceil(bodies / 250.0))))
        gpu.upload(*state)
        for _ in range(5):
            gpu.step_resident(params)
        gpu.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            gpu.step_resident(params)
        gpu.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / steps
        gpu.set_option("async", 0)
        gpu.step_resident(params)
        st = gpu.stats()
        us, _l, _c = gpu.measure_dominant(params, repeats=10)
    return {"islands_on_this_gpu": islands, "path": "strips" if strips else "islands", "ms_per_step": ms, "dominant_kernel_us": us, "launches": st["kernelLaunches"],
            "workgroups": st["stripCount"] if st["stripCount"] else st["groupCount"], "persistent": st["persistent"], "kernel": st["pairLanes"],
            "constraints": st["constraintCount"]}


def main():
    rows = []
    for islands in (512, 256, 128, 64):
        rows.append(run(islands, False))
        if islands <= 256:
            try:
                rows.append(run(islands, True))
            except Exception as e:  # a partition that fits no strip kernel
                rows.append({"islands_on_this_gpu": islands, "path": "strips", "error": repr(e)})
        sys.stderr.write(json.dumps(rows[-2:]) + "\n")
    best = {}
    for r in rows:
        if "ms_per_step" in r:
            n = 512 // r["islands_on_this_gpu"]
            if n not in best or r["ms_per_step"] < best[n]["ms_per_step"]:
                best[n] = r
    one = best[1]["ms_per_step"]
    curve = {str(n): {"ms_per_step": best[n]["ms_per_step"], "path": best[n]["path"], "speedup": one / best[n]["ms_per_step"],
                      "efficiency": one / best[n]["ms_per_step"] / n} for n in sorted(best)}
    print(json.dumps({"what": "512 x base-40 TGS_Soft 8/4, the per-rank work of an N-GPU run measured on one MI355X (no exchange in the timed loop)",
                      "rows": rows, "implied_strong_scaling": curve}, indent=1))


if __name__ == "__main__":
    main()
