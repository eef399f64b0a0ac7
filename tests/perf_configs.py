#!/usr/bin/env python3
"""Measures the five BASELINE.json configurations (not a pytest file; run on the GPU box):

    python tests/perf_configs.py [--quick] > gpurun_out/configs.jsonl

Configs 3 (Tumbler) needs real collision, so its solver input is captured from the reference world
(oracle/_ref, test infrastructure) after a settling period; configs 2, 4, 5 use the synthetic
step-0 snapshots.  For every config: GPU ms/step resident in HBM (graph replay), the oracle's ms/step
on the host (1 thread), constraint or joint iterations per second, colours, launches.
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
from tests import common, oraclebind, refbind  # noqa: E402


def measure(name, solver, vel, pos, state, steps, warmup, unit_count, unit_name, cpu_steps):
    params = wire.StepParams.make(solver, 1.0 / 60.0, vel, pos, True)
    sweeps = wire.solve_sweeps_per_step(solver, vel, pos)
    with hip.Solver(0) as gpu:
        gpu.upload(*state)
        gpu.save_bodies()
        for _ in range(warmup):
            gpu.restore_bodies()
            gpu.step_resident(params)
        t0 = time.perf_counter()
        for _ in range(steps):
            gpu.restore_bodies()
            gpu.step_resident(params)
        ms = 1e3 * (time.perf_counter() - t0) / steps
        st = gpu.stats()
        # PCIe-inclusive: the full drop-in call with host arrays in and out
        hs = common.copy3(state)
        t0 = time.perf_counter()
        for _ in range(max(steps // 4, 3)):
            gpu.solve(params, *hs)
        ms_pcie = 1e3 * (time.perf_counter() - t0) / max(steps // 4, 3)
    cs = common.copy3(state)
    b0 = cs[0].copy()
    t0 = time.perf_counter()
    for _ in range(cpu_steps):
        cs[0][:] = b0
        oraclebind.solve(params, *cs)
    cpu_ms = 1e3 * (time.perf_counter() - t0) / cpu_steps
    out = {
        "config": name, "solver": solver, "velIters": vel, "posIters": pos,
        "bodies": int((state[0]["type"] >= 0).sum()), "constraints": st["constraintCount"], "joints": st["jointCount"],
        "contact_colors": st["contactColors"], "joint_colors": st["jointColors"], "launches": st["kernelLaunches"],
        "solve_sweeps": sweeps, "gpu_ms_per_step": ms, "gpu_ms_per_step_pcie_inclusive": ms_pcie, "device_ms": st["deviceMs"],
        "cpu_port_ms_per_step": cpu_ms, "unit": unit_name,
        "gpu_units_per_s": unit_count * sweeps / (ms / 1e3), "cpu_units_per_s": unit_count * sweeps / (cpu_ms / 1e3),
    }
    print(json.dumps(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    args = ap.parse_args()
    steps, warm = (20, 5) if args.quick else (100, 20)

    # config 2
    st = synthetic.pyramid(200)
    measure("2: LargePyramid base-200", "TGS_Soft", 8, 4, st, steps, warm, 59900, "constraint-iters/s", 5)

    # config 4
    st = synthetic.joint_grid(100)
    measure("4: JointGrid 100x100", "PGS_NGS", 4, 2, st, steps, warm, len(st[2]), "joint-iters/s", 5)

    # config 5 (single GPU: all 512 islands on one device)
    n = 64 if args.quick else 512
    st = synthetic.pyramid(40, count=n)
    measure("5: %d x pyramid base-40 (1 GPU)" % n, "TGS_Soft", 8, 4, st, max(steps // 4, 5), 3, len(st[1]), "constraint-iters/s", 2)

    # config 3: captured from the reference world
    if refbind.available():
        count = 2000 if args.quick else 10000
        # settle under TGS_Soft (the reference's Jacobi solver diverges on piles -- it sums all
        # per-body corrections without averaging), then hand that state to the solver under test
        with refbind.RefWorld("tumbler", "TGS_Soft", count, 0) as w:
            for _ in range(60 if args.quick else 150):
                w.step(1.0 / 60.0, 8, 4, True)
            _params, pre, _post = w.step_captured(1.0 / 60.0, 8, 4, True)
        C = int((pre[1]["pointCount"] > 0).sum())
        measure("3: Tumbler %d boxes (captured after settling)" % count, "Jacobi", 4, 2, pre, steps, warm, C, "constraint-iters/s", 5)
        measure("3b: same input, TGS_Soft", "TGS_Soft", 8, 4, pre, max(steps // 4, 5), 3, C, "constraint-iters/s", 3)
    broadphase()
    narrowphase()


def narrowphase():
    """Stage 3 (s2UpdateContact over every live contact) at BASELINE size: GPU (host arrays in and out; kernel time
    from HIP events is not separated here) vs the reference's own loop on this host, 1 thread."""
    if not refbind.available():
        return
    L = refbind.lib()
    with refbind.RefWorld("pyramid", "TGS_Soft", 200, 0) as w:
        for _ in range(3):
            w.step_captured(1.0 / 60.0, 8, 4, True)
        cap = refbind.narrowphase_capture()
        L.s2ref_set_mode(3)
        refbind.narrowphase_seconds(True)
        steps = 5
        for _ in range(steps):
            w.step(1.0 / 60.0, 8, 4, True)
        ref_ms = 1e3 * refbind.narrowphase_seconds(True) / steps
        L.s2ref_set_mode(0)
    live = int((cap["pairs_pre"]["shapeA"] >= 0).sum())
    with hip.Solver(0) as gpu:
        pairs, contacts = cap["pairs_pre"].copy(), cap["contacts_pre"].copy()
        gpu.update_contacts(cap["bodies"], cap["origins"], cap["shapes"], pairs, contacts)
        reps = 10
        t0 = time.perf_counter()
        for _ in range(reps):
            pairs, contacts = cap["pairs_pre"].copy(), cap["contacts_pre"].copy()
            status = gpu.update_contacts(cap["bodies"], cap["origins"], cap["shapes"], pairs, contacts)
        gpu_ms = 1e3 * (time.perf_counter() - t0) / reps
        kernel_ms = gpu.stats()["deviceMs"]
    ok = bool(np.array_equal(contacts["points"][status == 0].tobytes(), cap["contacts_post"]["points"][status == 0].tobytes()))
    print(json.dumps({"config": "narrow phase, pyramid base-200 (%d live box-box contacts, step 3)" % live,
                      "manifolds_equal_reference": ok, "gpu_update_contacts_ms_pcie_inclusive": gpu_ms, "gpu_kernel_ms": kernel_ms,
                      "reference_update_contacts_ms": ref_ms}), flush=True)


def broadphase():
    """Stage 1 pair discovery + Stage 4 refit at BASELINE size: GPU (host arrays in and out) vs the
    reference's own s2UpdateBroadPhasePairs on this host."""
    if not refbind.available():
        return
    L = refbind.lib()
    with refbind.RefWorld("pyramid", "TGS_Soft", 200, 0) as w:
        shapes_before, origins_before = w.pack_shapes()
        _params, pre, post = w.step_captured(1.0 / 60.0, 8, 4, True)
        bp_shapes, moved, existing, created = refbind.broadphase_capture()
    with refbind.RefWorld("pyramid", "TGS_Soft", 200, 0) as w2:
        L.s2ref_set_mode(3)
        L.s2ref_broadphase_seconds(1)
        w2.step(1.0 / 60.0, 8, 4, True)
        ref_s = L.s2ref_broadphase_seconds(1)
        L.s2ref_set_mode(0)
    with hip.Solver(0) as gpu:
        gpu.find_pairs(pre[0], bp_shapes, moved, existing, pre[2])
        t0 = time.perf_counter()
        reps = 10
        for _ in range(reps):
            got = gpu.find_pairs(pre[0], bp_shapes, moved, existing, pre[2])
        pairs_ms = 1e3 * (time.perf_counter() - t0) / reps
        sh, og = shapes_before.copy(), origins_before.copy()
        gpu.refit_shapes(post[0], sh, og)
        t0 = time.perf_counter()
        for _ in range(reps):
            gpu.refit_shapes(post[0], sh, og)
        refit_ms = 1e3 * (time.perf_counter() - t0) / reps
    print(json.dumps({"config": "broad phase, pyramid base-200 first step (all %d proxies moved)" % int(moved.sum()),
                      "new_pairs": int(len(got)), "pairs_equal_reference": bool(len(got) == len(created)),
                      "gpu_find_pairs_ms_pcie_inclusive": pairs_ms, "gpu_refit_ms_pcie_inclusive": refit_ms,
                      "reference_update_pairs_ms": 1e3 * ref_s}), flush=True)


if __name__ == "__main__":
    main()
