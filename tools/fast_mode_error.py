"""How far libs2amd_fast.so (FMA contraction on) is from the oracle: every golden input, and base-40 / base-200 pyramids over
several steps, each solve compared with the oracle swept in the fast library's own constraint order.  Prints one JSON object
(the worst norm-wise relative error per field, per sweep) -- the measurement behind the bounds tests/test_gpu_fast.py checks.
Usage (GPU box): python tools/fast_mode_error.py [--out gpurun_out/fast_mode_error.json]"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from solver2d_amd import hip, synthetic, wire  # noqa: E402
from tests import common, golden_util, oraclebind  # noqa: E402


def one(solver, params, pre):
    got = common.copy3(pre)
    solver.solve(params, *got)
    order, _ = solver.contact_order()
    jorder, _ = solver.joint_order()
    want = common.copy3(pre)
    oraclebind.solve(params, *want, contact_order=order, joint_order=jorder)
    return got, want


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--steps", type=int, default=12)
    a = ap.parse_args()
    worst = {}
    worst_case = {}
    per_solver = {}
    with hip.Solver(0, fast=True) as s:
        for path in golden_util.golden_files():
            params, pre, _ = golden_util.load(path)
            got, want = one(s, params, pre)
            sweeps = common.sweeps_touching_bodies(params)
            name = wire.SOLVER_NAMES[params.solverType]
            for k, (e, sc, r) in common.relative_errors(got, want, params).items():
                per = r / sweeps
                if per > worst.get(k, 0.0):
                    worst[k], worst_case[k] = per, os.path.basename(path)
                per_solver[name] = max(per_solver.get(name, 0.0), per)
    traj = {}
    for base in (40, 200):
        for solver_name in ("TGS_Soft", "SoftStep", "PGS_Soft") + (("PGS_NGS_Block", "Jacobi") if base == 40 else ()):
            vel, pos = common.DEFAULT_ITERS[solver_name]
            params = wire.StepParams.make(solver_name, 1.0 / 60.0, vel, pos, True)
            sweeps = common.sweeps_touching_bodies(params)
            with hip.Solver(0, fast=True) as s:
                state = common.copy3(synthetic.pyramid(base))
                w = 0.0
                for step in range(a.steps if base == 40 else 4):
                    got, want = one(s, params, state)
                    w = max(w, max(r for (_e, _sc, r) in common.relative_errors(got, want, params).values()) / sweeps)
                    state = got  # the fast library's own trajectory: each step is compared from the same input
                traj["pyramid%d/%s" % (base, solver_name)] = w
    out = {"rtol_per_sweep_stated": common.FAST_RTOL_PER_SWEEP, "worst_per_sweep_by_field": worst, "worst_case_by_field": worst_case,
           "worst_per_sweep_by_solver": per_solver, "pyramid_steps_worst_per_sweep": traj,
           "build_flags": hip.load(fast=True).s2amd_build_flags().decode()}
    text = json.dumps(out, indent=1, sort_keys=True)
    print(text)
    if a.out:
        os.makedirs(os.path.dirname(a.out), exist_ok=True)
        open(a.out, "w").write(text + "\n")


if __name__ == "__main__":
    main()
