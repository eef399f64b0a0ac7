"""One-off soak of the persistent kernels' hand-off protocol (timing-dependent, so it is run long): a resident world stepped N
times on a one-launch kernel -- the op interpreter (generic_kernel.hip) for the non-soft solvers and joint worlds, the 512-thread
kernel with parked seam rounds for TGS_Soft -- against the same world on the multi-launch strip path, which shares its sweep
order: every array must be bit-equal at every checkpoint.
    python tools/handoff_soak.py [steps=2000] [checkpoint=250]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from solver2d_amd import hip, synthetic, wire  # noqa: E402
from tests import common  # noqa: E402


def run(name, world, solver_name, fast_opts, slow_opts, steps, every):
    vel, pos = common.DEFAULT_ITERS[solver_name]
    params = wire.StepParams.make(solver_name, 1.0 / 60.0, vel, pos, True)
    fast, slow = hip.Solver(0), hip.Solver(0)
    for s, opts in ((fast, fast_opts), (slow, slow_opts)):
        s.set_option("strip_patience", 0)
        s.set_option("strip_min_bodies", 0)
        for k, v in opts.items():
            s.set_option(k, v)
        s.upload(*world)
    t0 = time.time()
    bad = 0
    for step in range(1, steps + 1):
        fast.step_resident(params)
        slow.step_resident(params)
        if step == 1:
            assert fast.stats()["persistent"] == 1 and slow.stats()["persistent"] == 0, (fast.stats(), slow.stats())
        if step % every == 0 or step == steps:
            a, b = common.copy3(world), common.copy3(world)
            fast.download(*a)
            slow.download(*b)
            same = all(x.tobytes() == y.tobytes() for x, y in zip(a, b))
            finite = bool(np.isfinite(a[0]["position"]).all())
            bad += 0 if same else 1
            print("%s step %d: %s%s, kernel %d, fallbacks %d" % (name, step, "equal" if same else "DIFFERENT", "" if finite else " (non-finite)",
                                                                fast.stats()["pairLanes"], fast.stats()["persistFallbacks"]), flush=True)
    print("%s: %d steps in %.1f s, %d bad checkpoints" % (name, steps, time.time() - t0, bad), flush=True)
    fast.close(), slow.close()
    return bad


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    every = int(sys.argv[2]) if len(sys.argv) > 2 else 250
    bad = 0
    multi = {"generic": 0, "strips_any_solver": 1}
    bad += run("pyramid 120 / PGS_NGS_Block, op interpreter", synthetic.pyramid(120), "PGS_NGS_Block", {}, multi, steps, every)
    bad += run("pyramid 120 / XPBD, op interpreter", synthetic.pyramid(120), "XPBD", {}, multi, steps, every)
    bad += run("joint grid 70 / PGS_NGS, op interpreter", synthetic.joint_grid(70), "PGS_NGS", {}, multi, steps, every)
    bad += run("pyramid 120 / TGS_Soft, parked seam rounds", synthetic.pyramid(120), "TGS_Soft", {"persist_debug": 16}, {"persist": 0}, steps, every)
    bad += run("pyramid 200 / TGS_Soft, 512-thread kernel", synthetic.pyramid(200), "TGS_Soft", {}, {"persist": 0}, steps, every)
    print("HANDOFF SOAK", "FAILED" if bad else "OK")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
