"""Drop-in test: the UNMODIFIED reference (broad phase, narrow phase, contact bookkeeping, world
step) drives the HIP solver through the s2Solve_* plug point (oracle/ref_hook.c replace mode).

Per step: the HIP result must equal the oracle run in the device's order bit for bit (L2), and
the whole trajectory must stay physically close to the all-reference trajectory (L3; colour
order differs from pool order, so this link is a stated tolerance, not equality).
Needs oracle/_ref/libs2ref.so (shipped prebuilt to the GPU box).
"""
import numpy as np
import pytest

from solver2d_amd import hip, wire
from tests import common, oraclebind, refbind

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not refbind.available(), reason="oracle/_ref/libs2ref.so not built")]

CASES = [
    ("pyramid", 20, 0, 90),
    ("mixed", 24, 0, 90),
    ("joint_grid", 10, 10, 60),
    ("tumbler", 150, 0, 60),
    ("circle_pile", 16, 0, 80),
]


@pytest.mark.parametrize("solver_name", wire.SOLVER_NAMES)
@pytest.mark.parametrize("scene,p0,p1,steps", CASES)
def test_reference_world_with_hip_solver(scene, p0, p1, steps, solver_name):
    vel, pos = common.DEFAULT_ITERS[solver_name]
    with hip.Solver(0) as gpu:
        mismatches = []

        def replace(params, bodies, contacts, joints):
            pre = (bodies.copy(), contacts.copy(), joints.copy())
            gpu.solve(params, bodies, contacts, joints)
            order, _ = gpu.contact_order()
            jorder, _ = gpu.joint_order()
            oraclebind.solve(params, *pre, contact_order=order, joint_order=jorder)
            try:
                common.compare_exact((bodies, contacts, joints), pre, "step")
            except AssertionError as e:
                mismatches.append(str(e))
            return 0

        with refbind.RefWorld(scene, solver_name, p0, p1) as wg, refbind.RefWorld(scene, solver_name, p0, p1) as wr:
            with refbind.Replace(replace):
                for _ in range(steps):
                    wg.step(1.0 / 60.0, vel, pos, True)
            for _ in range(steps):
                wr.step(1.0 / 60.0, vel, pos, True)
            assert not mismatches, "%d steps differ from the oracle; first: %s" % (len(mismatches), mismatches[0])

            bg, cg, _ = wg.pack()
            br, cr, _ = wr.pack()
            # contact-pair indices: same live (shapeA, shapeB) set unless the trajectories diverged
            live = br["type"] >= 0
            assert np.isfinite(bg["position"][live]).all()
            if scene in ("pyramid", "joint_grid") and solver_name not in ("XPBD", "TGS_Sticky"):
                # settled / slowly moving scenes: trajectories stay close despite the different
                # Gauss-Seidel order.  Tolerance: 2 cm position, 0.02 rad rotation (sine) after `steps` steps.
                dp = np.abs(bg["position"][live] - br["position"][live]).max()
                dr = np.abs(bg["rot"][live] - br["rot"][live]).max()
                assert dp < 0.02 and dr < 0.02, (dp, dr)
                pa_g, pb_g = wg.contact_pairs()
                pa_r, pb_r = wr.contact_pairs()
                assert sorted(zip(pa_g.tolist(), pb_g.tolist())) == sorted(zip(pa_r.tolist(), pb_r.tolist()))
