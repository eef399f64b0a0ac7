"""Parity link L1: the CPU restatement (oracle/solver_oracle.c) must be BIT-IDENTICAL to the
unmodified reference (oracle/_ref/libs2ref.so) on the same solver inputs, for all ten solvers.

Method: step a reference world through the public s2World_Step with the capture hook armed;
the hook records the wire-format state at s2Solve_* entry and exit.  The oracle is run on a
copy of the entry state and every output word is compared with the exit state.
Needs the compiled reference => skipped where oracle/_ref is absent (the golden-vector test
covers the same pin from committed fixtures).
"""
import pytest

from solver2d_amd import wire
from tests import common, oraclebind, refbind

pytestmark = pytest.mark.skipif(not refbind.available(), reason="oracle/_ref/libs2ref.so not built")

SCENES = [
    ("pyramid", 10, 0, 40),
    ("mixed", 24, 0, 60),
    ("joint_grid", 6, 6, 30),
    ("vertical_stack", 8, 0, 30),
    ("circle_pile", 16, 0, 40),
    ("tumbler", 60, 0, 40),
    ("multi_pyramid", 3, 5, 20),
    # the reference's own edge-case samples (SURVEY.md 8c names Arch; section 4 names the 30 samples as the corpus),
    # restated in solver2d_amd/scenes/scenes.c: wedge hulls under friction, 100..400:1 mass ratios, deep initial overlap,
    # 2 mm cards, coordinates 100 km from the origin (one ulp = 7.8 mm), ragdolls on their joint limits, closed and heavy chains
    ("arch", 0, 0, 60),
    ("high_mass_ratio", 1, 0, 60),
    ("high_mass_ratio", 2, 0, 120),
    ("high_mass_ratio", 3, 0, 120),
    ("overlap_recovery", 0, 0, 40),
    ("card_house", 0, 0, 60),
    ("far_pyramid", 0, 0, 50),
    ("far_stack", 0, 0, 50),
    ("far_recovery", 0, 0, 40),
    ("far_ragdoll_pile", 0, 0, 80),
    ("far_chain", 0, 0, 50),
    ("ragdoll", 0, 0, 100),
    ("ball_and_chain", 40, 0, 60),
    ("bridge", 40, 0, 50),
    # (round 6) the rest of the reference's 26 samples: a lone box, warm-start energy released when the heavy top circle is destroyed at
    # step 120, contacts that mix two shape frictions, 400 circles under applied forces, a toppling domino row, 625 overlapping circles
    # in a closed box, a falling column of circles, ragdolls created and destroyed while the world runs, joints that start a metre open
    ("single_box", 0, 0, 60),
    ("warm_start_energy", 0, 0, 140),
    ("friction_ramp", 0, 0, 120),
    ("rush", 0, 0, 60),
    ("double_domino", 0, 0, 90),
    ("confined", 0, 0, 30),
    ("circle_stack", 0, 0, 120),
    ("ragdoll_stress", 0, 0, 100),
    ("stretched_chain", 0, 0, 60),
]
JOINT_ONLY = ("joint_grid", "far_chain", "ball_and_chain", "bridge", "stretched_chain")


@pytest.mark.parametrize("solver", wire.SOLVER_NAMES)
@pytest.mark.parametrize("scene,p0,p1,steps", SCENES)
def test_bit_exact(solver, scene, p0, p1, steps):
    vel, pos = common.DEFAULT_ITERS[solver]
    with refbind.RefWorld(scene, solver, p0, p1) as world:
        active = 0
        for step in range(steps):
            params, pre, post = world.step_captured(1.0 / 60.0, vel, pos, True)
            got = common.copy3(pre)
            oraclebind.solve(params, *got)
            common.compare_exact(got, post, "%s/%s step %d" % (scene, solver, step))
            active = max(active, int((pre[1]["pointCount"] > 0).sum()))
        if scene not in JOINT_ONLY:
            assert active > 0, "scene produced no contact constraints"


@pytest.mark.parametrize("solver", ["TGS_Soft", "PGS_NGS_Block", "PGS_Soft"])
def test_ragdolls_created_and_destroyed_while_the_world_runs(solver):
    """Ragdoll Stress long enough for the first ragdolls to fall through the funnel and be taken out of the world (steps 456, 396, 367
    under these three solvers): body, shape, joint and contact slots freed and reused mid-run."""
    vel, pos = common.DEFAULT_ITERS[solver]
    with refbind.RefWorld("ragdoll_stress", solver, 0, 0) as world:
        live = []
        for step in range(520):
            params, pre, post = world.step_captured(1.0 / 60.0, vel, pos, True)
            live.append(int((pre[0]["type"] >= 0).sum()))
            if step % 4 == 0 or step > 400:
                got = common.copy3(pre)
                oraclebind.solve(params, *got)
                common.compare_exact(got, post, "ragdoll_stress/%s step %d" % (solver, step))
        assert max(live) > live[0] and any(b < a for a, b in zip(live, live[1:])), "no ragdoll was created, or none destroyed"


@pytest.mark.parametrize("solver", wire.SOLVER_NAMES)
def test_bit_exact_no_warm_start_and_odd_iters(solver):
    with refbind.RefWorld("mixed", solver, 18, 0) as world:
        for step in range(25):
            warm = step % 3 != 0
            vel, pos = (3, 0) if step % 2 else (5, 3)
            params, pre, post = world.step_captured(1.0 / 30.0 if step % 5 == 0 else 1.0 / 60.0, vel, pos, warm)
            got = common.copy3(pre)
            oraclebind.solve(params, *got)
            common.compare_exact(got, post, "mixed/%s step %d" % (solver, step))


def test_larger_pyramid_tgs_soft():
    with refbind.RefWorld("pyramid", "TGS_Soft", 40, 0) as world:
        for step in range(10):
            params, pre, post = world.step_captured(1.0 / 60.0, 8, 4, True)
            got = common.copy3(pre)
            oraclebind.solve(params, *got)
            common.compare_exact(got, post, "pyramid40 step %d" % step)
        assert int((pre[1]["pointCount"] > 0).sum()) == 2380
