"""The headline step as ONE launch: wide_kernel.hip's self-contained variant (S2_WIDE_SELF) prepares its constraints from the wire
contacts, stages its bodies from the wire bodies, runs the whole s2Solve_TGS_Soft (src/solve_tgs_soft.c:138-280) and writes
bodies and impulses back behind a commit counter; its body-centric warm start (S2_WIDE_BODYWARM) replaces the coloured
s2WarmStartContacts sweep (src/solve_common.c:276-330).  Gate as everywhere: the C-ABI result equals the oracle BIT FOR BIT when the
oracle sweeps in the order the library reports.  The step is self-contained from the second resident step on (the first one after
an upload writes manifold.constraintIndex through the prologue launch)."""
import numpy as np
import pytest

from solver2d_amd import hip, synthetic, wire
from tests import common, oraclebind

pytestmark = pytest.mark.gpu


def resident_vs_oracle(s, params, pre, steps, what, launches=None):
    """`steps` consecutive resident steps, every one compared with the oracle swept in the device's order; returns the final state"""
    s.upload(*pre)
    want = common.copy3(pre)
    seen = []
    for step in range(steps):
        s.step_resident(params)
        order, _ = s.contact_order()
        oraclebind.solve(params, *want, contact_order=order)
        got = common.copy3(pre)
        s.download(*got)
        common.compare_exact(got, want, "%s step %d" % (what, step))
        seen.append(s.stats()["kernelLaunches"])
    if launches is not None:
        assert seen[1:] == [launches] * (steps - 1), seen
    return want


TGS = wire.StepParams.make("TGS_Soft", 1.0 / 60.0, 8, 4, True)


ONE = {"self_contained_strips": 1, "strip_body_warm": 1}  # (both are options, off by default: measured no faster than the three-launch step)


@pytest.mark.parametrize("options,launches", [(ONE, 1), ({"self_contained_strips": 1}, 1), ({"strip_body_warm": 1}, 3), ({}, 3),
                                              # (r5: the self-contained form exists for the <3, 2> layout; beside parked rounds it spilled 220-520 bytes
                                              # per lane and was dropped -- such a partition takes the three launches whatever the option says)
                                              ({"persist_debug": 16, "self_contained_strips": 1}, 3), ({"persist_debug": 16}, 3)],
                         ids=["self+bodywarm", "self", "bodywarm", "plain", "parked-self", "parked"])
def test_the_one_launch_step_equals_the_oracle(options, launches):
    with hip.Solver(0) as s:
        s.set_option("strip_patience", 0)
        for k, v in options.items():
            s.set_option(k, v)
        resident_vs_oracle(s, TGS, synthetic.pyramid(100), 5, str(options), launches)
        st = s.stats()
        assert st["persistent"] == 1 and st["pairLanes"] == 2, st


def test_the_headline_world_is_one_launch_per_step():
    """BASELINE configs[1] at full size: LargePyramid base-200, 59,900 constraints."""
    with hip.Solver(0) as s:
        s.set_option("strip_patience", 0)
        for k, v in ONE.items():
            s.set_option(k, v)
        resident_vs_oracle(s, TGS, synthetic.pyramid(200), 3, "base 200", 1)


@pytest.mark.parametrize("iters,warm", [((8, 4), False), ((3, 0), True), ((1, 1), True), ((5, 2), False)])
def test_iteration_shapes_and_cold_start(iters, warm):
    params = wire.StepParams.make("TGS_Soft", 1.0 / 60.0, iters[0], iters[1], warm)
    with hip.Solver(0) as s:
        s.set_option("strip_patience", 0)
        for k, v in ONE.items():
            s.set_option(k, v)
        resident_vs_oracle(s, params, synthetic.pyramid(100), 4, "%s warm=%s" % (iters, warm), 1)


def test_mixed_point_counts_and_empty_manifolds():
    """One-point manifolds, manifolds without points (no-ops wherever the sweep order puts them) and the general kernel variant
    (POINTS == 0: per-point masks in the chain, second-point bits in the body-centric warm start)."""
    pre = common.copy3(synthetic.pyramid(100))
    rng = np.random.default_rng(5)
    live = np.flatnonzero(pre[1]["pointCount"] == 2)
    pre[1]["pointCount"][rng.choice(live, size=400, replace=False)] = 1
    pre[1]["pointCount"][rng.choice(live, size=150, replace=False)] = 0
    for options, launches in ((ONE, 1), ({"self_contained_strips": 1}, 1), ({"strip_body_warm": 1}, 3)):
        with hip.Solver(0) as s:
            s.set_option("strip_patience", 0)
            for k, v in options.items():
                s.set_option(k, v)
            resident_vs_oracle(s, TGS, pre, 4, "mixed point counts %s" % options, launches)


def test_kinematic_and_heavy_bodies_inside_the_pile():
    """Bodies the sweeps do not write (kinematic: infinite mass, moving) are replicas in every strip that touches them; the commit
    counter is what keeps their owner's write-back behind every other workgroup's loads."""
    pre = common.copy3(synthetic.pyramid(100))
    dyn = np.flatnonzero(pre[0]["type"] == wire.BODY_DYNAMIC)
    rng = np.random.default_rng(9)
    for b in rng.choice(dyn, size=6, replace=False):
        pre[0]["type"][b] = wire.BODY_KINEMATIC
        pre[0]["invMass"][b] = 0.0
        pre[0]["invI"][b] = 0.0
        pre[0]["linearVelocity"][b] = (0.05, 0.0)
    with hip.Solver(0) as s:
        s.set_option("strip_patience", 0)
        for k, v in ONE.items():
            s.set_option(k, v)
        resident_vs_oracle(s, TGS, pre, 4, "kinematic bodies in the pile")
        assert s.stats()["persistent"] == 1


def test_a_dead_hand_off_in_the_one_launch_step_leaves_the_wire_arrays_alone():
    """Fault injection once the step IS the one launch: workgroup 1 never publishes its seam bodies.  No workgroup may commit -- the
    resident world stands where it stood --, the host repeats the step on the multi-launch path and stays exact."""
    pre = synthetic.pyramid(100)
    with hip.Solver(0) as s:
        s.set_option("strip_patience", 0)
        s.set_option("persist_spin_limit", 4096)
        for k, v in ONE.items():
            s.set_option(k, v)
        want = resident_vs_oracle(s, TGS, pre, 3, "before the fault", 1)
        s.set_option("persist_debug", 8)
        for step in range(2):
            s.step_resident(TGS)
            order, _ = s.contact_order()
            oraclebind.solve(TGS, *want, contact_order=order)
            got = common.copy3(pre)
            s.download(*got)
            common.compare_exact(got, want, "after the fault, step %d" % step)
            st = s.stats()
            assert st["persistFallbacks"] == 1 and st["persistent"] == 0, st


def test_a_dead_hand_off_under_async_drops_the_steps_behind_it():
    """... and with steps enqueued without a host sync: every launch behind the failed one stands down at its first instruction, the
    world is what it was before the first failed step, the caller repeats the steps."""
    pre = synthetic.pyramid(100)
    with hip.Solver(0) as s:
        s.set_option("strip_patience", 0)
        s.set_option("persist_spin_limit", 4096)
        for k, v in ONE.items():
            s.set_option(k, v)
        want = resident_vs_oracle(s, TGS, pre, 3, "before the fault", 1)
        s.set_option("persist_debug", 8)
        s.set_option("near_handoff", 0)  # (with the same-XCD path on, the first time-out only switches that off: test_gpu_strips.py)
        s.set_option("async", 1)
        for _ in range(3):
            s.step_resident(TGS)
        with pytest.raises(hip.S2AmdError):
            s.synchronize()
        got = common.copy3(pre)
        s.download(*got)
        common.compare_exact(got, want, "the world after the dropped steps")
        for step in range(3):
            s.step_resident(TGS)
            s.synchronize()
            order, _ = s.contact_order()
            oraclebind.solve(TGS, *want, contact_order=order)
        got = common.copy3(pre)
        s.download(*got)
        common.compare_exact(got, want, "the repeated steps")
        assert s.stats()["persistFallbacks"] == 1


def test_four_hundred_one_launch_steps_against_the_multi_launch_path():
    """The hand-off buffers are cleared by the kernel itself now (no epilogue launch): 400 consecutive steps, the bits of the
    multi-launch strip path at every checkpoint."""
    from tests.test_gpu_strips import _resident_states
    checkpoints = {2, 77, 200, 400}
    a = _resident_states(dict(ONE, persist=1), 400, checkpoints)
    b = _resident_states({"persist": 0}, 400, checkpoints)
    c = _resident_states({"persist": 1}, 400, checkpoints)
    for i, ((ba, pa), (bb, pb), (bc, pc)) in enumerate(zip(a, b, c)):
        for f in ("position", "rot", "linearVelocity", "angularVelocity"):
            assert np.array_equal(ba[f].view(np.uint32), bb[f].view(np.uint32)), "checkpoint %d field %s" % (i, f)
            assert np.array_equal(bc[f].view(np.uint32), bb[f].view(np.uint32)), "checkpoint %d field %s (three launches)" % (i, f)
        assert np.array_equal(pa["normalImpulse"].view(np.uint32), pb["normalImpulse"].view(np.uint32)), "checkpoint %d impulses" % i
