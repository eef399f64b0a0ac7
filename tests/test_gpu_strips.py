"""Strips: an island that fits no LDS group is cut along BFS level sets (solver_structure.cpp: partitionStrips);
a Gauss-Seidel sweep becomes phase A (strip interiors) + phase B (seams) = two launches.

Same gate as test_gpu_parity.py: the C-ABI result must equal the oracle BIT FOR BIT when the oracle
sweeps in the order the library reports (strip by strip colour-major, then seam by seam).
"""
import os

import numpy as np
import pytest

from solver2d_amd import hip, synthetic, wire
from tests import common, golden_util, oraclebind
from tests.test_gpu_parity import gpu_vs_oracle, gpu_vs_oracle_loose

pytestmark = pytest.mark.gpu

FILES = golden_util.golden_files()


@pytest.fixture(scope="module")
def tiny_strips():
    """Forces the strip path on the small golden worlds: no island fits a group, strips of ~12 bodies."""
    s = hip.Solver(0)
    s.set_option("strip_patience", 0)
    s.set_option("max_group_bodies", 48)
    s.set_option("strip_min_bodies", 0)
    s.set_option("strip_bodies", 12)
    s.set_option("strips_any_solver", 1)
    yield s
    s.close()


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(p)[:-4] for p in FILES])
def test_golden_inputs_bit_exact_through_strips(tiny_strips, path):
    params, pre, _post = golden_util.load(path)
    gpu_vs_oracle(tiny_strips, params, pre, os.path.basename(path))


def test_golden_worlds_really_ran_as_strips(tiny_strips):
    path = [f for f in FILES if "pyramid10_TGS_Soft_step045" in f][0]
    params, pre, _post = golden_util.load(path)
    gpu_vs_oracle(tiny_strips, params, pre, "pyramid10 strips")
    st = tiny_strips.stats()
    assert st["stripCount"] >= 2 and st["seamCount"] >= 1 and st["groupCount"] == 0


@pytest.mark.parametrize("solver_name", [n for n in wire.SOLVER_NAMES if n != "Jacobi"])
def test_big_pyramid_strips_default_options(solver_name):
    """Base-100 pyramid (5,050 bodies, one island): default options choose strips; three consecutive steps."""
    vel, pos = common.DEFAULT_ITERS[solver_name]
    pre = synthetic.pyramid(100)
    with hip.Solver(0) as s:
        s.set_option("strip_patience", 0)
        s.set_option("strips_any_solver", 1)
        state = common.copy3(pre)
        for step in range(3):
            params = wire.StepParams.make(solver_name, 1.0 / 60.0, vel, pos, True)
            state = gpu_vs_oracle(s, params, state, "pyramid100/%s step %d" % (solver_name, step))
        st = s.stats()
        assert st["stripCount"] >= 4 and st["seamCount"] == st["stripCount"] - 1, st


@pytest.mark.parametrize("solver_name,expect", [("TGS_Soft", True), ("SoftStep", True), ("PGS_Soft", True), ("PGS", True), ("XPBD", True), ("Jacobi", False)])
def test_default_policy_strips_for_every_gauss_seidel_solver(solver_name, expect):
    """Strips pay off through the one-launch kernels: the register-resident ones for the soft contact sweeps (strip_kernel.hip,
    wide_kernel.hip), the op interpreter for every other Gauss-Seidel family (generic_kernel.hip).  s2Solve_Jacobi has no colours
    and keeps its own structure; with the interpreter switched off the non-soft solvers keep the colour batches."""
    vel, pos = common.DEFAULT_ITERS[solver_name]
    pre = synthetic.pyramid(100)
    with hip.Solver(0) as s:
        s.set_option("strip_patience", 0)
        params = wire.StepParams.make(solver_name, 1.0 / 60.0, vel, pos, True)
        state = gpu_vs_oracle(s, params, pre, "pyramid100/%s default" % solver_name)
        assert (s.stats()["stripCount"] > 0) == expect
        # switching the solver on the same resident world
        other = wire.StepParams.make("Jacobi" if expect else "TGS_Soft", 1.0 / 60.0, 4, 2, True)
        gpu_vs_oracle(s, other, state, "pyramid100 switch from %s" % solver_name)
        assert (s.stats()["stripCount"] > 0) == (not expect)
    if expect and solver_name in ("PGS", "XPBD"):
        with hip.Solver(0) as s:
            s.set_option("strip_patience", 0)
            s.set_option("generic", 0)
            gpu_vs_oracle(s, params, pre, "pyramid100/%s without the interpreter" % solver_name)
            assert s.stats()["stripCount"] == 0


@pytest.mark.parametrize("lean", [1, 0])
@pytest.mark.parametrize("solver_name", ["TGS_Soft", "SoftStep", "PGS_Soft"])
def test_lean_strip_kernel_on_off(solver_name, lean):
    """The soft sweeps run through strip_kernel.hip (preloaded rounds, body stages and warm start folded into
    the phase A launch) or through the group interpreter: same bits, fewer launches."""
    vel, pos = common.DEFAULT_ITERS[solver_name]
    pre = synthetic.pyramid(100)
    with hip.Solver(0) as s:
        s.set_option("strip_patience", 0)
        s.set_option("strip_lean", lean)
        state = common.copy3(pre)
        for step in range(3):
            params = wire.StepParams.make(solver_name, 1.0 / 60.0, vel, pos, True)
            state = gpu_vs_oracle(s, params, state, "pyramid100/%s lean=%d step %d" % (solver_name, lean, step))
        print(solver_name, lean, s.stats()["kernelLaunches"])


@pytest.mark.parametrize("persist", [1, 0])
@pytest.mark.parametrize("solver_name", ["TGS_Soft", "SoftStep", "PGS_Soft"])
def test_persistent_strip_step_on_off(solver_name, persist):
    """One persistent launch per step (constraints resident in registers / LDS, seam bodies handed between
    neighbouring workgroups as tagged granules) against the multi-launch strip path: same bits."""
    vel, pos = common.DEFAULT_ITERS[solver_name]
    pre = synthetic.pyramid(100)
    with hip.Solver(0) as s:
        s.set_option("strip_patience", 0)
        s.set_option("persist", persist)
        state = common.copy3(pre)
        for step in range(4):
            params = wire.StepParams.make(solver_name, 1.0 / 60.0, vel, pos, True)
            state = gpu_vs_oracle(s, params, state, "pyramid100/%s persist=%d step %d" % (solver_name, persist, step))
        st = s.stats()
        assert st["persistent"] == persist, st
        if persist:
            assert st["kernelLaunches"] <= 10, st


@pytest.mark.parametrize("seam_regs", [1, 0])
def test_persistent_seams_in_registers_or_lds(seam_regs):
    """TGS_Soft keeps the seam constraints in registers when no seam needs more than two colours (the per-point kernel
    variant); with the option off they are walked from LDS records by the two-point variant.  Same bits."""
    pre = synthetic.pyramid(100)
    params = wire.StepParams.make("TGS_Soft", 1.0 / 60.0, 8, 4, True)
    with hip.Solver(0) as s:
        s.set_option("strip_patience", 0)
        s.set_option("seam_regs", seam_regs)
        state = common.copy3(pre)
        for step in range(3):
            state = gpu_vs_oracle(s, params, state, "pyramid100 seam_regs=%d step %d" % (seam_regs, step))
        assert s.stats()["persistent"] == 1


def test_persistent_two_point_fast_path_follows_the_point_counts():
    """Same contact graph, but manifolds drop from two points to one (a box tilts onto an edge) and come back: the
    persistent kernel's two-point variant is chosen from the point counts of THIS step, not of the step the strips were
    built on."""
    pre = synthetic.pyramid(100)
    params = wire.StepParams.make("TGS_Soft", 1.0 / 60.0, 8, 4, True)
    with hip.Solver(0) as s:
        s.set_option("strip_patience", 0)
        state = gpu_vs_oracle(s, params, common.copy3(pre), "two-point step 0")
        state = gpu_vs_oracle(s, params, state, "two-point step 1")
        active = np.flatnonzero(state[1]["pointCount"] == 2)
        for victim in (active[len(active) // 2], active[7]):
            state[1]["pointCount"][victim] = 1
        state = gpu_vs_oracle(s, params, state, "one-point contacts in the strips")
        assert s.stats()["persistent"] == 1
        state[1]["pointCount"][active[len(active) // 2]] = 2
        state[1]["pointCount"][active[7]] = 2
        state = gpu_vs_oracle(s, params, state, "two points again")
        assert s.stats()["persistent"] == 1


def test_persistent_strip_step_base_200():
    """BASELINE config 2 itself: 20,101 bodies, 59,900 constraints, one island."""
    pre = synthetic.pyramid(200)
    with hip.Solver(0) as s:
        s.set_option("strip_patience", 0)
        params = wire.StepParams.make("TGS_Soft", 1.0 / 60.0, 8, 4, True)
        state = gpu_vs_oracle(s, params, pre, "pyramid200 persistent")
        gpu_vs_oracle(s, params, state, "pyramid200 persistent step 2")
        assert s.stats()["persistent"] == 1


@pytest.mark.parametrize("warm", [True, False])
@pytest.mark.parametrize("iters", [(8, 4), (4, 0), (1, 1)])
@pytest.mark.parametrize("persist", [1, 0])
def test_lean_strip_kernel_iteration_shapes(iters, warm, persist):
    pre = synthetic.pyramid(100)
    with hip.Solver(0) as s:
        s.set_option("strip_patience", 0)
        s.set_option("persist", persist)
        params = wire.StepParams.make("TGS_Soft", 1.0 / 60.0, iters[0], iters[1], warm)
        state = gpu_vs_oracle(s, params, pre, "pyramid100 TGS_Soft %r warm=%r" % (iters, warm))
        gpu_vs_oracle(s, params, state, "pyramid100 TGS_Soft %r warm=%r step 2" % (iters, warm))


def test_jacobi_keeps_the_colour_batch_path():
    vel, pos = common.DEFAULT_ITERS["Jacobi"]
    pre = synthetic.pyramid(100)
    with hip.Solver(0) as s:
        s.set_option("strip_patience", 0)
        params = wire.StepParams.make("Jacobi", 1.0 / 60.0, vel, pos, True)
        gpu_vs_oracle(s, params, pre, "pyramid100/Jacobi")
        assert s.stats()["stripCount"] == 0


@pytest.mark.parametrize("strips", [1, 0])
def test_strips_on_off_launch_counts(strips):
    pre = synthetic.pyramid(100)
    with hip.Solver(0) as s:
        s.set_option("strip_patience", 0)
        s.set_option("strips", strips)
        params = wire.StepParams.make("TGS_Soft", 1.0 / 60.0, 8, 4, True)
        gpu_vs_oracle(s, params, pre, "pyramid100 strips=%d" % strips)
        st = s.stats()
        if strips:
            assert st["kernelLaunches"] < 60, st
        else:
            assert st["stripCount"] == 0 and st["kernelLaunches"] > 100, st


@pytest.mark.parametrize("solver_name", ["TGS_Soft", "PGS_NGS", "XPBD", "TGS_Sticky"])
def test_joint_grid_strips(solver_name):
    """70x70 joint grid (4,900 bodies, 9,660 revolute joints, one island): joints in strips."""
    vel, pos = common.DEFAULT_ITERS[solver_name]
    pre = synthetic.joint_grid(70)
    with hip.Solver(0) as s:
        s.set_option("strip_patience", 0)
        s.set_option("strips_any_solver", 1)
        state = common.copy3(pre)
        for step in range(2):
            params = wire.StepParams.make(solver_name, 1.0 / 60.0, vel, pos, True)
            state = gpu_vs_oracle(s, params, state, "jointgrid70/%s step %d" % (solver_name, step))
        assert s.stats()["stripCount"] >= 4


@pytest.mark.parametrize("solver_name", ["TGS_Soft", "PGS", "PGS_NGS_Block"])
def test_platform_high_degree_body_in_strips(solver_name):
    """The platform world has one body touching 60 boxes: levels around it are wide, still exact."""
    vel, pos = common.DEFAULT_ITERS[solver_name]
    pre = synthetic.platform(60, layers=80)
    with hip.Solver(0) as s:
        s.set_option("strip_patience", 0)
        s.set_option("strips_any_solver", 1)
        params = wire.StepParams.make(solver_name, 1.0 / 60.0, vel, pos, True)
        gpu_vs_oracle(s, params, pre, "platform60x80/%s" % solver_name)


def test_strips_wait_until_the_graph_has_settled():
    """Default policy for the drop-in call: the first solve after a graph change takes the colour-batch path (cheap
    host structure), the next one with the same graph builds the strips.  A destroyed contact lingers as a no-op; a contact
    created between two bodies of one strip (or of the two sides of a seam) takes a free position of one of its rounds
    (solver_incremental.cpp: stripPlace) and the persistent kernel keeps running; one that fits nowhere -- here: between two
    bodies twenty strips apart -- starts over, with the patience doubled because the strip structure had only lived for a step."""
    b, c, j = synthetic.pyramid(100)
    free = np.zeros(2, dtype=wire.contact_dtype)
    free["bodyA"], free["bodyB"], free["constraintIndex"] = -1, -1, -1  # free pool slots, as the reference binding packs them
    pre = (b, np.concatenate([c, free]), j)
    spare = len(c)
    with hip.Solver(0) as s:
        params = wire.StepParams.make("TGS_Soft", 1.0 / 60.0, 8, 4, True)
        state = gpu_vs_oracle(s, params, pre, "patience step 0")
        assert s.stats()["stripCount"] == 0
        state = gpu_vs_oracle(s, params, state, "patience step 1")
        assert s.stats()["stripCount"] > 0 and s.stats()["persistent"] == 1
        # a contact is destroyed: its entry lingers in the structure as a no-op, nothing is rebuilt
        record = state[1][7].copy()
        state[1][7] = free[0]
        state = gpu_vs_oracle_loose(s, params, state, "patience step 2")
        assert s.stats()["persistent"] == 1 and s.stats()["hostPrepMs"] == 0.0
        # a contact is CREATED between neighbours (here: the same pair, in another pool slot): a free position of its strip
        placed = s.stats()["placedContacts"]
        state[1][spare] = record
        state = gpu_vs_oracle_loose(s, params, state, "patience step 3")
        assert s.stats()["persistent"] == 1 and s.stats()["hostPrepMs"] == 0.0 and s.stats()["placedContacts"] == placed + 1
        # a contact between bodies of strips far apart has no place in the strip tables: new graph
        order, _ = s.contact_order()
        far = record.copy()
        far["bodyA"], far["bodyB"] = int(state[1][order[0]]["bodyB"]), int(state[1][order[-1]]["bodyB"])
        state[1][spare + 1] = far
        state = gpu_vs_oracle_loose(s, params, state, "patience step 4")
        assert s.stats()["stripCount"] == 0
        state = gpu_vs_oracle_loose(s, params, state, "patience step 5")
        assert s.stats()["stripCount"] == 0
        gpu_vs_oracle_loose(s, params, state, "patience step 6")
        assert s.stats()["persistent"] == 1


@pytest.mark.parametrize("solver_name", ["TGS_Soft", "SoftStep", "PGS_Soft"])
def test_manifolds_that_lose_and_regain_their_points_cost_the_host_nothing(solver_name):
    """The structure covers every POTENTIAL constraint (a contact slot with two live bodies), with or without manifold
    points: when manifolds of a base-100 pyramid lose their points and get them back -- what the narrow phase does to a
    breathing pile every step -- the strips, the persistent kernel and the captured step graph all stay, no host structure
    time is spent, and every step is still bit-exact against the oracle (which, like the reference, only gathers the
    manifolds WITH points)."""
    vel, pos = common.DEFAULT_ITERS[solver_name]
    pre = synthetic.pyramid(100)
    params = wire.StepParams.make(solver_name, 1.0 / 60.0, vel, pos, True)
    rng = np.random.default_rng(11)
    with hip.Solver(0) as s:
        s.set_option("strip_patience", 0)
        s.set_option("graph_min_launches", 0)  # (by default a step of three launches is enqueued directly, never captured)
        state = gpu_vs_oracle(s, params, pre, "flip warm-up 0")
        state = gpu_vs_oracle(s, params, state, "flip warm-up 1")  # step graph captured
        assert s.stats()["persistent"] == 1
        full = state[1]["pointCount"].copy()
        for step in range(6):
            off = rng.choice(len(full), size=40 + 30 * step, replace=False)
            state[1]["pointCount"][:] = full
            state[1]["pointCount"][off] = 0
            one = rng.choice(np.setdiff1d(np.arange(len(full)), off), size=25, replace=False)
            state[1]["pointCount"][one] = 1  # and some keep one of their two points
            state = gpu_vs_oracle_loose(s, params, state, "flip step %d" % step)
            st = s.stats()
            assert st["persistent"] == 1 and st["stripCount"] > 1 and st["hostPrepMs"] == 0.0, st
            assert st["constraintCount"] == int((state[1]["pointCount"] > 0).sum())
            if step >= 2:  # (the first flip leaves the all-two-points kernel variant: one new launch sequence, seen, captured, replayed)
                assert st["graphReplayed"] == 1, st


def _resident_states(options, steps, checkpoints, base=120, concurrent=None):
    """Body arrays of a resident base-`base` pyramid after every checkpoint step under `options`."""
    pre = synthetic.pyramid(base)
    params = wire.StepParams.make("TGS_Soft", 1.0 / 60.0, 8, 4, True)
    out = []
    with hip.Solver(0) as s:
        for k, v in options.items():
            s.set_option(k, v)
        s.set_option("strip_patience", 0)
        s.upload(*pre)
        for step in range(1, steps + 1):
            s.step_resident(params)
            if concurrent is not None:
                concurrent()
            if step in checkpoints:
                b, c, _j = common.copy3(pre)
                s.download(b, c, _j)
                out.append((b.copy(), c["points"].copy()))
        assert s.stats()["persistent"] == options.get("persist", 1)
    return out


def test_persistent_hand_offs_hold_over_hundreds_of_steps():
    """The granule hand-offs are a timing-dependent protocol: 400 consecutive resident steps (6,400 exchanges per
    workgroup pair) through the persistent kernel must give the bits of the multi-launch strip path at every
    checkpoint -- a stale or torn hand-off would show up as a diverging pile."""
    checkpoints = {1, 50, 137, 250, 400}
    a = _resident_states({"persist": 1}, 400, checkpoints)
    b = _resident_states({"persist": 0}, 400, checkpoints)
    for i, ((ba, pa), (bb, pb)) in enumerate(zip(a, b)):
        for f in ("position", "rot", "linearVelocity", "angularVelocity"):
            assert np.array_equal(ba[f].view(np.uint32), bb[f].view(np.uint32)), "checkpoint %d field %s" % (i, f)
        assert np.array_equal(pa["normalImpulse"].view(np.uint32), pb["normalImpulse"].view(np.uint32)), "checkpoint %d impulses" % i


def test_persistent_kernel_next_to_a_busy_neighbour():
    """Uneven load: a second solver keeps its own persistent kernel (and its prologue / epilogue launches) in flight
    on another stream of the same GPU while the first one steps; hand-offs must not depend on having the chip alone."""
    noisy = hip.Solver(0)
    try:
        noisy.set_option("strip_patience", 0)
        noisy.set_option("async", 1)
        big = synthetic.pyramid(150)
        noisy.upload(*big)
        nparams = wire.StepParams.make("TGS_Soft", 1.0 / 60.0, 8, 4, True)

        def poke():
            noisy.step_resident(nparams)
            noisy.step_resident(nparams)

        checkpoints = {1, 40, 120}
        a = _resident_states({"persist": 1}, 120, checkpoints, concurrent=poke)
        noisy.synchronize()
        b = _resident_states({"persist": 0}, 120, checkpoints)
        for i, ((ba, pa), (bb, pb)) in enumerate(zip(a, b)):
            for f in ("position", "rot", "linearVelocity", "angularVelocity"):
                assert np.array_equal(ba[f].view(np.uint32), bb[f].view(np.uint32)), "checkpoint %d field %s" % (i, f)
    finally:
        noisy.close()


@pytest.mark.parametrize("base,strip_bodies", [(100, 8), (100, 320), (72, 200)])
def test_wide_kernel_with_parked_seam_rounds(base, strip_bodies):
    """wide_kernel.hip keeps two seam records per lane in registers; a partition whose seams need a third or fourth colour next to
    seven or eight interior ones parks those rounds in LDS (variants <3,2,2> / <4,2,2>).  persist_debug 16 forces them on
    partitions that do not need them: same order, same bits as the register variants."""
    pre = synthetic.pyramid(base)
    params = wire.StepParams.make("TGS_Soft", 1.0 / 60.0, 8, 4, True)
    results = []
    for park in (0, 16):
        with hip.Solver(0) as s:
            s.set_option("strip_patience", 0)
            s.set_option("strip_min_bodies", 0)
            s.set_option("strip_retry", 0)
            s.set_option("strip_bodies", strip_bodies)
            s.set_option("persist_debug", park)
            state = common.copy3(pre)
            for step in range(3):
                state = gpu_vs_oracle(s, params, state, "pyramid%d parked=%d step %d" % (base, park, step))
            st = s.stats()
            assert st["persistent"] == 1 and st["pairLanes"] == 2, st
            results.append(state)
    common.compare_exact(results[0], results[1], "parked seam rounds vs registers")


@pytest.mark.parametrize("extra", [1, 2])
def test_wide_kernel_with_parked_interior_rounds(extra):
    """A body with seven or eight constraints inside a strip (a ball in the pile) forces a seventh / eighth interior colour on that
    strip: the 512-thread kernel keeps six interior rounds in registers and parks the rest in LDS (variant <3,2,2,2>).  Built here by
    giving one box of the pyramid `extra` more contacts with boxes two hops away; same bits as the oracle and as the 256-thread kernel."""
    b, c, j = common.copy3(synthetic.pyramid(100))
    a0, b0 = c["bodyA"].astype(int), c["bodyB"].astype(int)
    nbrs = {}
    for x, y in zip(a0.tolist(), b0.tolist()):
        nbrs.setdefault(x, set()).add(y), nbrs.setdefault(y, set()).add(x)
    hub = next(i for i in range(len(b) // 2, len(b)) if len(nbrs.get(i, ())) == 6 and b["invMass"][i] > 0)
    two_hops = sorted({t for n in nbrs[hub] for t in nbrs[n] if t != hub and t not in nbrs[hub] and b["invMass"][t] > 0})
    added = c[:extra].copy()
    template = int(np.flatnonzero((a0 == hub) | (b0 == hub))[0])
    for e in range(extra):
        added[e] = c[template]
        added[e]["bodyA"], added[e]["bodyB"] = hub, two_hops[e]
    pre = (b, np.concatenate([c, added]), j)
    params = wire.StepParams.make("TGS_Soft", 1.0 / 60.0, 8, 4, True)
    results = []
    for wide in (1, 0):
        with hip.Solver(0) as s:
            s.set_option("strip_patience", 0)
            s.set_option("wide", wide)
            state = common.copy3(pre)
            for step in range(3):
                state = gpu_vs_oracle(s, params, state, "hub of degree %d, wide=%d step %d" % (6 + extra, wide, step))
            st = s.stats()
            assert st["persistent"] == 1 and st["pairLanes"] == (2 if wide else 0), st
            results.append(state)
    common.compare_exact(results[0], results[1], "parked interior rounds vs the 256-thread kernel")


def test_a_dead_hand_off_falls_back_to_the_multi_launch_path():
    """Fault injection: workgroup 1 of the persistent kernel never publishes its seam bodies (what a non-resident
    workgroup looks like to its neighbours).  The polls give up, the epilogue leaves the wire arrays alone, the host
    repeats the step on the multi-launch strip path -- the caller sees a correct, bit-exact step and a counter."""
    pre = synthetic.pyramid(100)
    params = wire.StepParams.make("TGS_Soft", 1.0 / 60.0, 8, 4, True)
    with hip.Solver(0) as s:
        s.set_option("strip_patience", 0)
        s.set_option("persist_spin_limit", 4096)
        s.set_option("persist_debug", 8)
        state = gpu_vs_oracle(s, params, pre, "dead hand-off, step 0")
        st = s.stats()
        assert st["persistFallbacks"] == 1 and st["persistent"] == 0 and st["stripCount"] > 0, st
        # the solver stays on the multi-launch path (no second time-out) and stays exact
        state = gpu_vs_oracle(s, params, state, "dead hand-off, step 1")
        assert s.stats()["persistFallbacks"] == 1 and s.stats()["persistent"] == 0
    # resident stepping: same behaviour
    with hip.Solver(0) as s:
        s.set_option("strip_patience", 0)
        s.set_option("persist_spin_limit", 4096)
        s.set_option("persist_debug", 8)
        s.upload(*pre)
        s.step_resident(params)
        assert s.stats()["persistFallbacks"] == 1
        got = common.copy3(pre)
        s.download(*got)
        order, _ = s.contact_order()
        want = common.copy3(pre)
        oraclebind.solve(params, *want, contact_order=order)
        common.compare_exact(got, want, "dead hand-off, resident")


@pytest.mark.parametrize("solver_name", ["TGS_Soft", "PGS_NGS_Block"])
def test_a_solver_that_lost_a_hand_off_tries_the_persistent_kernel_again_later(solver_name):
    """`persistFailed` is not for ever: after "persist_retry" steps on the fallback path the one-launch kernels get another chance
    (the GPU may have been shared only for a while); a retry that times out again doubles the wait.  Every step stays exact."""
    vel, pos = common.DEFAULT_ITERS[solver_name]
    pre = synthetic.pyramid(100)
    params = wire.StepParams.make(solver_name, 1.0 / 60.0, vel, pos, True)
    with hip.Solver(0) as s:
        s.set_option("strip_patience", 0)
        s.set_option("persist_spin_limit", 4096)
        s.set_option("persist_retry", 3)
        s.set_option("persist_debug", 8)
        state = gpu_vs_oracle(s, params, pre, "retry: the fault")
        assert s.stats()["persistFallbacks"] == 1 and s.stats()["persistent"] == 0
        history = []
        for step in range(5):  # the fault persists: one more time-out at the retry, then a doubled wait
            state = gpu_vs_oracle(s, params, state, "retry: fault still there, step %d" % step)
            history.append((s.stats()["persistent"], s.stats()["persistFallbacks"]))
        assert history[-1] == (0, 2) and [h[1] for h in history].count(1) >= 2, history
        s.set_option("persist_debug", 0)  # the other tenant has left
        seen = []
        for step in range(8):
            state = gpu_vs_oracle(s, params, state, "retry: fault gone, step %d" % step)
            seen.append(s.stats()["persistent"])
        assert seen[-1] == 1 and 0 in seen and s.stats()["persistFallbacks"] == 2, seen


def test_a_dead_hand_off_next_to_resident_islands_leaves_their_contacts_alone():
    """The resident-island kernels write their impulses (and, alone in a world, their bodies) straight into the wire arrays.  In a
    step that also holds a persistent strip launch -- which may lose a hand-off, whereupon the step is repeated from untouched wire
    arrays -- the small islands therefore run on the group interpreter (SoA arrays, carried over by the epilogue, which stands down
    after a failure): the repeated step must not warm-start from impulses of the dropped one.  (They run BEFORE the strips: a strip
    may own a kinematic body an island reads.)"""
    big = synthetic.pyramid(100)
    small = synthetic.pyramid(12, count=6)
    shift = len(big[0])
    sb, sc, sj = common.copy3(small)
    sb["position"][:, 0] += 400.0
    live = sc["bodyA"] >= 0
    sc["bodyA"][live] += shift
    sc["bodyB"][live] += shift
    pre = (np.concatenate([big[0], sb]), np.concatenate([big[1], sc]), np.concatenate([big[2], sj]))
    params = wire.StepParams.make("TGS_Soft", 1.0 / 60.0, 8, 4, True)
    with hip.Solver(0) as s:
        s.set_option("strip_patience", 0)
        s.set_option("persist_spin_limit", 4096)
        s.set_option("persist_debug", 8)
        state = gpu_vs_oracle(s, params, pre, "dead hand-off next to resident islands, step 0")
        st = s.stats()
        assert st["persistFallbacks"] == 1 and st["stripCount"] > 0 and st["groupCount"] > 0, st
        gpu_vs_oracle(s, params, state, "dead hand-off next to resident islands, step 1")
    with hip.Solver(0) as s:  # and without the fault: both kinds of island in their one-launch kernels
        s.set_option("strip_patience", 0)
        state = gpu_vs_oracle(s, params, pre, "big island + resident islands")
        st = s.stats()
        assert st["persistent"] == 1 and st["groupCount"] > 0 and st["kernelLaunches"] <= 5, st


def test_a_dead_hand_off_under_async_is_reported_and_the_solver_recovers():
    """The same fault with option "async": steps are enqueued without a host sync, the failure surfaces at
    s2amd_synchronize.  Contract: the call fails with a device error, the resident world stands where it stood before the
    first failed step (every later epilogue saw the flag too), both error words are cleared and the solver stays on the
    multi-launch strip path -- the caller repeats the dropped steps and gets bit-exact results."""
    pre = synthetic.pyramid(100)
    params = wire.StepParams.make("TGS_Soft", 1.0 / 60.0, 8, 4, True)
    with hip.Solver(0) as s:
        s.set_option("strip_patience", 0)
        s.set_option("persist_spin_limit", 4096)
        s.set_option("persist_debug", 8)
        s.set_option("near_handoff", 0)  # (with the same-XCD path on, the first time-out only switches that off: the test below)
        s.set_option("async", 1)
        s.upload(*pre)
        for _ in range(3):
            s.step_resident(params)
        with pytest.raises(hip.S2AmdError):
            s.synchronize()
        assert s.stats()["persistFallbacks"] == 1
        got = common.copy3(pre)
        s.download(*got)
        for f in ("position", "rot", "linearVelocity", "angularVelocity"):
            assert np.array_equal(got[0][f].view(np.uint32), pre[0][f].view(np.uint32)), f
        assert got[1]["points"].tobytes() == pre[1]["points"].tobytes()
        # the caller repeats the three steps: no second time-out, results exact
        want = common.copy3(pre)
        for step in range(3):
            s.step_resident(params)
            s.synchronize()
            assert s.stats()["persistent"] == 0 and s.stats()["persistFallbacks"] == 1
            order, _ = s.contact_order()
            oraclebind.solve(params, *want, contact_order=order)
        got = common.copy3(pre)
        s.download(*got)
        common.compare_exact(got, want, "async dead hand-off, repeated steps")


@pytest.mark.parametrize("slack", [0, 1])
def test_strip_patience_backs_off_when_the_graph_keeps_changing(slack):
    """The strip structure takes milliseconds of host time: when it dies young (the graph changes again within a few
    steps) the patience doubles, so a world that keeps changing stays on the colour batches (strip_slack 0).  With free
    positions in the strips' rounds (the default) these changes -- a pair of neighbours re-created in another pool slot -- are
    placed into the strips and nothing is rebuilt at all."""
    pre = synthetic.pyramid(100)
    params = wire.StepParams.make("TGS_Soft", 1.0 / 60.0, 8, 4, True)
    free = np.zeros(1, dtype=wire.contact_dtype)
    free["bodyA"], free["bodyB"], free["constraintIndex"] = -1, -1, -1
    pre = (pre[0], np.concatenate([pre[1], free]), pre[2])
    slots = [int(np.flatnonzero(pre[1]["pointCount"] == 2)[5]), len(pre[1]) - 1]
    with hip.Solver(0) as s:  # default patience 1
        s.set_option("strip_slack", slack)
        state = common.copy3(pre)
        strips_seen = []
        for step in range(14):
            if step % 3 == 2:  # every third step a contact is destroyed and created again in another pool slot (destruction alone -- or a
                # manifold that merely loses its points -- is no graph change)
                src, dst = (slots[0], slots[1]) if state[1]["pointCount"][slots[0]] else (slots[1], slots[0])
                state[1][dst] = state[1][src]
                state[1][src] = free[0]
            state = gpu_vs_oracle_loose(s, params, state, "churn step %d" % step)
            strips_seen.append(s.stats()["stripCount"] > 0)
        if slack:
            assert all(strips_seen[1:]) and s.stats()["structureBuilds"] <= 3 and s.stats()["placedContacts"] >= 4, (strips_seen, s.stats())
        else:
            # built once or twice at the start, then the patience (2, 4, 8 ...) outlasts the three quiet steps
            assert any(strips_seen[:6]) and not any(strips_seen[8:]), strips_seen


@pytest.mark.parametrize("mode", ["wide", "pair", "one"])
@pytest.mark.parametrize("solver_name", ["TGS_Soft", "SoftStep", "PGS_Soft"])
def test_persistent_step_kernels_agree(solver_name, mode):
    """The three persistent step kernels on the same tables, in the same sweep order, to the same bits as the oracle:
    wide_kernel.hip (512 threads per strip: TGS_Soft and, since round 4, PGS_Soft and SoftStep), pair_kernel.hip (lanes 2c / 2c+1 are body A's / body B's side of
    constraint c, differences cross as DPP operands) and strip_kernel.hip (256 threads, one lane per constraint)."""
    vel, pos = common.DEFAULT_ITERS[solver_name]
    pre = synthetic.pyramid(100)
    with hip.Solver(0) as s:
        s.set_option("strip_patience", 0)
        s.set_option("wide", 1 if mode == "wide" else 0)
        s.set_option("pair_lanes", 1 if mode == "pair" else 0)
        state = common.copy3(pre)
        for step in range(4):
            params = wire.StepParams.make(solver_name, 1.0 / 60.0, vel, pos, True)
            state = gpu_vs_oracle(s, params, state, "pyramid100/%s %s step %d" % (solver_name, mode, step))
        st = s.stats()
        expect = {"wide": 2, "pair": 1, "one": 0}[mode]
        assert st["persistent"] == 1 and st["pairLanes"] == expect, st


@pytest.mark.parametrize("mode", ["wide", "pair"])
@pytest.mark.parametrize("iters,warm", [((8, 4), False), ((3, 0), True), ((1, 1), True), ((5, 2), False)])
def test_persistent_step_kernels_iteration_shapes_and_cold_start(iters, warm, mode):
    pre = synthetic.pyramid(100)
    with hip.Solver(0) as s:
        s.set_option("strip_patience", 0)
        s.set_option("wide", 1 if mode == "wide" else 0)
        s.set_option("pair_lanes", 1 if mode == "pair" else 0)
        state = common.copy3(pre)
        for step in range(3):
            params = wire.StepParams.make("TGS_Soft", 1.0 / 60.0, iters[0], iters[1], warm)
            state = gpu_vs_oracle(s, params, state, "%s %s warm=%s step %d" % (mode, iters, warm, step))
        assert s.stats()["pairLanes"] == (2 if mode == "wide" else 1)


@pytest.mark.parametrize("iters,warm", [((4, 2), True), ((3, 0), True), ((1, 1), False), ((8, 4), False)])
@pytest.mark.parametrize("base", [100, 72])
def test_pgs_soft_on_the_wide_kernel(base, iters, warm):
    """s2Solve_PGS_Soft (solve_pgs_soft.c:127-245) on the 512-thread strip kernel: the 22-dword record holds rA0 / rB0 (as perp, the form
    the sweep multiplies with) and the separation of s2PrepareContacts_Soft instead of the local anchors and the adjusted separation;
    iteration shapes, cold start, a pyramid whose strips are few."""
    pre = synthetic.pyramid(base)
    with hip.Solver(0) as s:
        s.set_option("strip_patience", 0)
        state = common.copy3(pre)
        for step in range(4):
            params = wire.StepParams.make("PGS_Soft", 1.0 / 60.0, iters[0], iters[1], warm)
            state = gpu_vs_oracle(s, params, state, "PGS_Soft wide base %d %s warm=%s step %d" % (base, iters, warm, step))
        st = s.stats()
        if base >= 100:  # (the smaller pyramid takes whatever the structure chooses for it: the parity above is the test)
            assert st["persistent"] == 1 and st["pairLanes"] == 2, {k: st[k] for k in ("persistent", "pairLanes", "stripCount", "groupCount")}


def test_pgs_soft_wide_kernel_takes_single_point_and_pointless_manifolds():
    """The general (POINTS == 0) variant: manifolds that lose one point or both and get them back, strips kept."""
    pre = synthetic.pyramid(100)
    params = wire.StepParams.make("PGS_Soft", 1.0 / 60.0, 4, 2, True)
    rng = np.random.default_rng(5)
    with hip.Solver(0) as s:
        s.set_option("strip_patience", 0)
        state = gpu_vs_oracle(s, params, common.copy3(pre), "PGS_Soft flip warm-up 0")
        state = gpu_vs_oracle(s, params, state, "PGS_Soft flip warm-up 1")
        full = state[1]["pointCount"].copy()
        for step in range(4):
            off = rng.choice(len(full), size=30 + 20 * step, replace=False)
            state[1]["pointCount"][:] = full
            state[1]["pointCount"][off] = 0
            one = rng.choice(np.setdiff1d(np.arange(len(full)), off), size=25, replace=False)
            state[1]["pointCount"][one] = 1
            state = gpu_vs_oracle_loose(s, params, state, "PGS_Soft flip step %d" % step)
            st = s.stats()
            assert st["persistent"] == 1 and st["pairLanes"] == 2 and st["stripCount"] > 1, st


@pytest.mark.parametrize("iters,warm", [((8, 4), True), ((3, 0), True), ((1, 1), False), ((5, 2), False)])
@pytest.mark.parametrize("base", [100, 72])
def test_softstep_on_the_wide_kernel(base, iters, warm):
    """s2Solve_SoftStep (solve_soft_step.c:182-310) on the 512-thread strip kernel: the TGS_Soft record in registers (the separation is
    measured from the anchors as the bodies stand), rA0 / rB0 -- what s2WarmStartContacts_Fixed and s2SolveContacts_TGS_Fixed push
    along -- in LDS; a warm start in every substep."""
    pre = synthetic.pyramid(base)
    with hip.Solver(0) as s:
        s.set_option("strip_patience", 0)
        state = common.copy3(pre)
        for step in range(4):
            params = wire.StepParams.make("SoftStep", 1.0 / 60.0, iters[0], iters[1], warm)
            state = gpu_vs_oracle(s, params, state, "SoftStep wide base %d %s warm=%s step %d" % (base, iters, warm, step))
        st = s.stats()
        if base >= 100:
            assert st["persistent"] == 1 and st["pairLanes"] == 2, {k: st[k] for k in ("persistent", "pairLanes", "stripCount", "groupCount")}


def test_softstep_wide_kernel_takes_single_point_and_pointless_manifolds():
    pre = synthetic.pyramid(100)
    params = wire.StepParams.make("SoftStep", 1.0 / 60.0, 8, 4, True)
    rng = np.random.default_rng(6)
    with hip.Solver(0) as s:
        s.set_option("strip_patience", 0)
        state = gpu_vs_oracle(s, params, common.copy3(pre), "SoftStep flip warm-up 0")
        state = gpu_vs_oracle(s, params, state, "SoftStep flip warm-up 1")
        full = state[1]["pointCount"].copy()
        for step in range(4):
            off = rng.choice(len(full), size=30 + 20 * step, replace=False)
            state[1]["pointCount"][:] = full
            state[1]["pointCount"][off] = 0
            one = rng.choice(np.setdiff1d(np.arange(len(full)), off), size=25, replace=False)
            state[1]["pointCount"][one] = 1
            state = gpu_vs_oracle_loose(s, params, state, "SoftStep flip step %d" % step)
            st = s.stats()
            assert st["persistent"] == 1 and st["pairLanes"] == 2 and st["stripCount"] > 1, st


def test_softstep_base_200_on_the_wide_kernel():
    """BASELINE config 2's world under s2Solve_SoftStep: 99 thin strips, rA0 / rB0 of five records per lane in LDS (80 KB)."""
    pre = synthetic.pyramid(200)
    params = wire.StepParams.make("SoftStep", 1.0 / 60.0, 8, 4, True)
    with hip.Solver(0) as s:
        s.set_option("strip_patience", 0)
        state = common.copy3(pre)
        for step in range(3):
            state = gpu_vs_oracle(s, params, state, "SoftStep base 200 step %d" % step)
        st = s.stats()
        assert st["persistent"] == 1 and st["pairLanes"] == 2 and st["kernelLaunches"] <= 3, st


@pytest.mark.parametrize("solver_name", ["SoftStep", "PGS_Soft"])
def test_soft_solvers_on_partitions_that_need_parked_rounds(solver_name):
    """The wide kernel's parked variants (persist_debug 16 forces them): PGS_Soft has them; SoftStep does not (rA0 / rB0 of parked
    records have no room in LDS) and takes the 256-thread kernel on the same thin strips -- either way one persistent launch, same bits
    as the oracle and as the run without parking."""
    vel, pos = common.DEFAULT_ITERS[solver_name]
    pre = synthetic.pyramid(100)
    params = wire.StepParams.make(solver_name, 1.0 / 60.0, vel, pos, True)
    results = []
    for park in (0, 16):
        with hip.Solver(0) as s:
            s.set_option("strip_patience", 0)
            s.set_option("persist_debug", park)
            state = common.copy3(pre)
            for step in range(3):
                state = gpu_vs_oracle(s, params, state, "%s parked=%d step %d" % (solver_name, park, step))
            st = s.stats()
            assert st["persistent"] == 1, st
            assert st["pairLanes"] == (2 if (park == 0 or solver_name == "PGS_Soft") else st["pairLanes"]), st
            if park and solver_name == "SoftStep":
                assert st["pairLanes"] != 2, st
            results.append(state)
    common.compare_exact(results[0], results[1], "parked rounds vs registers")


def _pyramid_with_free_bodies(base, balls, spare_slots):
    """The pyramid plus `balls` dynamic bodies that touch nothing (far to the right, no gravity) and `spare_slots` free contact slots."""
    b, c, j = common.copy3(synthetic.pyramid(base))
    extra = np.zeros(balls, dtype=wire.body_dtype)
    template = b[int(np.flatnonzero(b["invMass"] > 0)[0])]
    for i in range(balls):
        extra[i] = template
        extra[i]["position"] = (float(base + 40 + 3 * i), 5.0)
        extra[i]["linearVelocity"] = (0.0, 0.0)
        extra[i]["gravityScale"] = 0.0
    free = np.zeros(spare_slots, dtype=wire.contact_dtype)
    free["bodyA"], free["bodyB"], free["constraintIndex"] = -1, -1, -1
    return np.concatenate([b, extra]), np.concatenate([c, free]), j


def _touch(contacts, slot, template_slot, a, b):
    """Slot `slot` becomes a two-point manifold between bodies a and b (the geometry of an existing manifold: a solver input)."""
    contacts[slot] = contacts[template_slot]
    contacts[slot]["bodyA"], contacts[slot]["bodyB"] = a, b
    for p in range(2):
        contacts[slot]["points"][p]["normalImpulse"] = 0.0
        contacts[slot]["points"][p]["tangentImpulse"] = 0.0


@pytest.mark.parametrize("solver_name", ["TGS_Soft", "SoftStep", "PGS_Soft"])
def test_a_body_that_joins_the_island_moves_to_the_strip_it_touches(solver_name):
    """SURVEY.md 8f row 4, the wrecking ball: a dynamic body without constraints is owned by whichever strip the build put it in.  Its first
    contact -- with a box in the middle of the pile, many strips away -- moves it there (the receiving strip's body list is written
    again with one more entry, the imports' slots shift) and the contact is placed as an interior one: NO structure build, the step stays
    on the persistent kernel, same bits as the oracle.  Then a second contact with a box of the same strip, both destroyed, and a
    contact with a box in another strip: the body is free again and moves again."""
    vel, pos = common.DEFAULT_ITERS[solver_name]
    base = 100
    pre = _pyramid_with_free_bodies(base, 2, 4)
    params = wire.StepParams.make(solver_name, 1.0 / 60.0, vel, pos, True)
    nb = len(pre[0])
    ball, ball2 = nb - 2, nb - 1
    template = int(np.flatnonzero(pre[1]["pointCount"] == 2)[len(pre[1]) // 3])
    spare = [len(pre[1]) - 4 + i for i in range(4)]
    dyn = np.flatnonzero(pre[0]["invMass"][: nb - 2] > 0)
    mid, far = int(dyn[len(dyn) // 2]), int(dyn[len(dyn) // 5])
    with hip.Solver(0) as s:
        s.set_option("strip_patience", 0)
        state = common.copy3(pre)
        for step in range(3):
            state = gpu_vs_oracle_loose(s, params, state, "join warm-up %d" % step)
        st = s.stats()
        assert st["persistent"] == 1 and st["pairLanes"] == 2, st
        builds, placed = st["structureBuilds"], st["placedContacts"]
        script = {0: [(spare[0], ball, mid)], 2: [(spare[1], mid + 1, ball)], 3: [(spare[2], ball2, far)], 5: "free", 7: [(spare[3], far + 2, ball)]}
        for step in range(10):
            what = script.get(step)
            if what == "free":
                for sl in spare[:2]:
                    state[1][sl]["bodyA"], state[1][sl]["bodyB"], state[1][sl]["pointCount"] = -1, -1, 0
            elif what:
                for sl, a, b in what:
                    _touch(state[1], sl, template, a, b)
            state = gpu_vs_oracle_loose(s, params, state, "%s join step %d" % (solver_name, step))
            st = s.stats()
            assert st["persistent"] == 1 and st["pairLanes"] == 2 and st["structureBuilds"] == builds, (step, {k: st[k] for k in ("persistent", "pairLanes", "structureBuilds", "placedContacts")})
        assert s.stats()["placedContacts"] >= placed + 4 and s.stats()["bodiesAdopted"] >= 3, s.stats()


def test_joining_bodies_beyond_the_slack_rebuild():
    """A strip takes S2_STRIP_ADOPT_SLACK (8) bodies this way; the ninth forces the rebuild it always did -- still bit-exact."""
    base = 100
    balls = 10
    pre = _pyramid_with_free_bodies(base, balls, balls)
    params = wire.StepParams.make("TGS_Soft", 1.0 / 60.0, 8, 4, True)
    nb = len(pre[0])
    template = int(np.flatnonzero(pre[1]["pointCount"] == 2)[len(pre[1]) // 3])
    dyn = np.flatnonzero(pre[0]["invMass"][: nb - balls] > 0)
    mid = int(dyn[len(dyn) // 2])
    with hip.Solver(0) as s:
        s.set_option("strip_patience", 0)
        state = common.copy3(pre)
        for step in range(2):
            state = gpu_vs_oracle_loose(s, params, state, "slack warm-up %d" % step)
        builds = s.stats()["structureBuilds"]
        for i in range(balls):
            _touch(state[1], len(pre[1]) - balls + i, template, nb - balls + i, mid)
            state = gpu_vs_oracle_loose(s, params, state, "slack join %d" % i)
            if i < 6:  # (every ball is a seventh, eighth ... constraint on `mid`: the strip's rounds run out before its body slack does)
                assert s.stats()["structureBuilds"] == builds or i >= 2, (i, s.stats())
        assert s.stats()["structureBuilds"] > builds


def _box(base, i, j):
    """Body index of the box in row i (from the ground), column j of synthetic.pyramid(base) (i <= j < base)."""
    return 1 + sum(base - r for r in range(i)) + (j - i)


def _strip_picture(s, state):
    """(owner, seam, neighbours): strip of every body, the seam that carries it (-1), dynamic contact neighbours per body."""
    owner, seam, count = s.strip_owners(len(state[0]))
    nbrs = {}
    live = state[1]["bodyA"] >= 0
    for a, b in zip(state[1]["bodyA"][live].tolist(), state[1]["bodyB"][live].tolist()):
        nbrs.setdefault(a, set()).add(b), nbrs.setdefault(b, set()).add(a)
    return owner, seam, count, nbrs


@pytest.mark.parametrize("solver_name", ["TGS_Soft", "SoftStep", "PGS_Soft"])
def test_a_joined_body_that_touches_the_neighbouring_strip_extends_the_seam(solver_name):
    """The ball, adopted by the strip of the first box it touched, then touches a box of the NEXT strip: the seam between the two has to
    carry the ball from now on -- one more export of its strip, one more import of the neighbour (whose other imports move one LDS slot
    on), one more local body of the seam, a seam round with room (a spare one opens when the box already has both rounds taken).  The
    boxes are picked with s2amd_get_strip_owners: a box in the middle strip that is on no seam, then a box of the right-hand
    neighbour that the seam already carries, then one it does not.  No structure build, same bits as the oracle."""
    vel, pos = common.DEFAULT_ITERS[solver_name]
    base = 100
    pre = _pyramid_with_free_bodies(base, 1, 6)
    params = wire.StepParams.make(solver_name, 1.0 / 60.0, vel, pos, True)
    nb = len(pre[0])
    ball = nb - 1
    template = int(np.flatnonzero(pre[1]["pointCount"] == 2)[len(pre[1]) // 3])
    spare = [len(pre[1]) - 6 + i for i in range(6)]
    with hip.Solver(0) as s:
        s.set_option("strip_patience", 0)
        s.set_option("strip_bodies", 240)  # (strips of several levels: they have bodies no seam carries; in two-level strips every body is on a seam)
        s.set_option("strip_retry", 0)
        state = common.copy3(pre)
        for step in range(2):
            state = gpu_vs_oracle_loose(s, params, state, "seam warm-up %d" % step)
        builds = s.stats()["structureBuilds"]
        owner, seam, count, nbrs = _strip_picture(s, state)
        assert count > 8 and owner[ball] >= 0
        k = count // 2
        inner = [b for b in np.flatnonzero((owner == k) & (seam < 0)).tolist() if b != ball]
        right_on_seam = np.flatnonzero((owner == k + 1) & (seam == k)).tolist()
        right_far = np.flatnonzero((owner == k + 1) & (seam < 0)).tolist()
        left_on_seam = np.flatnonzero((owner == k - 1) & (seam == k - 1)).tolist()
        assert inner and right_on_seam and right_far and left_on_seam, (count, len(inner), len(right_on_seam), len(right_far), len(left_on_seam))
        touches = [(inner[0], True), (right_on_seam[0], True), (right_far[0], True),
                   (left_on_seam[0], False)]  # (the last: the ball would be on both seams of its strip -- refused, the structure is built again)
        for n, (box, placed) in enumerate(touches):
            _touch(state[1], spare[n], template, ball, box)
            state = gpu_vs_oracle_loose(s, params, state, "%s ball touches box %d" % (solver_name, box))
            state = gpu_vs_oracle_loose(s, params, state, "%s ball touches box %d, next step" % (solver_name, box))
            st = s.stats()
            if placed:
                assert st["persistent"] == 1 and st["pairLanes"] == 2 and st["structureBuilds"] == builds, (n, {k_: st[k_] for k_ in ("persistent", "pairLanes", "structureBuilds", "placedContacts")})
                seen = st
            else:
                assert st["structureBuilds"] > builds
        assert seen["placedContacts"] >= 3 and seen["bodiesAdopted"] == 1 and seen["seamBodiesAdded"] == 2, seen


def test_two_boxes_of_neighbouring_strips_that_never_faced_each_other():
    """A contact between two boxes of neighbouring strips of which the seam carries NEITHER (both sit on their strips' far sides, or in
    the middle of a thick strip): both become bodies of the seam.  And the rule that keeps it correct: a box that its strip's OTHER seam
    already carries is refused (the two seams of a strip are swept in the same rounds by different pairs of workgroups) -- that contact
    rebuilds the structure as it always did."""
    base = 100
    pre = _pyramid_with_free_bodies(base, 0, 2)
    params = wire.StepParams.make("TGS_Soft", 1.0 / 60.0, 8, 4, True)
    template = int(np.flatnonzero(pre[1]["pointCount"] == 2)[len(pre[1]) // 3])
    for case in ("both", "other seam"):
        with hip.Solver(0) as s:
            s.set_option("strip_patience", 0)
            s.set_option("strip_bodies", 240 if case == "both" else 8)  # (thick strips have bodies no seam carries; two-level strips do not)
            s.set_option("strip_retry", 0)
            state = common.copy3(pre)
            for step in range(2):
                state = gpu_vs_oracle_loose(s, params, state, "far levels warm-up %d" % step)
            builds = s.stats()["structureBuilds"]
            owner, seam, count, nbrs = _strip_picture(s, state)
            k = count // 2
            if case == "both":
                a = np.flatnonzero((owner == k) & (seam < 0)).tolist()
                b = np.flatnonzero((owner == k + 1) & (seam < 0)).tolist()
            else:
                a = np.flatnonzero((owner == k) & (seam == k - 1)).tolist()
                b = np.flatnonzero((owner == k + 1) & (seam == k)).tolist()
            assert a and b, (case, count)
            _touch(state[1], len(pre[1]) - 2, template, a[0], b[0])
            for step in range(3):
                state = gpu_vs_oracle_loose(s, params, state, "far levels (%s) step %d" % (case, step))
            st = s.stats()
            if case == "both":
                assert st["structureBuilds"] == builds and st["persistent"] == 1 and st["seamBodiesAdded"] == 2, st
            else:
                assert st["structureBuilds"] > builds, st


def test_a_time_out_on_the_same_xcd_hand_off_path_first_costs_only_that_path():
    """persist_handoff.h: putGranuleNear hands bodies to a neighbour on the same XCD with workgroup-scope stores -- an assumption about
    the cache hierarchy, checked by a census every launch, that the memory model does not promise.  When a hand-off times out while
    that path is in use, the step is tried again on the SAME one-launch kernel with agent-scope stores everywhere (stats
    nearHandoffTimeouts); only a second time-out puts the solver on the multi-launch path.  With the fault injected (workgroup 1 never
    publishes) both happen in one synchronous step; under "async" the first s2amd_synchronize reports the one, the second the other.
    Every result bit-exact; option near_handoff 0 never uses the path."""
    pre = synthetic.pyramid(100)
    params = wire.StepParams.make("TGS_Soft", 1.0 / 60.0, 8, 4, True)
    with hip.Solver(0) as s:
        s.set_option("strip_patience", 0)
        s.set_option("persist_spin_limit", 4096)
        s.set_option("persist_debug", 8)
        state = gpu_vs_oracle(s, params, pre, "near path: the fault, synchronous")
        st = s.stats()
        assert st["nearHandoffTimeouts"] == 1 and st["persistFallbacks"] == 1 and st["persistent"] == 0, st
    with hip.Solver(0) as s:
        s.set_option("strip_patience", 0)
        s.set_option("persist_spin_limit", 4096)
        s.set_option("persist_debug", 8)
        s.set_option("async", 1)
        s.upload(*pre)
        s.step_resident(params)
        with pytest.raises(hip.S2AmdError):
            s.synchronize()
        st = s.stats()
        assert st["nearHandoffTimeouts"] == 1 and st["persistFallbacks"] == 0, st
        s.step_resident(params)  # the repeated step: the one-launch kernel again, agent-scope stores -- and the injected fault again
        with pytest.raises(hip.S2AmdError):
            s.synchronize()
        assert s.stats()["persistFallbacks"] == 1
        want = common.copy3(pre)
        s.step_resident(params)
        s.synchronize()
        order, _ = s.contact_order()
        oraclebind.solve(params, *want, contact_order=order)
        got = common.copy3(pre)
        s.download(*got)
        common.compare_exact(got, want, "near path: the step repeated twice")
    results = []
    for near in (1, 0):  # ... and without a fault the two ways of handing off give the same bits
        with hip.Solver(0) as s:
            s.set_option("strip_patience", 0)
            s.set_option("near_handoff", near)
            state = common.copy3(pre)
            for step in range(3):
                state = gpu_vs_oracle(s, params, state, "near_handoff %d step %d" % (near, step))
            assert s.stats()["persistent"] == 1 and s.stats()["nearHandoffTimeouts"] == 0
            results.append(state)
    common.compare_exact(results[0], results[1], "near_handoff 1 vs 0")
