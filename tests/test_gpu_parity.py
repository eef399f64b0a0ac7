"""Parity link L2 (the gate): HIP kernels == oracle, BIT FOR BIT, when the oracle sweeps in the
device's own constraint order (s2amd_get_contact_order / s2amd_get_joint_order).

All calls go through the C-ABI (solver2d_amd/libs2amd.so).  The oracle is only the checker.
Tolerance: none -- fp32 outputs are compared as raw 32-bit words (the library is built with
-ffp-contract=off and IEEE divide/sqrt, see solver2d_amd/csrc/s2_device.h).
"""
import os

import numpy as np
import pytest

from solver2d_amd import hip, synthetic, wire
from tests import common, golden_util, oraclebind

pytestmark = pytest.mark.gpu

FILES = golden_util.golden_files()


@pytest.fixture(scope="module")
def solver():
    s = hip.Solver(0)
    yield s
    s.close()


def check_order_is_valid(order, offsets, contacts, bodies, colourless=False):
    """Every active contact appears exactly once; inside one colour batch no dynamic body repeats (s2Solve_Jacobi's contact
    pass writes no body -- solve_jacobi.c:21-132 accumulates per-constraint deltas -- so its sweep is one batch in pool
    order and has no colours to check)."""
    active = np.flatnonzero(contacts["pointCount"] > 0)
    assert sorted(order.tolist()) == active.tolist()
    assert offsets[0] == 0 and offsets[-1] == len(order)
    if colourless:
        return
    movable = (bodies["invMass"] != 0) | (bodies["invI"] != 0)
    for c in range(len(offsets) - 1):
        ids = order[offsets[c]:offsets[c + 1]]
        touched = np.concatenate([contacts["bodyA"][ids], contacts["bodyB"][ids]])
        touched = touched[movable[touched]]
        assert len(np.unique(touched)) == len(touched), "colour %d reuses a dynamic body" % c


def gpu_vs_oracle(solver, params, pre, what):
    got = common.copy3(pre)
    solver.solve(params, *got)
    order, offsets = solver.contact_order()
    jorder, joffsets = solver.joint_order()
    check_order_is_valid(order, offsets, pre[1], pre[0], colourless=params.solverType == wire.SOLVER_ID["Jacobi"])
    want = common.copy3(pre)
    oraclebind.solve(params, *want, contact_order=order, joint_order=jorder)
    common.compare_exact(got, want, what)
    return got


def gpu_vs_oracle_loose(solver, params, pre, what):
    """gpu_vs_oracle for arbitrary inputs: NaN == NaN bitwise is still required, but the colour check
    uses the library's own notion of a writable body (rotated statics count for position solvers)."""
    got = common.copy3(pre)
    solver.solve(params, *got)
    order, offsets = solver.contact_order()
    jorder, _ = solver.joint_order()
    active = np.flatnonzero(pre[1]["pointCount"] > 0)
    assert sorted(order.tolist()) == active.tolist()
    assert offsets[0] == 0 and offsets[-1] == len(order)
    if params.solverType != wire.SOLVER_ID["Jacobi"]:
        # no colour holds two constraints on one body its sweeps write (s2amd_get_writable_bodies: the library's own rule, which for
        # the position passes counts a rotated static body as written -- solve_common.c:383-392)
        writable, _cls = solver.writable_bodies(len(pre[0]))
        a, b = pre[1]["bodyA"][order], pre[1]["bodyB"][order]
        colour = np.repeat(np.arange(len(offsets) - 1), np.diff(offsets))
        touched = np.concatenate([a, b]).astype(np.int64)
        keys = np.concatenate([colour, colour]).astype(np.int64)[writable[touched] != 0] * (len(pre[0]) + 1) + touched[writable[touched] != 0]
        assert len(np.unique(keys)) == len(keys), "%s: a colour holds two constraints on one writable body" % what
    want = common.copy3(pre)
    oraclebind.solve(params, *want, contact_order=order, joint_order=jorder)
    common.compare_exact(got, want, what)
    return got


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(p)[:-4] for p in FILES])
def test_golden_inputs_bit_exact(solver, path):
    params, pre, _post = golden_util.load(path)
    gpu_vs_oracle(solver, params, pre, os.path.basename(path))


@pytest.fixture(scope="module")
def solver_nogroups():
    s = hip.Solver(0)
    s.set_option("groups", 0)
    yield s
    s.close()


@pytest.mark.parametrize("path", FILES[::3], ids=[os.path.basename(p)[:-4] for p in FILES[::3]])
def test_golden_inputs_bit_exact_without_groups(solver_nogroups, path):
    """Same inputs through the global colour-batch path only (LDS groups switched off)."""
    params, pre, _post = golden_util.load(path)
    gpu_vs_oracle(solver_nogroups, params, pre, os.path.basename(path))
    assert solver_nogroups.stats()["groupCount"] == 0


@pytest.mark.parametrize("message", [1, 0])
@pytest.mark.parametrize("solver_name", ["TGS_Soft", "SoftStep", "PGS", "PGS_Soft", "TGS_Sticky", "XPBD", "PGS_NGS"])
def test_big_island_message_passing_on_off(solver_name, message):
    """Global path (groups off) with and without the message-passing accessor: both bit-exact."""
    vel, pos = common.DEFAULT_ITERS[solver_name]
    pre = synthetic.pyramid(30)
    with hip.Solver(0) as s:
        s.set_option("groups", 0)
        s.set_option("message", message)
        state = common.copy3(pre)
        for step in range(3):
            params = wire.StepParams.make(solver_name, 1.0 / 60.0, vel, pos, True)
            state = gpu_vs_oracle(s, params, state, "pyramid30/%s msg=%d step %d" % (solver_name, message, step))
        expect = message == 1 and solver_name in ("TGS_Soft", "SoftStep", "PGS", "PGS_Soft", "TGS_Sticky")
        assert s.stats()["messagePassing"] == (1 if expect else 0)


@pytest.mark.parametrize("body_warm", [1, 0])
@pytest.mark.parametrize("solver_name", wire.SOLVER_NAMES)
def test_big_island_body_centric_warm_start_on_off(solver_name, body_warm):
    """The one-launch body-centric warm start must give the same bits as the coloured warm-start sweep."""
    vel, pos = common.DEFAULT_ITERS[solver_name]
    pre = synthetic.pyramid(24)
    with hip.Solver(0) as s:
        s.set_option("groups", 0)
        s.set_option("body_warm", body_warm)
        state = common.copy3(pre)
        launches = 0
        for step in range(3):
            params = wire.StepParams.make(solver_name, 1.0 / 60.0, vel, pos, True)
            state = gpu_vs_oracle(s, params, state, "pyramid24/%s bw=%d step %d" % (solver_name, body_warm, step))
            launches = s.stats()["kernelLaunches"]
    assert launches > 0


def test_small_worlds_run_as_lds_groups(solver):
    params, pre, _post = golden_util.load([f for f in FILES if "pyramid10_TGS_Soft_step045" in f][0])
    gpu_vs_oracle(solver, params, pre, "pyramid10 grouped")
    st = solver.stats()
    assert st["groupCount"] == 1 and st["kernelLaunches"] <= 8


@pytest.mark.parametrize("solver_name", wire.SOLVER_NAMES)
def test_big_island_with_high_degree_body_uses_global_tail(solver, solver_name):
    """2,400-body island (too big for an LDS group) whose platform touches 60 boxes: global colour
    batches + one sequential LDS tail launch per sweep."""
    vel, pos = common.DEFAULT_ITERS[solver_name]
    pre = synthetic.platform(60, layers=40)
    params = wire.StepParams.make(solver_name, 1.0 / 60.0, vel, pos, True)
    gpu_vs_oracle(solver, params, pre, "platform60x40/%s" % solver_name)
    st = solver.stats()
    # (s2Solve_Jacobi's contact pass has no colours: one batch, the platform's 60 contacts are a long incidence list)
    assert st["groupCount"] == 0 and st["contactColors"] >= (1 if solver_name == "Jacobi" else 60)


@pytest.mark.parametrize("solver_name", wire.SOLVER_NAMES)
def test_synthetic_pyramid40_all_solvers(solver, solver_name):
    vel, pos = common.DEFAULT_ITERS[solver_name]
    pre = synthetic.pyramid(40)
    state = common.copy3(pre)
    for step in range(3):  # impulses carried across steps: exercises warm starting
        params = wire.StepParams.make(solver_name, 1.0 / 60.0, vel, pos, True)
        state = gpu_vs_oracle(solver, params, state, "pyramid40/%s step %d" % (solver_name, step))


def test_baseline_config2_full_size(solver):
    """LargePyramid base-200, TGS_Soft 8 sub-steps / relax on: the BASELINE configuration, full size."""
    pre = synthetic.pyramid(200)
    assert (pre[1]["pointCount"] > 0).sum() == 59900
    state = common.copy3(pre)
    for step in range(2):
        params = wire.StepParams.make("TGS_Soft", 1.0 / 60.0, 8, 4, True)
        state = gpu_vs_oracle(solver, params, state, "pyramid200 step %d" % step)
    st = solver.stats()
    assert st["constraintCount"] == 59900 and st["solveSweeps"] == 16
    # physical sanity: the pile rests, nothing explodes
    assert np.isfinite(state[0]["position"]).all()
    assert np.abs(state[0]["linearVelocity"]).max() < 1.0


def test_multi_island_config5_shape(solver):
    pre = synthetic.pyramid(12, count=6)
    params = wire.StepParams.make("TGS_Soft", 1.0 / 60.0, 8, 4, True)
    gpu_vs_oracle(solver, params, pre, "6 x pyramid12")


def test_joint_grid_pgs_ngs(solver):
    pre = synthetic.joint_grid(20)
    state = common.copy3(pre)
    for step in range(3):
        params = wire.StepParams.make("PGS_NGS", 1.0 / 60.0, 4, 2, True)
        state = gpu_vs_oracle(solver, params, state, "joint_grid20 step %d" % step)
    assert solver.stats()["jointCount"] == 2 * 20 * 19


@pytest.mark.parametrize("solver_name", ["Jacobi", "TGS_Soft", "SoftStep", "PGS_NGS_Block", "PGS"])
def test_body_with_hundreds_of_constraints(solver, solver_name):
    """A platform carrying 300 boxes: its incidence list spans several 64-entry chunks of the wave that walks it in the
    body-centric kernels (jacobiApplyKernel: more than one 256-entry round; warmStartBodiesKernel in all three kinds).
    Second step: non-zero impulses to warm start from."""
    vel, pos = common.DEFAULT_ITERS[solver_name]
    pre = synthetic.platform(300, layers=8)
    params = wire.StepParams.make(solver_name, 1.0 / 60.0, vel, pos, True)
    state = gpu_vs_oracle(solver, params, common.copy3(pre), "platform300/%s step 0" % solver_name)
    gpu_vs_oracle(solver, params, state, "platform300/%s step 1" % solver_name)
    assert solver.stats()["groupCount"] == 0


@pytest.mark.parametrize("solver_name", wire.SOLVER_NAMES)
def test_high_degree_body_uses_sequential_tail(solver, solver_name):
    """One dynamic platform touching 60 boxes: 60+ colours; the high colours run as one sequential
    tail launch.  Must still equal the oracle in the reported order, bit for bit."""
    vel, pos = common.DEFAULT_ITERS[solver_name]
    pre = synthetic.platform(60, layers=2)
    state = common.copy3(pre)
    for step in range(2):
        params = wire.StepParams.make(solver_name, 1.0 / 60.0, vel, pos, True)
        state = gpu_vs_oracle(solver, params, state, "platform60/%s step %d" % (solver_name, step))
    st = solver.stats()
    assert st["contactColors"] >= (1 if solver_name == "Jacobi" else 60)  # (no colours under s2Solve_Jacobi)
    if solver_name != "Jacobi":
        assert st["groupCount"] == 1 and st["kernelLaunches"] <= 8


def test_resident_api_equals_solve(solver):
    pre = synthetic.pyramid(16)
    params = wire.StepParams.make("TGS_Soft", 1.0 / 60.0, 8, 4, True)
    a = common.copy3(pre)
    for _ in range(4):
        solver.solve(params, *a)
    b = common.copy3(pre)
    solver.upload(*b)
    for _ in range(4):
        solver.step_resident(params)
    solver.download(*b)
    common.compare_exact(b, a, "resident vs solve")


def test_graph_replay_equals_eager():
    pre = synthetic.pyramid(16)
    params = wire.StepParams.make("TGS_Soft", 1.0 / 60.0, 8, 4, True)
    outs = []
    for graph in (True, False):
        with hip.Solver(0, graph=graph) as s:
            s.set_option("graph_min_launches", 0)  # (a step of a few launches is never captured by default)
            st = common.copy3(pre)
            s.upload(*st)
            # a launch sequence is enqueued directly the first time, captured when it comes back, replayed from then on;
            # the structure is looked at once more when the graph has stayed the same for a step (strip_patience)
            for _ in range(5):
                s.step_resident(params)
            assert s.stats()["graphReplayed"] == (1 if graph else 0)
            assert s.stats()["groupCount"] == 1
            s.download(*st)
            outs.append(st)
    common.compare_exact(outs[0], outs[1], "graph vs eager")


def test_save_restore_bodies(solver):
    pre = synthetic.pyramid(8)
    params = wire.StepParams.make("TGS_Soft", 1.0 / 60.0, 8, 4, True)
    st = common.copy3(pre)
    solver.upload(*st)
    solver.save_bodies()
    solver.step_resident(params)
    solver.restore_bodies()
    out = common.copy3(pre)
    solver.download(*out)
    for f in common.BODY_OUT:
        assert np.array_equal(out[0][f], pre[0][f])
    assert np.abs(out[1]["points"]["normalImpulse"]).max() > 0  # impulses were kept


def test_edge_cases(solver):
    params = wire.StepParams.make("TGS_Soft", 1.0 / 60.0, 8, 4, True)
    # empty world
    e = (np.zeros(0, wire.body_dtype), np.zeros(0, wire.contact_dtype), np.zeros(0, wire.joint_dtype))
    solver.solve(params, *e)
    # bodies only, with free slots
    b, c, j = synthetic.pyramid(3)
    b2 = np.concatenate([b, np.zeros(3, wire.body_dtype)])
    b2["type"][-3:] = wire.BODY_FREE
    for name in wire.SOLVER_NAMES:
        vel, pos = common.DEFAULT_ITERS[name]
        p = wire.StepParams.make(name, 1.0 / 60.0, vel, pos, True)
        gpu_vs_oracle(solver, p, (b2.copy(), np.zeros(0, wire.contact_dtype), j.copy()), "no contacts/" + name)
        # inactive contact slots interleaved + one-point manifolds
        c2 = np.zeros(2 * len(c), wire.contact_dtype)
        c2[1::2] = c
        c2["bodyA"][0::2] = -1
        c2["bodyB"][0::2] = -1
        c2["constraintIndex"] = -1
        c2["pointCount"][1::4] = 1
        gpu_vs_oracle(solver, p, (b2.copy(), c2, j.copy()), "ragged/" + name)
        # no warm start, zero extra iterations
        p0 = wire.StepParams.make(name, 1.0 / 60.0, 3, 0, False)
        gpu_vs_oracle(solver, p0, (b2.copy(), c2.copy(), j.copy()), "cold/" + name)
    # XPBD early-outs (solve_xpbd.c:344-353)
    px = wire.StepParams.make("XPBD", 1.0 / 60.0, 0, 0, True)
    gpu_vs_oracle(solver, px, (b.copy(), c.copy(), j.copy()), "xpbd zero substeps")
    px = wire.StepParams.make("XPBD", 0.0, 4, 2, True)
    gpu_vs_oracle(solver, px, (b.copy(), c.copy(), j.copy()), "xpbd dt=0")


def test_invalid_arguments(solver):
    b, c, j = synthetic.pyramid(3)
    with pytest.raises(hip.S2AmdError):
        solver.solve(wire.StepParams.make(17), b, c, j)
    c2 = c.copy()
    c2["bodyB"][0] = 10 ** 6
    with pytest.raises(hip.S2AmdError):
        solver.solve(wire.StepParams.make("PGS"), b, c2, j)
