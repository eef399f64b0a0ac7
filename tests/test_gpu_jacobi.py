"""s2Solve_Jacobi as ONE persistent launch (solver2d_amd/csrc/jacobi_kernel.hip; tables: solver_jacobi.cpp; BASELINE.json configs[2]).

The bodies are dealt to blocks (chunks of a breadth-first order of the constraint graph), one workgroup per block for the whole step;
a block holds every constraint of its bodies in registers (a constraint between two blocks is held by both), velocities and the
per-constraint deltas in LDS, and an iteration is the contact pass, the per-body sums in pool order and ONE exchange of velocities
between the blocks.  Results against the oracle, bit for bit -- s2Solve_Jacobi has no sweep order to choose, only the order of a
body's additions, which is the pool's --, and against the multi-launch path (option jacobi_persist 0), which must give the same bits.
"""
import os

import numpy as np
import pytest

from solver2d_amd import hip, synthetic, wire
from tests import common, golden_util
from tests.test_gpu_incremental import _artificial_contact, _with_spare_slots
from tests.test_gpu_parity import gpu_vs_oracle, gpu_vs_oracle_loose

pytestmark = pytest.mark.gpu

JACOBI = wire.StepParams.make("Jacobi", 1.0 / 60.0, 4, 2, True)
FILES = [p for p in golden_util.golden_files() if "_Jacobi_" in os.path.basename(p)]


@pytest.mark.parametrize("base", [30, 60, 120])
def test_pyramid_in_one_launch(base):
    with hip.Solver(0) as s:
        state = common.copy3(synthetic.pyramid(base))
        for step in range(4):
            state = gpu_vs_oracle(s, JACOBI, state, "pyramid %d / Jacobi step %d" % (base, step))
        st = s.stats()
    assert st["persistent"] == 1 and st["kernelLaunches"] <= 3, st


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(p)[:-4] for p in FILES])
def test_golden_jacobi_inputs_through_the_persistent_launch(path):
    """every Jacobi fixture (joints, mouse joints, kinematic bodies, one-point manifolds, hubs) with the size threshold off: a world whose
    joints do not each lie inside one block keeps the multi-launch path -- either way the oracle's bits"""
    params, pre, _post = golden_util.load(path)
    with hip.Solver(0) as s:
        s.set_option("jacobi_min_constraints", 1)
        gpu_vs_oracle(s, params, pre, os.path.basename(path))


def _hub_world(base, spokes, motor):
    """a pyramid with one heavy body that touches `spokes` of its boxes (made-up manifolds: a solver input, not a scene) and hangs on a
    motorised revolute joint from the ground -- the Tumbler's drum in small"""
    rng = np.random.default_rng(base + spokes)
    b, c, j = _with_spare_slots(synthetic.pyramid(base), spokes)
    hub = np.zeros(1, dtype=wire.body_dtype)
    hub[0] = b[int(np.flatnonzero(b["type"] == wire.BODY_DYNAMIC)[0])]
    hub["position"] = (0.0, float(base) + 5.0)
    hub["mass"], hub["invMass"], hub["I"], hub["invI"] = 50.0, 1.0 / 50.0, 400.0, 1.0 / 400.0
    hub["angularVelocity"] = 0.3
    b = np.concatenate([b, hub])
    h = len(b) - 1
    dyn = np.flatnonzero(b["type"] == wire.BODY_DYNAMIC)[:-1]
    boxes = rng.choice(dyn, size=spokes, replace=False)
    n0 = len(c) - spokes
    for i, box in enumerate(boxes):
        new = _artificial_contact(rng, b, c[5], set())
        new["bodyA"], new["bodyB"] = (h, int(box)) if i % 2 else (int(box), h)
        c[n0 + i] = new
    if motor:
        jt = np.zeros(1, dtype=wire.joint_dtype)
        ground = int(np.flatnonzero(b["type"] == wire.BODY_STATIC)[0])
        jt["type"], jt["bodyA"], jt["bodyB"] = wire.JOINT_REVOLUTE, ground, h
        jt["enableMotor"], jt["motorSpeed"], jt["maxMotorTorque"] = 1, 0.5, 1e5
        jt["localOriginAnchorA"] = b["position"][h] - b["position"][ground]
        j = np.concatenate([j, jt])
    return b, c, j


@pytest.mark.parametrize("spokes,motor", [(40, False), (150, True), (300, True)])
def test_a_hub_body_is_one_body_of_one_block(spokes, motor):
    """the drum in small: every one of the hub's constraints is held by its block (and by the box's), its deltas are summed by one wave in
    pool order, its joint is solved by its block's lane 0 -- one launch, the oracle's bits"""
    world = _hub_world(40, spokes, motor)
    with hip.Solver(0) as s:
        state = common.copy3(world)
        for step in range(3):
            state = gpu_vs_oracle_loose(s, JACOBI, state, "hub %d motor %d step %d" % (spokes, motor, step))
        st = s.stats()
    assert st["persistent"] == 1 and st["kernelLaunches"] <= 4, st


def test_the_multi_launch_path_gives_the_same_bits():
    world = _hub_world(50, 200, True)
    out = []
    launches = []
    for persist in (1, 0):
        with hip.Solver(0) as s:
            s.set_option("jacobi_persist", persist)
            b, c, j = common.copy3(world)
            s.upload(b, c, j)
            for _ in range(5):
                s.step_resident(JACOBI)
            s.download(b, c, j)
            launches.append(s.stats()["kernelLaunches"])
            out.append((b, c, j))
    common.compare_exact(out[0], out[1], "persistent against multi-launch")
    assert launches[0] <= 4 < launches[1], launches


def test_a_created_contact_sends_the_steps_back_to_the_multi_launch_path_until_the_next_build():
    """the block tables know the constraints of the build: a contact placed into a free position afterwards (solver_incremental.cpp) is
    not in them -- the steps take the multi-launch path (which sees it) and stay the oracle's"""
    rng = np.random.default_rng(5)
    pre = _with_spare_slots(synthetic.pyramid(40), 8)
    n0 = len(pre[1]) - 8
    with hip.Solver(0) as s:
        state = common.copy3(pre)
        state = gpu_vs_oracle(s, JACOBI, state, "before")
        assert s.stats()["persistent"] == 1
        pairs = {(min(a, b), max(a, b)) for a, b in zip(state[1]["bodyA"][:n0].tolist(), state[1]["bodyB"][:n0].tolist())}
        state[1][n0] = _artificial_contact(rng, state[0], pre[1][5], pairs)
        state = gpu_vs_oracle(s, JACOBI, state, "with the created contact")
        st = s.stats()
        assert st["placedContacts"] >= 1 and st["persistent"] == 0, st
