"""The persistent strip step as an op interpreter (generic_kernel.hip: genericStepKernel): every Gauss-Seidel solver family and
every joint sweep of a big island in ONE launch per step -- strips of the island owned by one workgroup each for the whole
`s2Solve_*`, seams swept by their left strip between a forward and a return hand-off.

Same gate as everywhere: the C-ABI result must equal the oracle BIT FOR BIT when the oracle sweeps in the order the library
reports (per op: strip by strip colour-major, then seam by seam)."""
import os

import numpy as np
import pytest

from solver2d_amd import hip, synthetic, wire
from tests import common, fuzz_worlds, golden_util
from tests.test_gpu_parity import gpu_vs_oracle, gpu_vs_oracle_loose

pytestmark = pytest.mark.gpu

FILES = golden_util.golden_files()
GS_SOLVERS = [n for n in wire.SOLVER_NAMES if n != "Jacobi"]
SOFT = ("TGS_Soft", "SoftStep", "PGS_Soft")


@pytest.mark.parametrize("solver_name", GS_SOLVERS)
def test_every_solver_takes_one_launch_on_the_big_pyramid(solver_name):
    """Base-100 pyramid (5,050 bodies, one island), default options: prologue + ONE step kernel + epilogue, whatever the solver.
    The soft contact drivers keep their register-resident kernels; everything else runs on the op interpreter."""
    vel, pos = common.DEFAULT_ITERS[solver_name]
    pre = synthetic.pyramid(100)
    with hip.Solver(0) as s:
        s.set_option("strip_patience", 0)
        state = common.copy3(pre)
        for step in range(3):
            params = wire.StepParams.make(solver_name, 1.0 / 60.0, vel, pos, True)
            state = gpu_vs_oracle(s, params, state, "pyramid100/%s step %d" % (solver_name, step))
        st = s.stats()
        assert st["stripCount"] >= 4 and st["persistent"] == 1 and st["kernelLaunches"] <= 4, st
        assert (st["pairLanes"] == 3) == (solver_name not in SOFT), st


@pytest.mark.parametrize("solver_name", GS_SOLVERS)
def test_op_interpreter_equals_the_colour_batches_and_the_strip_launches(solver_name):
    """Three routes for the same world and solver -- the op interpreter, the multi-launch strip path (test option), colour batches
    -- are each bit-equal to the oracle in their own order; the first two share one order, so they are bit-equal to each other."""
    vel, pos = common.DEFAULT_ITERS[solver_name]
    pre = synthetic.pyramid(72)
    params = wire.StepParams.make(solver_name, 1.0 / 60.0, vel, pos, True)
    results = {}
    min_bodies = 0  # (2,628 bodies: under the default threshold for strips)
    for route in ("interpreter", "launches", "batches"):
        with hip.Solver(0) as s:
            s.set_option("strip_patience", 0)
            s.set_option("strip_min_bodies", min_bodies)
            if solver_name in SOFT:
                s.set_option("persist", 1 if route == "interpreter" else 0)  # (with "persist" off nothing runs in one launch)
                s.set_option("wide", 0)
            if route == "launches":
                s.set_option("generic", 0)
                s.set_option("strips_any_solver", 1)
            if route == "batches":
                s.set_option("strips", 0)
            state = common.copy3(pre)
            for step in range(2):
                state = gpu_vs_oracle(s, params, state, "pyramid72/%s %s step %d" % (solver_name, route, step))
            st = s.stats()
            assert (st["stripCount"] > 0) == (route != "batches"), (route, st)
            assert st["persistent"] == (1 if route == "interpreter" else 0), (route, st)
            results[route] = state
    common.compare_exact(results["interpreter"], results["launches"], "%s: interpreter vs strip launches" % solver_name)


@pytest.mark.parametrize("solver_name", ["TGS_Soft", "PGS_NGS", "PGS_NGS_Block", "XPBD", "TGS_Sticky", "SoftStep", "TGS_NGS"])
def test_joint_grid_in_one_launch(solver_name):
    """70x70 joint grid (4,900 bodies, 9,660 revolute joints, one island, no contacts): joint sweeps in the strips."""
    vel, pos = common.DEFAULT_ITERS[solver_name]
    pre = synthetic.joint_grid(70)
    with hip.Solver(0) as s:
        s.set_option("strip_patience", 0)
        state = common.copy3(pre)
        for step in range(3):
            params = wire.StepParams.make(solver_name, 1.0 / 60.0, vel, pos, True)
            state = gpu_vs_oracle(s, params, state, "jointgrid70/%s step %d" % (solver_name, step))
        st = s.stats()
        assert st["stripCount"] >= 4 and st["persistent"] == 1 and st["pairLanes"] == 3 and st["kernelLaunches"] <= 5, st


@pytest.fixture(scope="module")
def tiny_strips():
    """Forces strips on the small golden worlds (joints, every shape of contact graph): no island fits a group, strips of ~12
    bodies, the op interpreter for every solver."""
    s = hip.Solver(0)
    s.set_option("strip_patience", 0)
    s.set_option("max_group_bodies", 48)
    s.set_option("strip_min_bodies", 0)
    s.set_option("strip_bodies", 12)
    s.set_option("persist", 1)
    yield s
    s.close()


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(p)[:-4] for p in FILES])
def test_golden_inputs_bit_exact_through_the_op_interpreter(tiny_strips, path):
    params, pre, _post = golden_util.load(path)
    gpu_vs_oracle(tiny_strips, params, pre, os.path.basename(path))


def test_golden_worlds_really_ran_on_the_op_interpreter(tiny_strips):
    seen = 0
    for path in FILES:
        if "Jacobi" in path:
            continue
        params, pre, _post = golden_util.load(path)
        gpu_vs_oracle(tiny_strips, params, pre, os.path.basename(path))
        st = tiny_strips.stats()
        seen += st["stripCount"] >= 2 and st["persistent"] == 1 and st["pairLanes"] == 3
    assert seen >= 30, seen


def pile_with_joints(seed, base, joints):
    """The pyramid's contact graph (long BFS diameter: strips apply) with randomised numbers, one-point and inactive contacts,
    kinematic and static bodies in the pile (tests/test_fuzz.py: perturbed_pyramid) -- plus `joints` revolute and mouse joints with
    limits, motors and springs (tests/fuzz_worlds.py's mix) between boxes that touch, i.e. inside strips and across seams."""
    from tests.test_fuzz import perturbed_pyramid
    rng = np.random.default_rng(7000 + seed)
    bodies, contacts, _ = perturbed_pyramid(seed, base)
    # (a static body whose rot is not a fixed point of the normalisation is written by the position sweeps and can be owned by no
    # strip: such a world keeps its colour batches under the position solvers -- tests/test_fuzz.py covers that; here strips are wanted)
    static = bodies["type"] == wire.BODY_STATIC
    bodies["rot"][static, 0], bodies["rot"][static, 1] = 0.0, 1.0
    # contacts between two immovable bodies (a kinematic box on the static ground) cannot exist in the reference -- kinematic proxies only
    # query the dynamic tree, src/broad_phase.c:301-305 -- and XPBD divides by their zero effective mass (solve_xpbd.c:186)
    immovable = (bodies["invMass"] == 0.0) & (bodies["invI"] == 0.0)
    both = (contacts["bodyA"] >= 0) & immovable[np.maximum(contacts["bodyA"], 0)] & immovable[np.maximum(contacts["bodyB"], 0)]
    contacts["pointCount"][both] = 0
    donor = fuzz_worlds.random_world(seed + 900, n_bodies=30, n_contacts=10, n_joints=joints)[2]
    live = np.flatnonzero(contacts["bodyA"] >= 0)
    jn = donor.copy()
    for i in range(len(jn)):
        if jn[i]["type"] == wire.JOINT_FREE:
            continue
        k = int(rng.choice(live))
        a, b = int(contacts["bodyA"][k]), int(contacts["bodyB"][k])
        if bodies["invMass"][b] == 0.0:
            a, b = b, a
        if bodies["invMass"][b] == 0.0:
            jn[i]["type"] = wire.JOINT_FREE
            jn[i]["bodyA"] = jn[i]["bodyB"] = -1
            continue
        jn[i]["bodyA"], jn[i]["bodyB"] = a, b
    return bodies, contacts, jn


@pytest.mark.parametrize("joints", [0, 30])
@pytest.mark.parametrize("seed", list(range(8)))
def test_perturbed_piles_with_joints_through_the_op_interpreter(seed, joints):
    """Every Gauss-Seidel solver, warm and cold, on piles with kinematic / static bodies, one-point and inactive contacts and joints
    inside strips and across seams: most runs must take the interpreter (the soft solvers keep their own kernels on the joint-free
    piles; a partition with a level set wider than a strip may hold falls back), all must be bit-exact."""
    world = pile_with_joints(seed, 36, joints)
    ran = total = 0
    with hip.Solver(0) as gpu:
        gpu.set_option("strip_patience", 0)
        gpu.set_option("max_group_bodies", 256)  # the 666-body pile fits no group; a strip (>= 2 levels of <= 36 bodies) does
        gpu.set_option("strip_min_bodies", 0)
        gpu.set_option("strip_bodies", 8 if seed % 2 else 60)
        for solver_name in GS_SOLVERS:
            vel, pos = common.DEFAULT_ITERS[solver_name]
            state = world
            for warm in (True, False):
                p = wire.StepParams.make(solver_name, 1.0 / 60.0, vel, pos, warm)
                state = gpu_vs_oracle_loose(gpu, p, state, "pile seed %d %s warm=%d joints=%d" % (seed, solver_name, warm, joints))
                st = gpu.stats()
                total += 1
                ran += st["persistent"] == 1 and st["pairLanes"] == 3
    print("seed", seed, "joints", joints, "interpreter runs", ran, "of", total)
    assert ran >= (total if joints else total - 6) - 4, (ran, total)


@pytest.mark.parametrize("solver_name", ["TGS_Soft", "PGS_NGS_Block", "XPBD"])
def test_a_hub_body_keeps_its_island_off_the_strips(solver_name):
    """A body with many constraints among the strip candidates -- the Tumbler's drum has 238 -- would be as many colour rounds of one
    strip per sweep: when that costs more than the colour batches and their sequential tail (solver_internal.h: S2_COST_*; one hub:
    more than 23 constraints) the graph gets no strips (solver_structure.cpp: cutStrips), it runs on colour batches and the tail
    and is bit-exact there.  (Tumbler 10k under TGS_Soft: 3.2 ms that way, 5.8 ms through the interpreter.)  The same pile without
    the hub takes the strips."""
    bodies, contacts, joints = pile_with_joints(3, 36, 0)
    vel, pos = common.DEFAULT_ITERS[solver_name]
    params = wire.StepParams.make(solver_name, 1.0 / 60.0, vel, pos, True)
    hubbed = contacts.copy()
    movable = np.flatnonzero(bodies["invMass"] > 0.0)
    hub = int(movable[len(movable) // 2])
    live = np.flatnonzero((hubbed["bodyA"] >= 0) & (hubbed["bodyA"] != hub) & (hubbed["bodyB"] != hub) & (hubbed["pointCount"] > 0))
    picked = live[:: max(1, len(live) // 60)][:60]
    hubbed["bodyB"][picked] = hub  # 60 boxes all over the pile lean on one
    for cs, want_strips in ((contacts, True), (hubbed, False)):
        with hip.Solver(0) as gpu:
            gpu.set_option("strip_patience", 0)
            gpu.set_option("max_group_bodies", 256)
            gpu.set_option("strip_min_bodies", 0)
            gpu.set_option("strip_bodies", 60)
            state = (bodies, cs, joints)
            for step in range(2):
                state = gpu_vs_oracle_loose(gpu, params, state, "hub=%d %s step %d" % (not want_strips, solver_name, step))
            st = gpu.stats()
            assert (st["stripCount"] > 0) == want_strips, st
            if not want_strips:
                assert st["contactColors"] >= 60, st


@pytest.mark.parametrize("solver_name", ["TGS_Soft", "PGS_NGS_Block"])
def test_a_created_contact_of_a_hub_takes_a_place_in_the_sequential_tail(solver_name):
    """A hub (60 boxes lean on one) uses every parallel colour of the global part and has the rest of its constraints in the
    sequential tail.  A contact created on it finds no free colour: it takes a free position at the END of the tail
    (IncrementalGlobal::tailFree: any position of a sequential sweep is a valid one), the box it touches joins the tail's body
    list if the tail did not stage it yet -- no structure build.  Destroyed, it gives the position back; created again (another box)
    it is placed again.  Every step against the oracle."""
    bodies, contacts, joints = pile_with_joints(3, 36, 0)
    vel, pos = common.DEFAULT_ITERS[solver_name]
    params = wire.StepParams.make(solver_name, 1.0 / 60.0, vel, pos, True)
    movable = np.flatnonzero(bodies["invMass"] > 0.0)
    hub = int(movable[len(movable) // 2])
    live = np.flatnonzero((contacts["bodyA"] >= 0) & (contacts["bodyA"] != hub) & (contacts["bodyB"] != hub) & (contacts["pointCount"] > 0))
    picked = live[:: max(1, len(live) // 60)][:60]
    contacts = contacts.copy()
    contacts["bodyB"][picked] = hub
    free = np.zeros(3, dtype=wire.contact_dtype)
    free["bodyA"], free["bodyB"], free["constraintIndex"] = -1, -1, -1
    contacts = np.concatenate([contacts, free])
    spare = [len(contacts) - 3 + i for i in range(3)]
    touching = set(contacts["bodyA"][contacts["bodyB"] == hub].tolist()) | set(contacts["bodyB"][contacts["bodyA"] == hub].tolist())
    others = [int(b) for b in movable if int(b) != hub and int(b) not in touching]
    template = int(picked[0])
    with hip.Solver(0) as gpu:
        gpu.set_option("strip_patience", 0)
        gpu.set_option("max_group_bodies", 256)
        gpu.set_option("strip_min_bodies", 0)
        gpu.set_option("strip_bodies", 60)
        state = (bodies, contacts, joints)
        for step in range(2):
            state = gpu_vs_oracle_loose(gpu, params, state, "hub tail warm-up %d" % step)
        st = gpu.stats()
        assert st["stripCount"] == 0 and st["contactColors"] >= 60, st
        builds, placed = st["structureBuilds"], st["placedContacts"]
        script = {0: ("create", spare[0], others[3]), 1: ("create", spare[1], others[40]), 3: ("destroy", spare[0], None), 4: ("create", spare[2], others[77]),
                  5: ("create", spare[0], others[5])}
        for step in range(7):
            what = script.get(step)
            if what and what[0] == "create":
                state[1][what[1]] = state[1][template]
                state[1][what[1]]["bodyA"], state[1][what[1]]["bodyB"] = what[2], hub
                for p in range(2):
                    state[1][what[1]]["points"][p]["normalImpulse"] = 0.0
                    state[1][what[1]]["points"][p]["tangentImpulse"] = 0.0
            elif what:
                state[1][what[1]] = free[0]
            state = gpu_vs_oracle_loose(gpu, params, state, "hub tail %s step %d" % (solver_name, step))
            st = gpu.stats()
            assert st["structureBuilds"] == builds, (step, st["structureBuilds"], builds)
        assert gpu.stats()["placedContacts"] >= placed + 4, gpu.stats()


@pytest.mark.parametrize("degree,want_strips", [(18, True), (30, False)])
def test_the_hub_rule_is_a_cost_comparison(degree, want_strips):
    """Both sides of the threshold under the reference's default solver (the op interpreter takes strips of up to 32 colour rounds):
    18 boxes leaning on one -- 18 x 1.4 us per sweep on the strips against 24 + 18 x 0.38 us on the batches: strips --, 30 boxes --
    42 us against 35.4: no strips.  Bit-exact against the oracle either way."""
    bodies, contacts, joints = pile_with_joints(3, 36, 0)
    params = wire.StepParams.make("PGS_NGS_Block", 1.0 / 60.0, 4, 2, True)
    hubbed = contacts.copy()
    movable = np.flatnonzero(bodies["invMass"] > 0.0)
    hub = int(movable[len(movable) // 2])
    mine = int(((hubbed["bodyA"] == hub) | (hubbed["bodyB"] == hub)).sum())
    live = np.flatnonzero((hubbed["bodyA"] >= 0) & (hubbed["bodyA"] != hub) & (hubbed["bodyB"] != hub) & (hubbed["pointCount"] > 0))
    extra = degree - mine
    picked = live[:: max(1, len(live) // extra)][:extra]
    hubbed["bodyB"][picked] = hub
    with hip.Solver(0) as gpu:
        gpu.set_option("strip_patience", 0)
        gpu.set_option("max_group_bodies", 256)
        gpu.set_option("strip_min_bodies", 0)
        gpu.set_option("strip_bodies", 60)
        state = (bodies, hubbed, joints)
        for step in range(2):
            state = gpu_vs_oracle_loose(gpu, params, state, "hub of degree %d step %d" % (degree, step))
        st = gpu.stats()
        assert (st["stripCount"] > 0) == want_strips, st


def test_consecutive_resident_steps_with_joints_and_contacts():
    """A world the soft kernels cannot take -- joints inside the big island -- stepped for 6 steps on the op interpreter
    against the oracle chain: platform of boxes with a chain of revolute joints hung into it."""
    b, c, j = synthetic.pyramid(60)
    grid_b, grid_c, grid_j = synthetic.joint_grid(20)
    # glue: shift the grid's body indices behind the pyramid's and joint its first body to a box of the pile
    nb = len(b)
    grid_j = grid_j.copy()
    grid_j["bodyA"] += nb
    grid_j["bodyB"] += nb
    link = grid_j[:1].copy()
    link["bodyA"], link["bodyB"] = nb // 2, nb
    bodies = np.concatenate([b, grid_b])
    joints = np.concatenate([j, grid_j, link])
    pre = (bodies, c, joints)
    for solver_name in ("TGS_Soft", "PGS_NGS_Block"):
        vel, pos = common.DEFAULT_ITERS[solver_name]
        with hip.Solver(0) as s:
            s.set_option("strip_patience", 0)
            s.set_option("strip_min_bodies", 0)
            state = common.copy3(pre)
            for step in range(6):
                params = wire.StepParams.make(solver_name, 1.0 / 60.0, vel, pos, True)
                state = gpu_vs_oracle(s, params, state, "pile+grid/%s step %d" % (solver_name, step))
            st = s.stats()
            assert st["persistent"] == 1 and st["pairLanes"] == 3, st


def test_hand_off_timeout_falls_back():
    """Fault injection (persist_debug 8: workgroup 1 never publishes): the interpreter's hand-off times out, the epilogue leaves the
    wire arrays untouched, the step is repeated on another path, the result is still the oracle's."""
    solver_name = "PGS_NGS_Block"
    vel, pos = common.DEFAULT_ITERS[solver_name]
    pre = synthetic.pyramid(72)
    with hip.Solver(0) as s:
        s.set_option("strip_patience", 0)
        s.set_option("strip_min_bodies", 0)
        s.set_option("persist_spin_limit", 2000)
        s.set_option("persist_debug", 8)
        params = wire.StepParams.make(solver_name, 1.0 / 60.0, vel, pos, True)
        state = gpu_vs_oracle(s, params, pre, "pyramid72 fault")
        st = s.stats()
        assert st["persistFallbacks"] == 1 and st["persistent"] == 0, st
        gpu_vs_oracle(s, params, state, "pyramid72 after the fault")
