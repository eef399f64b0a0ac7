"""Created contacts are placed into the existing constraint-graph structure (solver2d_amd/csrc/solver_incremental.cpp): a
free position of a colour batch that is unused on both bodies, two entries in the bodies' incidence lists -- no rebuild,
same launch sequence, same captured step graph.  The result is one more valid sweep order: the C-ABI output must equal
the oracle swept in the order the library reports, bit for bit, and the colouring must stay proper (no dynamic body twice
in one colour batch)."""
import numpy as np
import pytest

from solver2d_amd import hip, synthetic, wire
from tests import common, oraclebind
from tests.test_gpu_parity import gpu_vs_oracle, gpu_vs_oracle_loose

pytestmark = pytest.mark.gpu


def _with_spare_slots(state, spare):
    b, c, j = state
    free = np.zeros(spare, dtype=wire.contact_dtype)
    free["bodyA"], free["bodyB"], free["constraintIndex"] = -1, -1, -1  # free pool slots as the reference binding packs them
    return b, np.concatenate([c, free]), j


def _artificial_contact(rng, bodies, template, exclude):
    """A two- or one-point manifold between two dynamic boxes that share no contact yet (the geometry is made up: this
    is a solver input, not a scene)."""
    dyn = np.flatnonzero(bodies["type"] == wire.BODY_DYNAMIC)
    while True:
        a, b = (int(x) for x in rng.choice(dyn, size=2, replace=False))
        if (min(a, b), max(a, b)) not in exclude:
            break
    exclude.add((min(a, b), max(a, b)))
    c = template.copy()
    c["bodyA"], c["bodyB"] = a, b
    c["pointCount"] = int(rng.integers(1, 3))
    ang = rng.uniform(0, 2 * np.pi)
    c["normal"] = (np.cos(ang), np.sin(ang))
    c["friction"] = rng.uniform(0.2, 0.9)
    for p in range(2):
        c["points"][p]["localAnchorA"] = rng.uniform(-0.5, 0.5, 2)
        c["points"][p]["localAnchorB"] = rng.uniform(-0.5, 0.5, 2)
        c["points"][p]["separation"] = rng.uniform(-0.02, 0.01)
        c["points"][p]["normalImpulse"] = 0.0
        c["points"][p]["tangentImpulse"] = 0.0
    return c


@pytest.mark.parametrize("solver_name", wire.SOLVER_NAMES)
def test_created_contacts_are_placed_without_a_rebuild(solver_name):
    vel, pos = common.DEFAULT_ITERS[solver_name]
    params = wire.StepParams.make(solver_name, 1.0 / 60.0, vel, pos, True)
    rng = np.random.default_rng(wire.SOLVER_ID[solver_name])
    pre = _with_spare_slots(synthetic.pyramid(30), 200)
    n0 = len(pre[1]) - 200
    pairs = {(min(a, b), max(a, b)) for a, b in zip(pre[1]["bodyA"][:n0].tolist(), pre[1]["bodyB"][:n0].tolist())}
    free_slot = np.zeros(1, dtype=wire.contact_dtype)[0]
    free_slot["bodyA"], free_slot["bodyB"], free_slot["constraintIndex"] = -1, -1, -1
    with hip.Solver(0) as s:
        s.set_option("groups", 0)  # the big-island path: colour batches over HBM-resident bodies
        s.set_option("strips", 0)
        state = gpu_vs_oracle(s, params, pre, "%s step 0" % solver_name)
        builds = s.stats()["structureBuilds"]
        spare = list(range(n0, n0 + 200))
        created = rebuilt_steps = 0
        for step in range(1, 13):
            # create a few contacts, destroy a few, and put a NEW pair into a slot whose contact was destroyed earlier
            for _ in range(int(rng.integers(1, 6))):
                state[1][spare.pop(0)] = _artificial_contact(rng, state[0], pre[1][5], pairs)
                created += 1
            if step % 2 == 0:
                victim = int(rng.integers(0, n0))
                state[1][victim] = free_slot
                spare.append(victim)
            # (s2Solve_Jacobi's contact pass writes no body: a placed contact takes any free position there, whatever colour the
            # batch had at build time -- the colour offsets of its reported order carry no meaning)
            check = gpu_vs_oracle_loose if solver_name == "Jacobi" else gpu_vs_oracle
            state = check(s, params, state, "%s step %d" % (solver_name, step))
            st = s.stats()
            rebuilt_steps += 1 if st["hostPrepMs"] > 0.0 else 0
            assert st["constraintCount"] == int((state[1]["pointCount"] > 0).sum()), st
        st = s.stats()
        # A box inside the pyramid already uses all six colours, so the first contact that lands on one needs a colour that does
        # not exist: ONE rebuild, which also lays out two spare (empty) colour batches; after that contacts are placed again.
        # s2Solve_Jacobi needs no colours at all: never a rebuild.
        assert st["structureBuilds"] - builds == rebuilt_steps <= (0 if solver_name == "Jacobi" else 3), (st, rebuilt_steps)
        assert st["placedContacts"] >= created - 5 * rebuilt_steps > 0, (st, created)


def test_a_contact_that_fits_no_colour_rebuilds_and_stays_exact():
    """A body that gets more contacts than there are colour batches (the seventh contact of a box inside a pyramid needs a
    seventh colour) cannot be placed: the structure is rebuilt for that step -- once -- and later contacts are placed again."""
    params = wire.StepParams.make("TGS_Soft", 1.0 / 60.0, 8, 4, True)
    rng = np.random.default_rng(5)
    pre = _with_spare_slots(synthetic.pyramid(20), 64)
    n0 = len(pre[1]) - 64
    deg = np.bincount(np.concatenate([pre[1]["bodyA"][:n0], pre[1]["bodyB"][:n0]]), minlength=len(pre[0]))
    hub = int(np.argmax(deg * (pre[0]["type"] == wire.BODY_DYNAMIC)))
    with hip.Solver(0) as s:
        s.set_option("groups", 0)
        s.set_option("strips", 0)
        state = gpu_vs_oracle(s, params, pre, "hub step 0")
        colours = s.stats()["contactColors"]
        assert deg[hub] == colours == 6
        builds = s.stats()["structureBuilds"]
        others = [i for i in np.flatnonzero(pre[0]["type"] == wire.BODY_DYNAMIC).tolist() if i != hub][-3:]
        for n, other in enumerate(others):
            c = _artificial_contact(rng, state[0], pre[1][5], set())
            c["bodyA"], c["bodyB"] = hub, other
            state[1][n0 + n] = c
            state = gpu_vs_oracle(s, params, state, "hub step %d" % (n + 1))
        st = s.stats()
        # the hub's 7th contact needed a colour that did not exist: one rebuild, which added two spare colour batches -- where its
        # 8th and 9th contact were then placed
        assert st["structureBuilds"] == builds + 1 and st["placedContacts"] == 2, st
        builds = st["structureBuilds"]
        # an ordinary contact afterwards is placed again
        state[1][n0 + 10] = _artificial_contact(rng, state[0], pre[1][5], set())
        gpu_vs_oracle(s, params, state, "hub step after")
        assert s.stats()["structureBuilds"] == builds and s.stats()["placedContacts"] == 3


def test_incremental_off_rebuilds_every_time_with_identical_results():
    """Option "incremental" = 0 keeps the old behaviour (every created contact rebuilds the structure); the bits of the
    bodies may differ between the two settings only through the sweep order, so each is checked against the oracle in
    its own order."""
    params = wire.StepParams.make("PGS_Soft", 1.0 / 60.0, 4, 2, True)
    for inc in (1, 0):
        rng = np.random.default_rng(9)
        pre = _with_spare_slots(synthetic.pyramid(16), 32)
        n0 = len(pre[1]) - 32
        with hip.Solver(0) as s:
            s.set_option("groups", 0)
            s.set_option("incremental", inc)
            state = gpu_vs_oracle(s, params, pre, "inc=%d step 0" % inc)
            builds = s.stats()["structureBuilds"]
            for step in range(1, 5):
                state[1][n0 + step] = _artificial_contact(rng, state[0], pre[1][5], set())
                state = gpu_vs_oracle(s, params, state, "inc=%d step %d" % (inc, step))
            assert (s.stats()["structureBuilds"] <= builds + 1) == (inc == 1), s.stats()


def test_heavy_body_list_follows_the_incidence_lists():
    """A body whose incidence list grows past S2_HEAVY_DEGREE (12) entries moves to the wave-per-body path of the
    body-centric kernels (and back when it shrinks): Jacobi sums and the body-centric warm start must stay exact."""
    rng = np.random.default_rng(21)
    for solver_name in ("Jacobi", "TGS_Soft"):
        vel, pos = common.DEFAULT_ITERS[solver_name]
        params = wire.StepParams.make(solver_name, 1.0 / 60.0, vel, pos, True)
        pre = _with_spare_slots(synthetic.platform(10, layers=2), 64)
        n0 = len(pre[1]) - 64
        with hip.Solver(0) as s:
            s.set_option("groups", 0)
            s.set_option("strips", 0)
            state = gpu_vs_oracle(s, params, pre, "%s platform step 0" % solver_name)
            plat = 1  # the platform touches the ground and ten boxes: 11 contacts
            top = [i for i in range(12, 22)]
            for n in range(6):  # 12, 13, 14 ... entries: crosses the threshold; under TGS_Soft these need new colours (rebuilds), under Jacobi any position does
                c = _artificial_contact(rng, state[0], pre[1][1], set())
                c["bodyA"], c["bodyB"] = plat, top[n]
                state[1][n0 + n] = c
                state = gpu_vs_oracle_loose(s, params, state, "%s platform step %d" % (solver_name, n + 1))
            for n in range(4):  # and shrinks again (destroyed contacts linger; a re-used slot removes the entry)
                c = _artificial_contact(rng, state[0], pre[1][1], set())
                c["bodyA"], c["bodyB"] = top[n], top[n + 5]
                state[1][n0 + n] = c
                state = gpu_vs_oracle_loose(s, params, state, "%s platform shrink %d" % (solver_name, n))


def test_a_contact_without_points_between_strip_bodies_is_only_watched():
    """Option "defer" (default on): a created contact has no manifold points yet.  Where it cannot be placed -- both boxes
    belong to strips, whose tables are register / LDS layouts -- it gets no entry in the structure and the persistent strip
    kernel keeps running; its first points are the change of the graph.  With the option off its creation already is."""
    params = wire.StepParams.make("TGS_Soft", 1.0 / 60.0, 8, 4, True)
    for defer in (1, 0):
        rng = np.random.default_rng(31)
        pre = _with_spare_slots(synthetic.pyramid(80), 16)
        n0 = len(pre[1]) - 16
        pairs = {(min(a, b), max(a, b)) for a, b in zip(pre[1]["bodyA"][:n0].tolist(), pre[1]["bodyB"][:n0].tolist())}
        with hip.Solver(0) as s:
            s.set_option("strip_patience", 0)
            s.set_option("strip_min_bodies", 0)  # (a base-80 pyramid is below the size at which strips are the default)
            s.set_option("defer", defer)
            state = gpu_vs_oracle(s, params, pre, "defer=%d step 0" % defer)
            st = s.stats()
            assert st["stripCount"] > 0 and st["persistent"] == 1, st
            builds = st["structureBuilds"]
            # a potential contact: two boxes whose fat boxes overlap, nothing touching yet
            c = _artificial_contact(rng, state[0], pre[1][5], pairs)
            c["pointCount"] = 0
            state[1][n0] = c
            state = gpu_vs_oracle(s, params, state, "defer=%d potential contact" % defer)
            st = s.stats()
            if defer:
                assert st["structureBuilds"] == builds and st["persistent"] == 1, st
            else:
                assert st["structureBuilds"] > builds, st
            builds = st["structureBuilds"]
            state = gpu_vs_oracle(s, params, state, "defer=%d quiet step" % defer)
            assert s.stats()["structureBuilds"] == builds
            # its first points: now it is a constraint, and the structure has to hold it
            state[1][n0]["pointCount"] = 2
            state = gpu_vs_oracle(s, params, state, "defer=%d first points" % defer)
            st = s.stats()
            assert (st["structureBuilds"] > builds) == bool(defer), st
            assert st["constraintCount"] == int((state[1]["pointCount"] > 0).sum())
            # ... and a deferred contact that is destroyed before it ever touched leaves no trace
            c2 = _artificial_contact(rng, state[0], pre[1][5], pairs)
            c2["pointCount"] = 0
            state[1][n0 + 1] = c2
            state = gpu_vs_oracle(s, params, state, "defer=%d second potential contact" % defer)
            state[1][n0 + 1]["bodyA"], state[1][n0 + 1]["bodyB"] = -1, -1
            gpu_vs_oracle(s, params, state, "defer=%d destroyed untouched" % defer)
