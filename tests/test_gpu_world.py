"""Resident world chain on the device (s2amd_world_upload / _step / _download, solver2d_amd/csrc/world.hip): stage 3
update contacts -> s2Solve_* -> stage 4 refit on arrays that stay in HBM, against the same chain through the CPU oracle
(tests/world_chain.py, pinned to the reference by tests/test_world_chain.py) with the solve swept in the order the
device reports.  Bit for bit, every array, every step."""
import glob
import os

import numpy as np
import pytest

from solver2d_amd import hip, synthetic, wire
from tests import world_chain

pytestmark = pytest.mark.gpu

FILES = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "world_*.npz")))


def run_chain(s, params, world, steps, what):
    """Uploads `world`, steps both sides, compares after every step; returns the device's last state and the infos."""
    ref = world_chain.copy_world(world)
    s.world_upload(*[world[k] for k in world_chain.WORLD_KEYS])
    infos = []
    got = None
    for step in range(steps):
        info = s.world_step(params)
        order, _ = s.contact_order()
        jorder, _ = s.joint_order()
        status = world_chain.oracle_world_step(params, ref, contact_order=order, joint_order=jorder)
        out = world_chain.copy_world(world)
        res = s.world_download(*[out[k] for k in world_chain.WORLD_KEYS])
        got = dict(zip(world_chain.WORLD_KEYS, res[:6]))
        world_chain.assert_device_equals_oracle(got, ref, "%s step %d" % (what, step))
        assert np.array_equal(res[6], status), "%s step %d: stage-3 status" % (what, step)
        assert info["separatedCount"] == int((status == wire.PAIR_SEPARATED).sum())
        assert info["activeContacts"] == int((ref["contacts"]["pointCount"] > 0).sum())
        live_shapes = ref["shapes"]["type"] != wire.SHAPE_FREE
        assert info["movedCount"] == int((ref["shapes"]["enlarged"][live_shapes] != 0).sum())
        infos.append(info)
    return got, infos


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(p)[6:-4] for p in FILES])
def test_golden_world_chains(path):
    d = np.load(path)
    params = world_chain.params_of(d)
    world = world_chain.load_world(d)
    with hip.Solver(0) as s:
        run_chain(s, params, world, int(d["steps"][0]) + 3, os.path.basename(path))


@pytest.mark.parametrize("solver_name", wire.SOLVER_NAMES)
def test_golden_world_every_solver(solver_name):
    """The shapes-zoo world (all shape types, joints) through every driver."""
    from tests import common
    path = [f for f in FILES if "shapes_zoo" in f][0]
    d = np.load(path)
    vel, pos = common.DEFAULT_ITERS[solver_name]
    params = wire.StepParams.make(solver_name, 1.0 / 60.0, vel, pos, True)
    with hip.Solver(0) as s:
        run_chain(s, params, world_chain.load_world(d), 4, "shapes_zoo/" + solver_name)


def test_big_pyramid_world_runs_the_persistent_strips():
    """Base-100 pyramid (5,050 boxes, 14,950 manifolds recomputed by the device narrow phase every step): after the
    patience step the solve inside the chain is the persistent strip kernel."""
    params = wire.StepParams.make("TGS_Soft", 1.0 / 60.0, 8, 4, True)
    world = synthetic.pyramid_world(100)
    with hip.Solver(0) as s:
        _, infos = run_chain(s, params, world, 4, "pyramid100 world")
        st = s.stats()
        assert st["persistent"] == 1 and st["stripCount"] > 1, st
        assert all(i["activeContacts"] == 14950 and i["separatedCount"] == 0 for i in infos), infos


def test_world_chain_follows_a_changing_contact_graph():
    """The top box starts 5 cm above its seat: its two manifolds have no points (beyond the speculative distance, inside
    the fat AABBs), gain them when it lands, and the device must rebuild the solve structure in exactly those steps."""
    params = wire.StepParams.make("TGS_Soft", 1.0 / 60.0, 8, 4, True)
    world = synthetic.pyramid_world(12)
    top = len(world["bodies"]) - 1
    world["bodies"]["position"][top][1] += np.float32(0.05)
    world["origins"][top][1] += np.float32(0.05)
    with hip.Solver(0) as s:
        _, infos = run_chain(s, params, world, 12, "pyramid12 with a falling top box")
        active = [i["activeContacts"] for i in infos]
        changed = [i["graphChanged"] for i in infos]
        assert active[0] == len(world["contacts"]) - 2 and active[-1] == len(world["contacts"]), active
        assert changed[0] == 1 and sum(changed) >= 2, changed
        for a0, a1, c in zip(active[:-1], active[1:], changed[1:]):
            assert (a0 != a1) <= bool(c)


def test_world_chain_destroys_separated_pairs():
    """A box shot upwards leaves the fat AABBs of its neighbours: stage 3 reports the pairs separated, the device frees
    them (no manifold, pair slot free) as src/world.c:149-167 does, the chain goes on."""
    params = wire.StepParams.make("TGS_Soft", 1.0 / 60.0, 8, 4, True)
    world = synthetic.pyramid_world(12)
    top = len(world["bodies"]) - 1
    world["bodies"]["linearVelocity"][top] = (0.0, 40.0)
    with hip.Solver(0) as s:
        got, infos = run_chain(s, params, world, 5, "pyramid12 losing its top box")
        assert sum(i["separatedCount"] for i in infos) == 2, infos
        assert infos[0]["movedCount"] >= 1
        assert int((got["pairs"]["shapeA"] < 0).sum()) == 2


def test_world_step_consumes_applied_forces():
    params = wire.StepParams.make("PGS", 1.0 / 60.0, 4, 2, True)
    world = synthetic.pyramid_world(6)
    world["bodies"]["force"][3] = (25.0, 5.0)
    world["bodies"]["torque"][4] = 2.0
    with hip.Solver(0) as s:
        got, _ = run_chain(s, params, world, 2, "pyramid6 with forces")
        assert not got["bodies"]["force"].any() and not got["bodies"]["torque"].any()


def test_world_api_state_errors():
    params = wire.StepParams.make("PGS", 1.0 / 60.0, 4, 2, True)
    with hip.Solver(0) as s:
        with pytest.raises(hip.S2AmdError):
            s.world_step(params)
        world = synthetic.pyramid_world(4)
        bad = world_chain.copy_world(world)
        bad["pairs"]["shapeB"][0] = len(world["shapes"]) + 5
        with pytest.raises(hip.S2AmdError):
            s.world_upload(*[bad[k] for k in world_chain.WORLD_KEYS])
        s.upload(world["bodies"], world["contacts"], world["joints"])
        with pytest.raises(hip.S2AmdError):
            s.world_step(params)  # plain upload: no shapes resident


def _live_pairs(world):
    live = world["pairs"]["shapeA"] >= 0
    return np.stack([world["pairs"]["shapeA"][live], world["pairs"]["shapeB"][live]], axis=1).astype(np.int32)


def _create_contacts(world, new_pairs):
    """The caller's side of s2CreateContact (src/contact.c:137-203) for worlds of default-friction shapes: first free
    slot, empty manifold."""
    slots, contacts, pairs = [], [], []
    free = np.flatnonzero(world["pairs"]["shapeA"] < 0).tolist()
    for a, b in new_pairs.tolist():
        k = free.pop(0)
        c = np.zeros(1, dtype=wire.contact_dtype)[0]
        c["bodyA"], c["bodyB"] = world["shapes"]["body"][a], world["shapes"]["body"][b]
        c["friction"] = 0.6
        c["constraintIndex"] = -1
        p = np.zeros(1, dtype=wire.pair_state_dtype)[0]
        p["shapeA"], p["shapeB"] = a, b
        world["contacts"][k] = c
        world["pairs"][k] = p
        slots.append(k)
        contacts.append(c)
        pairs.append(p)
    return np.array(slots, dtype=np.int32), np.array(contacts, dtype=wire.contact_dtype), np.array(pairs, dtype=wire.pair_state_dtype)


def test_world_loop_with_pair_creation():
    """The whole s2World_Step loop with the control plane on the host: the top box hops off (its pairs separate and are
    destroyed on the device), lands again (the refit moves its proxy, s2amd_world_find_pairs reports the new pairs, the
    caller creates the contacts with s2amd_world_set_contacts).  Every step bit-exact against the oracle chain, every
    pair query equal to the oracle's."""
    from tests import oraclebind
    params = wire.StepParams.make("TGS_Soft", 1.0 / 60.0, 8, 4, True)
    world = synthetic.pyramid_world(12)
    top = len(world["bodies"]) - 1
    world["bodies"]["linearVelocity"][top] = (0.0, 3.0)
    ref = world_chain.copy_world(world)
    separated = created = queries = 0
    with hip.Solver(0) as s:
        s.world_upload(*[world[k] for k in world_chain.WORLD_KEYS])
        for step in range(60):
            info = s.world_step(params)
            order, _ = s.contact_order()
            status = world_chain.oracle_world_step(params, ref, contact_order=order)
            separated += info["separatedCount"]
            assert info["separatedCount"] == int((status == wire.PAIR_SEPARATED).sum())
            if info["movedCount"] > 0:
                queries += 1
                got = s.world_find_pairs()
                want = world_chain.oracle_find_pairs(ref)
                assert np.array_equal(got, want), "step %d: new pairs" % step
                if len(got):
                    created += len(got)
                    slots, contacts, pairs = _create_contacts(ref, got)
                    s.world_set_contacts(slots, contacts, pairs)
            out = world_chain.copy_world(world)
            res = s.world_download(*[out[k] for k in world_chain.WORLD_KEYS])
            world_chain.assert_device_equals_oracle(dict(zip(world_chain.WORLD_KEYS, res[:6])), ref, "hop step %d" % step)
        assert separated == 2 and created == 2 and queries >= 2, (separated, created, queries)
        assert int((ref["contacts"]["pointCount"] > 0).sum()) == len(ref["contacts"])


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(p)[6:-4] for p in FILES])
def test_resident_pair_query_equals_the_stage_function(path):
    """s2amd_world_find_pairs on the golden worlds (joints, kinematic bodies, all shape types) whenever the refit moved
    something, against the oracle's pair discovery on the same state."""
    from tests import oraclebind
    d = np.load(path)
    params = world_chain.params_of(d)
    world = world_chain.load_world(d)
    ref = world_chain.copy_world(world)
    asked = 0
    with hip.Solver(0) as s:
        s.world_upload(*[world[k] for k in world_chain.WORLD_KEYS])
        for step in range(8):
            info = s.world_step(params)
            order, _ = s.contact_order()
            jorder, _ = s.joint_order()
            world_chain.oracle_world_step(params, ref, contact_order=order, joint_order=jorder)
            if info["movedCount"] > 0:
                asked += 1
                want = world_chain.oracle_find_pairs(ref)
                assert np.array_equal(s.world_find_pairs(), want), "step %d" % step
    if any(name in os.path.basename(path) for name in ("mixed", "shapes_zoo", "circle_pile", "tumbler")):
        assert asked > 0


def test_tumbler_world_loop():
    """The whole loop on the messiest world there is: a drum (four shapes on one motor-driven body) full of boxes that
    never goes three steps without gaining and losing contacts.  Forty steps of s2amd_world_step, with every pair query
    compared with the oracle's and every new pair turned into a contact by the caller, bit-exact against the oracle chain
    fed the same contacts."""
    from tests import oraclebind
    path = [f for f in FILES if "tumbler" in f][0]
    d = np.load(path)
    params = world_chain.params_of(d)
    world = world_chain.load_world(d)
    ref = world_chain.copy_world(world)
    separated = created = 0
    with hip.Solver(0) as s:
        s.world_upload(*[world[k] for k in world_chain.WORLD_KEYS])
        for step in range(40):
            info = s.world_step(params)
            order, _ = s.contact_order()
            jorder, _ = s.joint_order()
            status = world_chain.oracle_world_step(params, ref, contact_order=order, joint_order=jorder)
            separated += info["separatedCount"]
            assert info["separatedCount"] == int((status == wire.PAIR_SEPARATED).sum())
            if info["movedCount"] > 0:
                got = s.world_find_pairs()
                want = world_chain.oracle_find_pairs(ref)
                assert np.array_equal(got, want), "step %d: new pairs" % step
                if len(got):
                    created += len(got)
                    slots, contacts, pairs = _create_contacts(ref, got)
                    s.world_set_contacts(slots, contacts, pairs)
            out = world_chain.copy_world(world)
            res = s.world_download(*[out[k] for k in world_chain.WORLD_KEYS])
            world_chain.assert_device_equals_oracle(dict(zip(world_chain.WORLD_KEYS, res[:6])), ref, "tumbler step %d" % step)
    assert separated > 5 and created > 5, (separated, created)


def test_world_from_scratch_finds_every_touching_pair():
    """A pyramid world uploaded with NO contacts and every proxy in the move buffer (as after creation): the resident
    pair query must report exactly the touching pairs of the scene, the caller creates them, and from there the chain
    runs bit-exact against the oracle chain given the same contacts."""
    params = wire.StepParams.make("TGS_Soft", 1.0 / 60.0, 8, 4, True)
    full = synthetic.pyramid_world(14)
    expected = {(min(a, b), max(a, b)) for a, b in zip(full["pairs"]["shapeA"].tolist(), full["pairs"]["shapeB"].tolist())}
    world = world_chain.copy_world(full)
    world["contacts"][:] = np.zeros(1, dtype=wire.contact_dtype)[0]
    world["contacts"]["constraintIndex"] = -1
    world["pairs"]["shapeA"] = -1
    world["pairs"]["shapeB"] = -1
    world["shapes"]["enlarged"] = 1
    ref = world_chain.copy_world(world)
    with hip.Solver(0) as s:
        s.world_upload(*[world[k] for k in world_chain.WORLD_KEYS])
        got = s.world_find_pairs()
        ref["shapes"]["enlarged"] = 0  # the query consumed the move buffer
        assert {(min(a, b), max(a, b)) for a, b in got.tolist()} == expected and len(got) == len(expected)
        slots, contacts, pairs = _create_contacts(ref, got)
        s.world_set_contacts(slots, contacts, pairs)
        for step in range(4):
            info = s.world_step(params)
            order, _ = s.contact_order()
            world_chain.oracle_world_step(params, ref, contact_order=order)
            out = world_chain.copy_world(world)
            res = s.world_download(*[out[k] for k in world_chain.WORLD_KEYS])
            world_chain.assert_device_equals_oracle(dict(zip(world_chain.WORLD_KEYS, res[:6])), ref, "from scratch step %d" % step)
            assert info["activeContacts"] == len(expected)


def test_static_shape_with_the_higher_proxy_key_leaves_the_move_buffer():
    """A static shape uploaded "in the move buffer" (enlarged = 1, as after its creation) whose proxy key is HIGHER than
    that of a dynamic box falling towards it.  The reference clears the move buffer at the end of every pair update
    (src/broad_phase.c: s2UpdateBroadPhasePairs), so the ground is a moved proxy for one query only; if its flag stuck,
    the falling box's query would skip the pair for good ("both moved: the lower key reports", and the static shape
    never queries) and the box would tunnel.  Whole loop against the oracle chain; the box must come to rest on the ground."""
    params = wire.StepParams.make("TGS_Soft", 1.0 / 60.0, 8, 4, True)
    bodies = np.zeros(2, dtype=wire.body_dtype)
    synthetic._dynamic_body(bodies[0], 0.0, 1.2, synthetic.BOX_MASS, synthetic.BOX_I)
    synthetic._static_body(bodies[1], 0.0, -1.0)
    shapes = np.zeros(2, dtype=wire.shape_dtype)
    synthetic._box_shape(shapes[0], 0, wire.BODY_DYNAMIC, 0.5, 0.5, 0.0, 1.2, 0)
    synthetic._box_shape(shapes[1], 1, wire.BODY_STATIC, 20.0, 1.0, 0.0, -1.0, 1)
    assert shapes["proxyKey"][1] > shapes["proxyKey"][0]
    shapes["enlarged"] = 1
    contacts = np.zeros(4, dtype=wire.contact_dtype)
    contacts["constraintIndex"] = -1
    pairs = np.zeros(4, dtype=wire.pair_state_dtype)
    pairs["shapeA"] = -1
    pairs["shapeB"] = -1
    world = {"bodies": bodies, "contacts": contacts, "joints": np.zeros(0, dtype=wire.joint_dtype), "shapes": shapes, "pairs": pairs,
             "origins": np.ascontiguousarray(bodies["position"], dtype=np.float32).copy()}
    ref = world_chain.copy_world(world)
    created_at = None
    with hip.Solver(0) as s:
        s.world_upload(*[world[k] for k in world_chain.WORLD_KEYS])
        for step in range(90):
            if world_chain.moved_any(ref):
                got = s.world_find_pairs()
                want = world_chain.oracle_find_pairs(ref)
                assert np.array_equal(got, want), "step %d" % step
                if len(got):
                    assert created_at is None and got.tolist() == [[0, 1]]
                    created_at = step
                    slots, cs, ps = _create_contacts(ref, got)
                    s.world_set_contacts(slots, cs, ps)
            info = s.world_step(params)
            order, _ = s.contact_order()
            world_chain.oracle_world_step(params, ref, contact_order=order)
            out = world_chain.copy_world(world)
            res = s.world_download(*[out[k] for k in world_chain.WORLD_KEYS])
            world_chain.assert_device_equals_oracle(dict(zip(world_chain.WORLD_KEYS, res[:6])), ref, "falling box step %d" % step)
            if step == 0:
                assert info["movedCount"] == 0 or not ref["shapes"]["enlarged"][1]  # the ground left the move buffer with the first query
        got_bodies = res[0]
    assert created_at is not None and created_at > 0, "the pair with the static ground was never reported"
    assert 0.45 < float(got_bodies["position"][0][1]) < 0.55 and abs(float(got_bodies["linearVelocity"][0][1])) < 0.05, got_bodies[0]


@pytest.mark.parametrize("seed,solver_name", [(1, "TGS_Soft"), (2, "PGS_Soft"), (3, "SoftStep"), (4, "TGS_Sticky"), (5, "XPBD"),
                                              (6, "Jacobi"), (7, "PGS"), (8, "PGS_NGS"), (9, "PGS_NGS_Block"), (10, "TGS_NGS")])
def test_rain_world_loop(seed, solver_name):
    """Fuzz of the whole loop: bodies of every shape type (two-shape bodies, rounded polygons, segments, a kinematic
    paddle) rain into a trough.  Each step: resident pair query == oracle's, the caller creates the contacts,
    s2amd_world_step, every array bit-exact against the oracle chain.  Contacts come and go all the time."""
    from tests import common, oraclebind
    vel, pos = common.DEFAULT_ITERS[solver_name]
    params = wire.StepParams.make(solver_name, 1.0 / 60.0, vel, pos, True)
    world = world_chain.rain_world(seed, 100 + 7 * seed)
    ref = world_chain.copy_world(world)
    separated = created = 0
    with hip.Solver(0) as s:
        s.world_upload(*[world[k] for k in world_chain.WORLD_KEYS])
        for step in range(70):
            if world_chain.moved_any(ref):
                got = s.world_find_pairs()
                want = world_chain.oracle_find_pairs(ref)
                assert np.array_equal(got, want), "step %d: new pairs" % step
                if len(got):
                    created += len(got)
                    slots, contacts, pairs = _create_contacts(ref, got)
                    s.world_set_contacts(slots, contacts, pairs)
            info = s.world_step(params)
            order, _ = s.contact_order()
            status = world_chain.oracle_world_step(params, ref, contact_order=order)
            separated += info["separatedCount"]
            assert info["separatedCount"] == int((status == wire.PAIR_SEPARATED).sum()), "step %d" % step
            if step % 3 == 2 or step < 5:
                out = world_chain.copy_world(world)
                res = s.world_download(*[out[k] for k in world_chain.WORLD_KEYS])
                world_chain.assert_device_equals_oracle(dict(zip(world_chain.WORLD_KEYS, res[:6])), ref,
                                                        "rain %d %s step %d" % (seed, solver_name, step))
    assert separated > 20 and created > 100, (separated, created)


@pytest.mark.parametrize("seed,solver_name", [(2, "PGS_Soft"), (6, "Jacobi"), (11, "TGS_Soft")])
def test_rain_world_loop_with_the_pair_query_on_demand(seed, solver_name):
    """The same loop with option `pairs_in_step` 0: every s2amd_world_find_pairs runs its query itself (the round-4 form) instead of
    collecting the one s2amd_world_step enqueued behind its stage 4 -- the route a caller takes that sets a contact between a step and
    its query, or whose step was repeated.  Same pairs as the oracle, same world; and the created pairs' small directory overflows
    (more than 255 contacts created) on the way, in both forms."""
    from tests import common
    vel, pos = common.DEFAULT_ITERS[solver_name]
    params = wire.StepParams.make(solver_name, 1.0 / 60.0, vel, pos, True)
    world = world_chain.rain_world(seed, 100 + 7 * seed)
    ref = world_chain.copy_world(world)
    created = 0
    with hip.Solver(0) as s:
        s.set_option("pairs_in_step", 0)
        s.world_upload(*[world[k] for k in world_chain.WORLD_KEYS])
        for step in range(70):
            if world_chain.moved_any(ref):
                got = s.world_find_pairs()
                want = world_chain.oracle_find_pairs(ref)
                assert np.array_equal(got, want), "step %d: new pairs" % step
                if len(got):
                    created += len(got)
                    slots, contacts, pairs = _create_contacts(ref, got)
                    s.world_set_contacts(slots, contacts, pairs)
            s.world_step(params)
            order, _ = s.contact_order()
            world_chain.oracle_world_step(params, ref, contact_order=order)
        out = world_chain.copy_world(world)
        res = s.world_download(*[out[k] for k in world_chain.WORLD_KEYS])
        world_chain.assert_device_equals_oracle(dict(zip(world_chain.WORLD_KEYS, res[:6])), ref, "rain %d %s, query on demand" % (seed, solver_name))
    assert created > 255, created


@pytest.mark.parametrize("seed,solver_name", [(0, "TGS_Soft"), (14, "PGS_Soft"), (19, "SoftStep"), (4, "PGS"), (5, "XPBD")])
def test_wrecking_ball_world_loop(seed, solver_name):
    """Heavy balls shot into a pyramid, whole loop (pair query, contact creation, s2amd_world_step) for 70 steps with the
    strip options drawn at random: bursts of graph changes (structure rebuilt, patience counter) alternate with quiet
    stretches on the strip / persistent kernels.  Bit-exact against the oracle chain; a one-off run of 110 seeds found
    no difference."""
    from tests import common, oraclebind
    rng = np.random.default_rng(1000 + seed)
    base = int(rng.integers(30, 75))
    world = world_chain.wreck_world(seed, base)
    vel, pos = common.DEFAULT_ITERS[solver_name]
    params = wire.StepParams.make(solver_name, 1.0 / 60.0, vel, pos, True)
    ref = world_chain.copy_world(world)
    changed = strips = created = 0
    with hip.Solver(0) as s:
        s.set_option("strip_patience", int(rng.integers(0, 4)))
        s.set_option("strip_min_bodies", 0)
        s.set_option("strip_bodies", int(rng.integers(40, 200)))
        s.set_option("max_group_bodies", int(rng.choice([64, 512, 2816])))
        s.world_upload(*[world[k] for k in world_chain.WORLD_KEYS])
        for step in range(70):
            if world_chain.moved_any(ref):
                got = s.world_find_pairs()
                want = world_chain.oracle_find_pairs(ref)
                assert np.array_equal(got, want), "step %d: new pairs" % step
                if len(got):
                    created += len(got)
                    slots, contacts, pairs = _create_contacts(ref, got)
                    s.world_set_contacts(slots, contacts, pairs)
            info = s.world_step(params)
            order, _ = s.contact_order()
            status = world_chain.oracle_world_step(params, ref, contact_order=order)
            assert info["separatedCount"] == int((status == wire.PAIR_SEPARATED).sum()), "step %d" % step
            changed += info["graphChanged"]
            strips += 1 if s.stats()["stripCount"] > 0 else 0
            if step % 4 == 3 or step < 3:
                out = world_chain.copy_world(world)
                res = s.world_download(*[out[k] for k in world_chain.WORLD_KEYS])
                world_chain.assert_device_equals_oracle(dict(zip(world_chain.WORLD_KEYS, res[:6])), ref,
                                                        "wreck %d %s step %d" % (seed, solver_name, step))
    assert changed > 20 and created > 50, (changed, created)
    if seed in (0, 14, 19):
        assert strips > 0


# (s2Solve_SoftStep runs on the 512-thread kernel in its <3, 2> layout only -- wide_kernel.hip -- and a pile with a ball in it needs more
# rounds: it keeps the build in the step, as before round 5)
@pytest.mark.parametrize("overflow_kernel", [1, 0])
@pytest.mark.parametrize("seed,solver_name", [(1, "TGS_Soft"), (3, "TGS_Soft"), (7, "TGS_Soft"), (6, "PGS_Soft")])
def test_a_contact_that_fits_nowhere_in_the_strips_waits_behind_them(seed, solver_name, overflow_kernel):
    """SURVEY.md 8f row 4 (round 5): a ball that comes to touch boxes two strips apart used to cost a structure build in the step that
    found the contact (5 ms on the caller's thread at base 200).  Now the contact takes an OVERFLOW position behind the strips
    (solver_internal.h: IncrementalStrips), the steps run SLICED -- the persistent kernel launched once per sweep, the overflow contacts
    swept behind each launch -- while a worker thread builds the structure that holds it, adopted at a step boundary.  The whole loop at
    base 100 with the default options, 150 steps, every step of it bit-exact against the oracle chain swept in the device's order -- through
    the sliced steps, the adoption, and the steps on the adopted structure; and no structure build in a step that found such a contact.
    `overflow_kernel` 1 (the default): the step stays ONE launch of the persistent kernel, which carries one more workgroup that sweeps
    the overflow contacts after every sweep of the strips (wide_kernel.hip: wideOverflowWorker; stats.slicedStep 2); 0: sliced steps."""
    from tests import common
    vel, pos = common.DEFAULT_ITERS[solver_name]
    params = wire.StepParams.make(solver_name, 1.0 / 60.0, vel, pos, True)
    world = world_chain.wreck_world(seed, 100)
    ref = world_chain.copy_world(world)
    sliced = adopted = 0
    rows = []
    with hip.Solver(0) as s:
        s.set_option("overflow_kernel", overflow_kernel)
        s.world_upload(*[world[k] for k in world_chain.WORLD_KEYS])
        for step in range(150):
            if world_chain.moved_any(ref):
                got = s.world_find_pairs()
                want = world_chain.oracle_find_pairs(ref)
                assert np.array_equal(got, want), "step %d: new pairs" % step
                if len(got):
                    slots, contacts, pairs = _create_contacts(ref, got)
                    s.world_set_contacts(slots, contacts, pairs)
            info = s.world_step(params)
            order, _ = s.contact_order()
            status = world_chain.oracle_world_step(params, ref, contact_order=order)
            assert info["separatedCount"] == int((status == wire.PAIR_SEPARATED).sum()), "step %d" % step
            st = s.stats()
            rows.append((st["overflowContacts"], st["slicedStep"], st["structureBuilds"], st["asyncBuildsAdopted"], st["kernelLaunches"]))
            assert st["slicedStep"] in (0, 2 if overflow_kernel else 1) and st["persistFallbacks"] == 0, (step, st["slicedStep"], st["persistFallbacks"])
            assert st["slicedStep"] != 2 or st["kernelLaunches"] <= 4, (step, st["kernelLaunches"])
            sliced += 1 if st["slicedStep"] else 0
            adopted = st["asyncBuildsAdopted"]
            if st["slicedStep"] or step % 8 == 7 or step < 3:
                out = world_chain.copy_world(world)
                res = s.world_download(*[out[k] for k in world_chain.WORLD_KEYS])
                world_chain.assert_device_equals_oracle(dict(zip(world_chain.WORLD_KEYS, res[:6])), ref,
                                                        "overflow %d %s step %d (overflow %d, sliced %d)" % (seed, solver_name, step, st["overflowContacts"], st["slicedStep"]))
    changes = [(i, r) for i, r in enumerate(rows) if i == 0 or r[:4] != rows[i - 1][:4]]  # (step, row) wherever anything but the launch count moved
    assert sliced > 0 and adopted >= 1, str(changes)
    for i in range(1, len(rows)):
        if rows[i][0] > rows[i - 1][0]:
            # one more contact in the overflow region: that step built nothing (an adoption in the same step counts as a build)
            assert rows[i][2] == rows[i - 1][2] or rows[i][3] > rows[i - 1][3], (i, rows[i - 1], rows[i])


@pytest.mark.parametrize("solver_name", ["TGS_Soft", "SoftStep", "PGS_Soft"])
def test_settling_pyramid_physical_tolerances(solver_name):
    """SURVEY.md 8c, parity link L3: the device (colour order, device narrow phase and refit) against the reference
    algorithm in POOL order (the oracle chain without the device's order) on a settling base-40 pyramid, 120 steps.
    Gauss-Seidel is order dependent -- the pool-order sweep itself pushes the top of this pile 20 cm sideways in 120
    steps -- so the position bound is relative, as in tests/test_gpu_dropin.py: the device may differ from the pool order
    by at most 1.5x what the REVERSED pool order differs from it (floor 2 cm).  Absolute bounds: no NaN; the pile at
    rest (|v| < 1 cm/s); the ground carries the pile -- the normal impulses of the ground manifolds sum to the weight
    times the sub-step within 1 %; the pyramid keeps its height (the top box sinks less than 15 cm: soft contacts
    compress by 3 mm per layer); the same number of live pairs."""
    from tests import common
    vel, pos = common.DEFAULT_ITERS[solver_name]
    params = wire.StepParams.make(solver_name, 1.0 / 60.0, vel, pos, True)
    world = synthetic.pyramid_world(40)
    ref = world_chain.copy_world(world)
    rev = world_chain.copy_world(world)
    with hip.Solver(0) as s:
        s.world_upload(*[world[k] for k in world_chain.WORLD_KEYS])
        for _ in range(120):
            s.world_step(params)
            world_chain.oracle_world_step(params, ref)
            world_chain.oracle_world_step(params, rev, reverse=True)
        out = world_chain.copy_world(world)
        res = s.world_download(*[out[k] for k in world_chain.WORLD_KEYS])
    got = dict(zip(world_chain.WORLD_KEYS, res[:6]))
    b, c = got["bodies"], got["contacts"]
    rb, vb = ref["bodies"], rev["bodies"]
    assert np.isfinite(b["position"]).all() and np.isfinite(b["linearVelocity"]).all()
    yard = float(np.abs(vb["position"] - rb["position"]).max())
    dev = float(np.abs(b["position"] - rb["position"]).max())
    assert dev <= max(1.5 * yard, 0.02), "device order deviates %.4g m from the pool order, reversed pool order %.4g m" % (dev, yard)

    def angle_of(x):
        return np.arctan2(x["rot"][:, 0], x["rot"][:, 1])
    assert float(np.abs(angle_of(b) - angle_of(rb)).max()) <= max(1.5 * float(np.abs(angle_of(vb) - angle_of(rb)).max()), 0.01)
    # TGS_Soft and SoftStep (8 sub-steps) have the 40-high pile at rest after two seconds; PGS_Soft (4 iterations, no
    # sub-steps) does not converge on it in either order -- the pool-order chain still moves at 0.75 m/s -- so its
    # bounds are the pool-order chain's own state
    substepping = solver_name in ("TGS_Soft", "SoftStep")
    speed = float(np.abs(b["linearVelocity"]).max())
    assert speed < (0.01 if substepping else float(np.abs(rb["linearVelocity"]).max()) + 0.01), speed
    dynamic = b["type"] == wire.BODY_DYNAMIC
    weight_impulse = float(b["mass"][dynamic].sum()) * 10.0 / 60.0 / (vel if substepping else 1)
    ground = np.flatnonzero(b["type"] == wire.BODY_STATIC)
    on_ground = (np.isin(c["bodyA"], ground) | np.isin(c["bodyB"], ground)) & (c["pointCount"] > 0)
    carried = sum(float(c["points"][k][j]["normalImpulse"]) for k in np.flatnonzero(on_ground) for j in range(c["pointCount"][k]))
    assert abs(carried / weight_impulse - 1.0) < (0.01 if substepping else 0.15), (carried, weight_impulse)
    top = int(np.argmax(world["bodies"]["position"][:, 1]))
    sink = float(world["bodies"]["position"][top, 1] - b["position"][top, 1])
    assert 0.0 <= sink < (0.15 if substepping else 0.35), sink  # soft contacts: 3 mm of compression per layer
    assert (got["pairs"]["shapeA"] >= 0).sum() == (ref["pairs"]["shapeA"] >= 0).sum()


@pytest.mark.parametrize("seed,solver_name", [(1, "SoftStep"), (9, "TGS_Soft"), (17, "PGS_Soft"), (15, "TGS_Soft")])
def test_rain_world_loop_through_the_strip_paths(seed, solver_name):
    """The rain worlds again, larger (300-900 bodies) and with the strip options forced and drawn at random, so that the
    pile of mixed shapes -- one- and two-point manifolds, a kinematic paddle that several strips touch -- is swept by the
    persistent strip kernel while contacts come and go: 120 steps of the whole loop, bit-exact against the oracle chain
    (a one-off run of 160 seeds found no difference)."""
    from tests import common, oraclebind
    rng = np.random.default_rng(5000 + seed)
    vel, pos = common.DEFAULT_ITERS[solver_name]
    params = wire.StepParams.make(solver_name, 1.0 / 60.0, vel, pos, True)
    world = world_chain.rain_world(seed, int(rng.integers(300, 900)), spin=bool(seed % 2))
    ref = world_chain.copy_world(world)
    persistent = 0
    with hip.Solver(0) as s:
        s.set_option("strip_patience", int(rng.integers(0, 3)))
        s.set_option("strip_min_bodies", 0)
        s.set_option("strip_bodies", int(rng.integers(30, 160)))
        s.set_option("max_group_bodies", int(rng.choice([32, 64, 128])))
        s.world_upload(*[world[k] for k in world_chain.WORLD_KEYS])
        for step in range(120):
            if world_chain.moved_any(ref):
                got = s.world_find_pairs()
                want = world_chain.oracle_find_pairs(ref)
                assert np.array_equal(got, want), "step %d: new pairs" % step
                if len(got):
                    slots, contacts, pairs = _create_contacts(ref, got)
                    s.world_set_contacts(slots, contacts, pairs)
            info = s.world_step(params)
            order, _ = s.contact_order()
            status = world_chain.oracle_world_step(params, ref, contact_order=order)
            assert info["separatedCount"] == int((status == wire.PAIR_SEPARATED).sum()), "step %d" % step
            persistent += s.stats()["persistent"]
            if step % 5 == 4:
                out = world_chain.copy_world(world)
                res = s.world_download(*[out[k] for k in world_chain.WORLD_KEYS])
                world_chain.assert_device_equals_oracle(dict(zip(world_chain.WORLD_KEYS, res[:6])), ref,
                                                        "rain-strips %d %s step %d" % (seed, solver_name, step))
    assert persistent > 20, persistent  # (of 120 steps: the pile has to form first, and every created contact rebuilds the structure)


def _rain_loop_on_the_interpreters_strips(seed, solver_name):
    """120 steps of the whole loop under a solver the op interpreter sweeps, strips forced: bit-exact against the oracle chain swept in the
    reported order; returns (steps on the persistent launch, steps that placed a contact into a running strip structure)."""
    from tests import common, oraclebind
    rng = np.random.default_rng(7000 + seed)
    vel, pos = common.DEFAULT_ITERS[solver_name]
    params = wire.StepParams.make(solver_name, 1.0 / 60.0, vel, pos, True)
    world = world_chain.rain_world(seed, int(rng.integers(300, 900)), spin=bool(seed % 2))
    ref = world_chain.copy_world(world)
    persistent = placed_while_persistent = 0
    last = None
    with hip.Solver(0) as s:
        s.set_option("strip_patience", 0)
        s.set_option("strip_min_bodies", 0)
        s.set_option("strip_bodies", int(rng.integers(30, 160)))
        s.set_option("max_group_bodies", int(rng.choice([32, 64, 128])))
        s.world_upload(*[world[k] for k in world_chain.WORLD_KEYS])
        for step in range(120):
            if world_chain.moved_any(ref):
                got = s.world_find_pairs()
                want = world_chain.oracle_find_pairs(ref)
                assert np.array_equal(got, want), "step %d: new pairs" % step
                if len(got):
                    slots, contacts, pairs = _create_contacts(ref, got)
                    s.world_set_contacts(slots, contacts, pairs)
            info = s.world_step(params)
            order, _ = s.contact_order()
            jorder, _ = s.joint_order()
            status = world_chain.oracle_world_step(params, ref, contact_order=order, joint_order=jorder)
            assert info["separatedCount"] == int((status == wire.PAIR_SEPARATED).sum()), "step %d" % step
            st = s.stats()
            persistent += st["persistent"]
            if last is not None and st["persistent"] and st["structureBuilds"] == last["structureBuilds"] and st["placedContacts"] > last["placedContacts"]:
                placed_while_persistent += 1
            last = st
            if step % 5 == 4:
                out = world_chain.copy_world(world)
                res = s.world_download(*[out[k] for k in world_chain.WORLD_KEYS])
                world_chain.assert_device_equals_oracle(dict(zip(world_chain.WORLD_KEYS, res[:6])), ref,
                                                        "rain-interpreter %d %s step %d" % (seed, solver_name, step))
    return persistent, placed_while_persistent


@pytest.mark.parametrize("seed,solver_name", [(3, "PGS_NGS_Block"), (5, "XPBD"), (7, "TGS_Sticky"), (13, "TGS_NGS")])
def test_rain_world_loop_on_the_op_interpreters_strips(seed, solver_name):
    """... and under the solvers the op interpreter sweeps (generic_kernel.hip): their strips take a created contact only where a round the
    build laid out has a free position between two bodies the strip or seam already lists (IncrementalStrips::takeOnly, r6) -- every
    other contact still builds.  Every world bit-exact against the oracle chain; of three worlds per solver at least one must have run
    the persistent launch with contacts placed since the last build (which world does depends on every structure policy there is)."""
    persistent = exercised = 0
    for k in range(3):
        p, e = _rain_loop_on_the_interpreters_strips(seed + 100 * k, solver_name)
        persistent += p
        exercised += e
        if exercised and persistent > 5:
            break
    assert persistent > 5 and exercised > 0, (persistent, exercised)


def test_world_download_boxes_equals_the_shape_records():
    """s2amd_world_download_boxes: the 36-byte {aabb, fatAABB, enlarged} of every shape slot == the same fields of the full
    shape records s2amd_world_download returns (what the reference-side binding reads after every step)."""
    params = wire.StepParams.make("TGS_Soft", 1.0 / 60.0, 8, 4, True)
    world = world_chain.rain_world(3, 120)
    with hip.Solver(0) as s:
        s.world_upload(*[world[k] for k in world_chain.WORLD_KEYS])
        for _ in range(5):
            s.world_step(params)
        out = world_chain.copy_world(world)
        res = s.world_download(*[out[k] for k in world_chain.WORLD_KEYS])
        shapes = res[3]
        boxes = s.world_download_boxes(len(shapes))
    assert np.array_equal(boxes["aabb"].view(np.uint32), np.ascontiguousarray(shapes["aabb"]).view(np.uint32))
    assert np.array_equal(boxes["fatAABB"].view(np.uint32), np.ascontiguousarray(shapes["fatAABB"]).view(np.uint32))
    assert np.array_equal(boxes["enlarged"] != 0, shapes["enlarged"] != 0) and (boxes["enlarged"] != 0).any()
