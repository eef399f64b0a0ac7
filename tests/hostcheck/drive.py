"""TEST INFRASTRUCTURE.  Drives the HOST side of the library (tests/hostcheck/_build/libs2amd_hostcheck.so: every translation
unit compiled --cuda-host-only with ASan + UBSan, linked against hip_stub.cpp; kernels never run) through the paths whose
index arithmetic the parity tests only see from the outside: structure builds of random worlds under every solver and option
mix, strip / group / resident-island tables, created and destroyed contacts between solves (incremental placement, slack
exhaustion, rebuilds), the order queries, the world chain's bookkeeping.  Run by tests/test_hostcheck.py in a child process
with the ASan runtime preloaded; exits non-zero on the first sanitizer report (-fno-sanitize-recover, ASan's abort)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from solver2d_amd import hip, synthetic, wire  # noqa: E402
from tests import common, fuzz_worlds  # noqa: E402
from tests.test_gpu_incremental import _artificial_contact, _with_spare_slots  # noqa: E402


def fuzz_solves(seeds):
    n = 0
    for seed in range(seeds):
        rng = np.random.default_rng(1000 + seed)
        world = fuzz_worlds.random_world(seed, n_bodies=int(rng.integers(30, 400)), n_contacts=int(rng.integers(40, 900)), n_joints=int(rng.integers(0, 40)))
        with hip.Solver(0) as s:
            s.set_option("strip_patience", 0)
            mode = seed % 4
            if mode == 1:
                s.set_option("groups", 0)
            elif mode == 2:
                s.set_option("max_group_bodies", 48), s.set_option("strip_min_bodies", 0), s.set_option("strip_bodies", 12)
                s.set_option("strips_any_solver", 1)
            elif mode == 3:
                s.set_option("max_group_bodies", 64), s.set_option("island_resident", int(rng.integers(0, 2)))
            for name in wire.SOLVER_NAMES:
                vel, pos = common.DEFAULT_ITERS[name]
                params = wire.StepParams.make(name, 1.0 / 60.0, vel, pos, bool(rng.integers(0, 2)))
                state = common.copy3(world)
                s.solve(params, *state)
                s.contact_order()
                s.joint_order()
                n += 1
    return n


def churn(solver_name, base, steps, options):
    vel, pos = common.DEFAULT_ITERS[solver_name]
    params = wire.StepParams.make(solver_name, 1.0 / 60.0, vel, pos, True)
    rng = np.random.default_rng(7 + base)
    spare_n = 400
    pre = _with_spare_slots(synthetic.pyramid(base), spare_n)
    n0 = len(pre[1]) - spare_n
    pairs = {(min(a, b), max(a, b)) for a, b in zip(pre[1]["bodyA"][:n0].tolist(), pre[1]["bodyB"][:n0].tolist())}
    free_slot = np.zeros(1, dtype=wire.contact_dtype)[0]
    free_slot["bodyA"], free_slot["bodyB"], free_slot["constraintIndex"] = -1, -1, -1
    with hip.Solver(0) as s:
        for k, v in options.items():
            s.set_option(k, v)
        state = common.copy3(pre)
        spare = list(range(n0, n0 + spare_n))
        for step in range(steps):
            for _ in range(int(rng.integers(0, 12))):
                if spare:
                    state[1][spare.pop(0)] = _artificial_contact(rng, state[0], pre[1][5], pairs)
            for _ in range(int(rng.integers(0, 4))):
                victim = int(rng.integers(0, n0))
                if state[1][victim]["bodyA"] >= 0:
                    state[1][victim] = free_slot
                    spare.append(victim)
            # manifolds lose and regain their points
            flip = rng.integers(0, n0, size=8)
            for f in flip:
                if state[1][f]["bodyA"] >= 0:
                    state[1][f]["pointCount"] = int(rng.integers(0, 3))
            s.solve(params, *state)
            s.contact_order()
        return s.stats()["structureBuilds"]


def neighbour_churn(solver_name, base, steps, options):
    """Contacts between boxes two hops apart in the contact graph (what a disturbed pile creates): same or adjacent strips,
    so they can take free positions of the strips' rounds (IncrementalStrips) -- appear without points, gain them, lose them,
    get destroyed, their slots reused."""
    vel, pos = common.DEFAULT_ITERS[solver_name]
    params = wire.StepParams.make(solver_name, 1.0 / 60.0, vel, pos, True)
    rng = np.random.default_rng(11 + base)
    spare_n = 600
    pre = _with_spare_slots(synthetic.pyramid(base), spare_n)
    n0 = len(pre[1]) - spare_n
    a0, b0 = pre[1]["bodyA"][:n0].astype(int), pre[1]["bodyB"][:n0].astype(int)
    pairs = {(min(a, b), max(a, b)) for a, b in zip(a0.tolist(), b0.tolist())}
    nbrs = {}
    for a, b in zip(a0.tolist(), b0.tolist()):
        nbrs.setdefault(a, []).append(b), nbrs.setdefault(b, []).append(a)
    dynamic = np.flatnonzero(pre[0]["type"] == wire.BODY_DYNAMIC)
    free_slot = np.zeros(1, dtype=wire.contact_dtype)[0]
    free_slot["bodyA"], free_slot["bodyB"], free_slot["constraintIndex"] = -1, -1, -1
    with hip.Solver(0) as s:
        for k, v in options.items():
            s.set_option(k, v)
        state = common.copy3(pre)
        s.solve(params, *state)
        spare = list(range(n0, n0 + spare_n))
        mine = []
        for step in range(steps):
            for _ in range(int(rng.integers(0, 8))):
                a = int(rng.choice(dynamic))
                mid = int(rng.choice(nbrs[a]))
                c = int(rng.choice(nbrs[mid]))
                if c == a or (min(a, c), max(a, c)) in pairs or pre[0]["type"][c] != wire.BODY_DYNAMIC or not spare:
                    continue
                pairs.add((min(a, c), max(a, c)))
                slot = spare.pop(0)
                new = _artificial_contact(rng, state[0], pre[1][5], set())
                new["bodyA"], new["bodyB"] = a, c
                new["pointCount"] = 0 if rng.random() < 0.7 else int(rng.integers(1, 3))
                state[1][slot] = new
                mine.append(slot)
            for slot in list(mine):
                r = rng.random()
                if r < 0.25:
                    state[1][slot]["pointCount"] = int(rng.integers(0, 3))
                elif r < 0.30:
                    a, c = int(state[1][slot]["bodyA"]), int(state[1][slot]["bodyB"])
                    pairs.discard((min(a, c), max(a, c)))
                    state[1][slot] = free_slot
                    mine.remove(slot)
                    spare.append(slot)
            s.solve(params, *state)
            s.contact_order()
        st = s.stats()
        return st["structureBuilds"], st["placedContacts"], st["persistent"]


def joining_bodies(solver_name, base, steps, balls):
    """Bodies without constraints that come to touch the pile (IncrementalStrips: a body moves to the strip it first touches, a seam comes
    to carry a body it did not, spare rounds open): free bodies touch random boxes, then boxes next to those, let go, touch elsewhere."""
    vel, pos = common.DEFAULT_ITERS[solver_name]
    params = wire.StepParams.make(solver_name, 1.0 / 60.0, vel, pos, True)
    rng = np.random.default_rng(23 + base)
    b, c, j = common.copy3(synthetic.pyramid(base))
    extra = np.zeros(balls, dtype=wire.body_dtype)
    template_body = b[int(np.flatnonzero(b["invMass"] > 0)[0])]
    for i in range(balls):
        extra[i] = template_body
        extra[i]["position"] = (float(base + 40 + 3 * i), 5.0)
    n0 = len(c)
    spare_n = 8 * balls
    pre = _with_spare_slots((np.concatenate([b, extra]), c, j), spare_n)
    nbrs = {}
    for a, bb in zip(c["bodyA"].astype(int).tolist(), c["bodyB"].astype(int).tolist()):
        nbrs.setdefault(a, []).append(bb), nbrs.setdefault(bb, []).append(a)
    dynamic = np.flatnonzero(b["type"] == wire.BODY_DYNAMIC)
    free_slot = np.zeros(1, dtype=wire.contact_dtype)[0]
    free_slot["bodyA"], free_slot["bodyB"], free_slot["constraintIndex"] = -1, -1, -1
    with hip.Solver(0) as s:
        s.set_option("strip_patience", 0)
        state = common.copy3(pre)
        s.solve(params, *state)
        spare = list(range(n0, n0 + spare_n))
        held = {i: [] for i in range(balls)}  # ball -> [(slot, box)]
        most = {"bodiesAdopted": 0, "seamBodiesAdded": 0, "roundsOpened": 0}
        for step in range(steps):
            for i in range(balls):
                ball = len(b) + i
                r = rng.random()
                if r < 0.5 and spare and len(held[i]) < 5:
                    box = int(rng.choice(dynamic)) if not held[i] else int(rng.choice(nbrs[held[i][-1][1]]))
                    if b["type"][box] != wire.BODY_DYNAMIC or any(box == h[1] for h in held[i]):
                        continue
                    slot = spare.pop(0)
                    new = pre[1][5].copy()
                    new["bodyA"], new["bodyB"] = (ball, box) if rng.random() < 0.5 else (box, ball)
                    new["pointCount"] = int(rng.integers(0, 3))
                    state[1][slot] = new
                    held[i].append((slot, box))
                elif r < 0.6 and held[i]:
                    for slot, _ in held[i]:
                        state[1][slot] = free_slot
                        spare.append(slot)
                    held[i] = []
                elif held[i]:
                    slot = held[i][int(rng.integers(0, len(held[i])))][0]
                    state[1][slot]["pointCount"] = int(rng.integers(0, 3))
            s.solve(params, *state)
            s.contact_order()
            st = s.stats()
            for k in most:  # (the three counters belong to the structure in use: a rebuild starts them again)
                most[k] = max(most[k], st[k])
        return st["structureBuilds"], st["placedContacts"], most["bodiesAdopted"], most["seamBodiesAdded"], most["roundsOpened"]


def world_chain(base, steps):
    world = synthetic.pyramid_world(base)
    keys = ("bodies", "contacts", "joints", "shapes", "pairs", "origins")
    params = wire.StepParams.make("TGS_Soft", 1.0 / 60.0, 8, 4, True)
    with hip.Solver(0) as s:
        s.world_upload(*[world[k] for k in keys])
        for _ in range(steps):
            s.world_step(params)
            s.world_find_pairs()
        s.world_download(*[world[k] for k in keys])


def world_chain_async(base, steps):
    """Structure builds in a worker thread (solver_async.cpp): a resident world whose big island is due for strips gets them from a
    copy of the solver built off the caller's thread and adopted a fixed number of steps later; contacts created and destroyed in
    between are replayed on the copy.  (Kernels never run here: this drives the threads, the copy, the swap and the replay.)"""
    world = synthetic.pyramid_world(base)
    keys = ("bodies", "contacts", "joints", "shapes", "pairs", "origins")
    params = wire.StepParams.make("TGS_Soft", 1.0 / 60.0, 8, 4, True)
    rng = np.random.default_rng(3)
    free = sorted(np.flatnonzero(world["pairs"]["shapeA"] < 0).tolist(), reverse=True)
    live = np.flatnonzero(world["pairs"]["shapeA"] >= 0)
    seen = []
    with hip.Solver(0) as s:
        for k, v in (("strip_min_bodies", 0), ("max_group_bodies", 96), ("strip_bodies", 24), ("strip_patience", 1), ("async_build", 2), ("async_build_delay", 3)):
            s.set_option(k, v)
        s.world_upload(*[world[k] for k in keys])
        mine = []
        for step in range(steps):
            if step % 4 == 1 and free:
                # a pair of boxes two rows apart gets a contact (nothing the narrow phase would find: the stub runs no kernels)
                a = int(rng.choice(live))
                contacts = np.zeros(1, dtype=wire.contact_dtype)
                pairs = np.zeros(1, dtype=wire.pair_state_dtype)
                contacts["bodyA"], contacts["bodyB"] = world["contacts"]["bodyA"][a], world["contacts"]["bodyB"][(a + 7) % len(world["contacts"])]
                if contacts["bodyA"][0] != contacts["bodyB"][0] and contacts["bodyB"][0] >= 0:
                    contacts["friction"], contacts["constraintIndex"] = 0.6, -1
                    pairs["shapeA"], pairs["shapeB"] = contacts["bodyA"][0], contacts["bodyB"][0]  # (one shape per body, same index)
                    slot = free.pop()
                    s.world_set_contacts(np.array([slot], dtype=np.int32), contacts, pairs)
                    mine.append(slot)
            if step % 6 == 5 and mine:
                # ... and one of them is destroyed again by the caller
                slot = mine.pop(0)
                contacts = np.zeros(1, dtype=wire.contact_dtype)
                pairs = np.zeros(1, dtype=wire.pair_state_dtype)
                contacts["bodyA"], contacts["bodyB"], contacts["constraintIndex"] = -1, -1, -1
                pairs["shapeA"], pairs["shapeB"] = -1, -1
                s.world_set_contacts(np.array([slot], dtype=np.int32), contacts, pairs)
                free.append(slot)
            s.world_step(params)
            st = s.stats()
            seen.append((st["stripCount"], st["asyncBuildsRequested"], st["asyncBuildsAdopted"], st["structureBuilds"]))
        s.contact_order()
    return seen


def overflow_chain(base, steps):
    """A contact that fits nowhere in the strips (two boxes many strips apart: a body on two seams is what the partition cannot have)
    takes an overflow position behind the strips instead of a rebuild in that step; the steps run sliced while a worker thread builds
    the structure that holds it, adopted a fixed number of steps later (solver_internal.h: IncrementalStrips; solver_async.cpp).
    Returns per step (overflowContacts, slicedStep, requested, adopted, structureBuilds, kernelLaunches)."""
    world = synthetic.pyramid_world(base)
    keys = ("bodies", "contacts", "joints", "shapes", "pairs", "origins")
    params = wire.StepParams.make("TGS_Soft", 1.0 / 60.0, 8, 4, True)
    # sixteen free pool slots behind the live ones
    spare_c = np.zeros(16, dtype=wire.contact_dtype)
    spare_c["bodyA"], spare_c["bodyB"], spare_c["constraintIndex"] = -1, -1, -1
    spare_p = np.zeros(16, dtype=wire.pair_state_dtype)
    spare_p["shapeA"], spare_p["shapeB"] = -1, -1
    world["contacts"] = np.concatenate([world["contacts"], spare_c])
    world["pairs"] = np.concatenate([world["pairs"], spare_p])
    free = sorted(np.flatnonzero(world["pairs"]["shapeA"] < 0).tolist(), reverse=True)
    assert len(free) == 16
    dynamic = np.flatnonzero(world["bodies"]["type"] == wire.BODY_DYNAMIC)
    seen = []

    def touch(s, a, b, slot, points):
        contacts = np.zeros(1, dtype=wire.contact_dtype)
        pairs = np.zeros(1, dtype=wire.pair_state_dtype)
        contacts["bodyA"], contacts["bodyB"], contacts["friction"], contacts["constraintIndex"] = a, b, 0.6, -1
        contacts["pointCount"] = points
        contacts["normal"] = (0.0, 1.0)
        pairs["shapeA"], pairs["shapeB"] = a, b  # (one shape per body, same index)
        s.world_set_contacts(np.array([slot], dtype=np.int32), contacts, pairs)

    def release(s, slot):
        contacts = np.zeros(1, dtype=wire.contact_dtype)
        pairs = np.zeros(1, dtype=wire.pair_state_dtype)
        contacts["bodyA"], contacts["bodyB"], contacts["constraintIndex"] = -1, -1, -1
        pairs["shapeA"], pairs["shapeB"] = -1, -1
        s.world_set_contacts(np.array([slot], dtype=np.int32), contacts, pairs)

    with hip.Solver(0) as s:
        for k, v in (("strip_patience", 0), ("async_build_delay", 4)):
            s.set_option(k, v)
        s.world_upload(*[world[k] for k in keys])
        mine = []
        for step in range(steps):
            if step in (3, 5, 6) and free:
                owner, _seam, strips = s.strip_owners(len(world["bodies"]))
                assert strips > 4, strips
                # two boxes whose strips are not neighbours
                a = int(dynamic[step])
                far = [int(d) for d in dynamic if owner[d] >= 0 and abs(int(owner[d]) - int(owner[a])) >= 3]
                b = far[len(far) // 2 + step]
                slot = free.pop()
                touch(s, a, b, slot, 2)
                mine.append(slot)
            if step == 8 and mine:
                release(s, mine.pop(0))  # (destroyed while it waits in the overflow region, or just after the adoption)
            s.world_step(params)
            st = s.stats()
            seen.append((st["overflowContacts"], st["slicedStep"], st["asyncBuildsRequested"], st["asyncBuildsAdopted"], st["structureBuilds"], st["kernelLaunches"]))
        # (a build that needs the search over strip widths is not waited for: it falls due a few steps later -- under the sanitizers the
        # worker is slow and these steps are not steps at all, so give it wall-clock time)
        import time
        for _ in range(600):
            if seen[-1][0] == 0:
                break
            time.sleep(0.02)
            s.world_step(params)
            st = s.stats()
            seen.append((st["overflowContacts"], st["slicedStep"], st["asyncBuildsRequested"], st["asyncBuildsAdopted"], st["structureBuilds"], st["kernelLaunches"]))
        order, offsets = s.contact_order()
        assert len(order) == len(set(order.tolist()))
    return seen


def hub_rule():
    """A writable body whose constraints would cost more as colour rounds of a strip than on the tail (S2_COST_*: more than 23) keeps its graph off the strips
    (solver_structure.cpp: cutStrips); the same pile without the hub is cut into strips."""
    bodies, contacts, joints = common.copy3(synthetic.pyramid(36))
    hubbed = contacts.copy()
    dynamic = np.flatnonzero(bodies["type"] == wire.BODY_DYNAMIC)
    hub = int(dynamic[len(dynamic) // 2])
    live = np.flatnonzero((hubbed["bodyA"] >= 0) & (hubbed["bodyA"] != hub) & (hubbed["bodyB"] != hub))
    hubbed["bodyB"][live[:: max(1, len(live) // 60)][:60]] = hub
    counts = []
    for cs in (contacts, hubbed):
        for name in ("TGS_Soft", "PGS_NGS_Block"):
            vel, pos = common.DEFAULT_ITERS[name]
            with hip.Solver(0) as s:
                s.set_option("strip_patience", 0), s.set_option("max_group_bodies", 256), s.set_option("strip_min_bodies", 0), s.set_option("strip_bodies", 60)
                s.solve(wire.StepParams.make(name, 1.0 / 60.0, vel, pos, True), bodies.copy(), cs.copy(), joints.copy())
                s.contact_order()
                counts.append(s.stats()["stripCount"])
    assert counts[0] > 0 and counts[1] > 0 and counts[2] == 0 and counts[3] == 0, counts
    return counts


def hub_tail_churn(solver_name, steps):
    """Contacts created on and destroyed from a hub body (60 boxes lean on one: no strips, its constraints in the sequential tail):
    they take the free positions behind the tail's constraints (IncrementalGlobal::tailFree), boxes the tail does not stage yet join its
    body list; when the tail's slack is used up the structure is built again."""
    vel, pos = common.DEFAULT_ITERS[solver_name]
    params = wire.StepParams.make(solver_name, 1.0 / 60.0, vel, pos, True)
    rng = np.random.default_rng(31)
    bodies, contacts, joints = common.copy3(synthetic.pyramid(36))
    dynamic = np.flatnonzero(bodies["type"] == wire.BODY_DYNAMIC)
    hub = int(dynamic[len(dynamic) // 2])
    live = np.flatnonzero((contacts["bodyA"] >= 0) & (contacts["bodyA"] != hub) & (contacts["bodyB"] != hub))
    contacts["bodyB"][live[:: max(1, len(live) // 60)][:60]] = hub
    spare_n = 160
    pre = _with_spare_slots((bodies, contacts, joints), spare_n)
    n0 = len(pre[1]) - spare_n
    free_slot = np.zeros(1, dtype=wire.contact_dtype)[0]
    free_slot["bodyA"], free_slot["bodyB"], free_slot["constraintIndex"] = -1, -1, -1
    with hip.Solver(0) as s:
        s.set_option("strip_patience", 0), s.set_option("max_group_bodies", 256), s.set_option("strip_min_bodies", 0), s.set_option("strip_bodies", 60)
        state = common.copy3(pre)
        s.solve(params, *state)
        assert s.stats()["stripCount"] == 0
        spare = list(range(n0, n0 + spare_n))
        mine = []
        most_placed = 0
        for step in range(steps):
            for _ in range(int(rng.integers(0, 5))):
                if not spare:
                    break
                box = int(rng.choice(dynamic))
                if box == hub:
                    continue
                slot = spare.pop(0)
                new = pre[1][int(live[3])].copy()
                new["bodyA"], new["bodyB"] = (box, hub) if rng.random() < 0.5 else (hub, box)
                new["pointCount"] = int(rng.integers(0, 3))
                state[1][slot] = new
                mine.append(slot)
            for slot in list(mine):
                r = rng.random()
                if r < 0.2:
                    state[1][slot]["pointCount"] = int(rng.integers(0, 3))
                elif r < 0.3:
                    state[1][slot] = free_slot
                    mine.remove(slot)
                    spare.append(slot)
            s.solve(params, *state)
            s.contact_order()
            most_placed = max(most_placed, s.stats()["placedContacts"])
        return s.stats()["structureBuilds"], most_placed


def main():
    quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
    n = fuzz_solves(6 if quick else 40)
    print("fuzz solves:", n)
    for name, base, opts in (("TGS_Soft", 30, {"groups": 0, "strips": 0}), ("Jacobi", 30, {"groups": 0, "strips": 0}),
                             ("TGS_Soft", 60, {"strip_patience": 2}), ("PGS_NGS_Block", 30, {"groups": 0, "strips": 0, "incremental": 1}),
                             ("SoftStep", 40, {"strip_patience": 0, "max_group_bodies": 256}),
                             # the op interpreter's tables (generic_kernel.hip) under a churning graph, with joints nowhere / strips of two levels
                             ("PGS_NGS_Block", 70, {"strip_patience": 0, "strip_min_bodies": 0}), ("XPBD", 70, {"strip_patience": 1, "strip_min_bodies": 0, "persist_retry": 2})):
        builds = churn(name, base, 6 if quick else 25, opts)
        print("churn %s base %d: %d structure builds" % (name, base, builds))
    for name, base, opts in (("TGS_Soft", 100, {"strip_patience": 0}), ("SoftStep", 100, {"strip_patience": 1}), ("TGS_Soft", 110, {"strip_patience": 0, "wide": 0, "strip_bodies": 160}),
                             ("TGS_Soft", 100, {"strip_patience": 0, "persist_debug": 16}), ("PGS", 100, {"strip_patience": 0})):
        builds, placed, persistent = neighbour_churn(name, base, 10 if quick else 40, opts)
        print("neighbour churn %s base %d: %d structure builds, %d contacts placed, persistent %d" % (name, base, builds, placed, persistent))
    for name in ("TGS_Soft", "SoftStep"):
        got = joining_bodies(name, 100, 12 if quick else 60, 6)
        print("joining bodies %s: %d structure builds, %d contacts placed, at most %d bodies moved to another strip, %d added to a seam, %d spare rounds opened per structure" % ((name,) + got))
        assert got[2] >= 1, got
    print("hub rule: strips", hub_rule())
    for name in ("TGS_Soft", "PGS_NGS_Block"):
        builds, placed = hub_tail_churn(name, 10 if quick else 60)
        print("hub tail churn %s: %d structure builds, %d contacts placed" % (name, builds, placed))
        assert placed >= 3, (builds, placed)
    world_chain(20 if quick else 60, 3 if quick else 8)
    print("world chain ok")
    seen = world_chain_async(30, 24 if quick else 60)
    print("world chain, structure builds in a worker thread: (strips, requested, adopted, builds) per step:", seen[::4])
    assert seen[-1][1] >= 1 and seen[-1][2] >= 1 and seen[-1][0] > 0, seen
    seen = overflow_chain(100, 24)
    print("overflow positions behind the strips: (overflow contacts, sliced, requested, adopted, builds, launches) per step:", seen[:24], "...", seen[-1], "after", len(seen), "steps")
    assert max(x[0] for x in seen) >= 1 and any(x[1] for x in seen), seen  # a contact waited in the overflow region, steps ran sliced
    assert seen[-1][3] >= 1 and seen[-1][0] == 0 and seen[-1][1] == 0, seen  # ... until the worker's structure was adopted
    first = next(i for i, x in enumerate(seen) if x[0] > 0)
    assert seen[first][4] == seen[first - 1][4], seen  # no structure build in the step that found the contact
    print("HOSTCHECK OK")


if __name__ == "__main__":
    main()
