"""The reference's broad-phase trees kept on the device (solver2d_amd/csrc/tree_mirror.hip).

Checker: the reference's OWN tree code -- s2DynamicTree_CreateProxy / _EnlargeProxy / _Rebuild / _Query compiled from
/root/reference/src/dynamic_tree.c into oracle/_ref/libs2ref.so -- driven beside the device:
  * after every s2amd_world_step the device's node arrays equal the reference's node for node (ids, links, boxes,
    heights, category bits, flags, and the free list's `next` chain);
  * s2amd_world_find_pairs returns the new pairs in the order the reference's s2FindPairs + LIFO pair lists
    (src/broad_phase.c:253-254, :273-320, :332-357) would create them in, computed here from s2DynamicTree_Query's real
    callback sequence on those trees;
  * through the product binding (shim/s2_amd_binding.c) with S2AMD_CHECK_TREES: round 5's host replay
    (s2amdBinding_OrderPairs on the reference's trees) agrees with the device on every query of whole-world loops."""
import ctypes
import numpy as np
import pytest

from solver2d_amd import hip, wire
from tests import common, refbind, world_chain
from tests.test_gpu_world import _create_contacts

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not refbind.available(), reason="oracle/_ref/libs2ref.so not built")]


def _query(tree, box):
    """[(proxy, userData)] in s2DynamicTree_Query's callback order"""
    from tests import treebind
    seen = []
    CB = ctypes.CFUNCTYPE(ctypes.c_bool, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p)

    def cb(proxy, user, ctx):
        seen.append((proxy, user))
        return True
    L = refbind.lib()
    L.s2DynamicTree_Query.argtypes = [ctypes.POINTER(treebind.DynamicTree), treebind.Box, CB, ctypes.c_void_p]
    L.s2DynamicTree_Query.restype = None
    L.s2DynamicTree_Query(ctypes.byref(tree.t), treebind.box(box), CB(cb), None)
    return seen


def _reference_creation_order(found, shapes, move_buffer, trees):
    """the sequence s2UpdateBroadPhasePairs creates the pairs of `found` in: src/broad_phase.c:273-320 (one query per proxy of the
    move buffer: dynamic, kinematic, static tree for a dynamic proxy; the dynamic tree for a kinematic one), :166-258 (who reports
    a pair; every report goes to the FRONT of the proxy's list), :332-357 (lists walked in move-buffer order)"""
    by_set = {frozenset(p): tuple(p) for p in found.tolist()}
    moved = set(move_buffer)
    involved = set(found.ravel().tolist())
    out = []
    for q in move_buffer:
        if q not in involved:
            continue
        keyq = int(shapes["proxyKey"][q])
        typeq = keyq & 0xF
        fat = shapes["fatAABB"][q]
        mine = []
        for tt in ([2, 1, 0] if typeq == 2 else [2] if typeq == 1 else []):
            for proxy, user in _query(trees[tt], fat):
                key = (proxy << 4) | tt
                if key == keyq or (user in moved and key > keyq):
                    continue
                pair = by_set.get(frozenset((q, user)))
                if pair is not None:
                    mine.insert(0, pair)
        out += mine
    assert len(out) == len(found) and len(set(out)) == len(out)
    return np.array(out, dtype=np.int32).reshape(-1, 2)


def _plant_trees(world):
    """the three reference trees of a synthetic world: one proxy per shape, created in shape order as s2CreateShape would
    (src/shape.c -> s2BroadPhase_CreateProxy); the shapes take the proxy keys the reference's allocator hands out"""
    from tests import treebind
    trees = [treebind.RefTree() for _ in range(3)]
    shapes, bodies = world["shapes"], world["bodies"]
    for si in range(len(shapes)):
        if shapes["type"][si] == wire.SHAPE_FREE:
            continue
        t = int(bodies["type"][shapes["body"][si]])
        proxy = trees[t].create_proxy(shapes["fatAABB"][si], int(shapes["categoryBits"][si]), si)
        shapes["proxyKey"][si] = (proxy << 4) | t
    return trees


def _refit_order(world):
    shapes, bodies = world["shapes"], world["bodies"]
    return np.array([si for si in range(len(shapes)) if shapes["type"][si] != wire.SHAPE_FREE and bodies["type"][shapes["body"][si]] != wire.BODY_STATIC],
                    dtype=np.int32)


@pytest.mark.parametrize("seed,solver_name,count,beside", [(1, "TGS_Soft", 110, 1), (2, "PGS_NGS_Block", 160, 1), (3, "Jacobi", 90, 0), (4, "SoftStep", 400, 0),
                                                           (5, "PGS_Soft", 2500, 1), (6, "TGS_Soft", 5000, 1), (7, "SoftStep", 1500, 1)])
def test_device_trees_follow_the_reference(seed, solver_name, count, beside):
    """2,500 and 5,000 bodies: segments longer than a workgroup, split in global memory before their parts are finished in LDS;
    beside: the rebuild on a stream of its own beside stage 3 and the solve (the default) or on the step's stream"""
    from tests import treebind
    vel, pos = common.DEFAULT_ITERS[solver_name]
    params = wire.StepParams.make(solver_name, 1.0 / 60.0, vel, pos, True)
    world = world_chain.rain_world(seed, count)
    trees = _plant_trees(world)
    ref = world_chain.copy_world(world)
    order = _refit_order(world)
    created = rebuilt = 0
    try:
        with hip.Solver(0) as s:
            s.set_option("tree_stream", beside)
            s.world_upload(*[world[k] for k in world_chain.WORLD_KEYS])
            s.world_set_refit_order(order)
            for t in range(3):
                s.world_set_tree(t, trees[t].nodes(), trees[t].root)
            # (after their creation every proxy of a movable body is in the move buffer, in creation order; the static ones were
            # never buffered, src/broad_phase.c:94-104: rain_world flags them all, the query ignores static askers)
            move_buffer = [int(si) for si in order]
            for step in range(60 if count < 1000 else 25):
                if world_chain.moved_any(ref):
                    got = s.world_find_pairs()
                    want_set = world_chain.oracle_find_pairs(ref)
                    assert sorted(map(tuple, got.tolist())) == sorted(map(tuple, want_set.tolist())), "step %d: the pair set" % step
                    want = _reference_creation_order(got, ref["shapes"], move_buffer, trees)
                    assert np.array_equal(got, want), "step %d: creation order of %d pairs" % (step, len(got))
                    if len(got):
                        created += len(got)
                        slots, contacts, pairs = _create_contacts(ref, got)
                        s.world_set_contacts(slots, contacts, pairs)
                # stage 2 (src/world.c:130), the step, stage 4's enlarges (world.c:283-290) in refit order
                for t in (1, 2):
                    trees[t].rebuild()
                s.world_step(params)
                co, _ = s.contact_order()
                world_chain.oracle_world_step(params, ref, contact_order=co)
                move_buffer = [int(si) for si in order if ref["shapes"]["enlarged"][si] != 0]
                for si in move_buffer:
                    key = int(ref["shapes"]["proxyKey"][si])
                    trees[key & 0xF].enlarge(key >> 4, ref["shapes"]["fatAABB"][si])
                rebuilt += 1 if len(move_buffer) > 1 else 0
                for t in (1, 2):
                    nodes, root = s.world_get_tree(t, trees[t].t.nodeCapacity)
                    assert root == trees[t].root, "step %d tree %d: root" % (step, t)
                    treebind.same_nodes(trees[t].nodes().view(wire.tree_node_dtype), nodes, "step %d tree %d" % (step, t))
    finally:
        for t in trees:
            t.close()
    assert created > 100 and rebuilt > 15, (created, rebuilt)


def test_set_tree_round_trip_and_refusals():
    from tests import treebind
    world = world_chain.rain_world(9, 60)
    trees = _plant_trees(world)
    try:
        with hip.Solver(0) as s:
            with pytest.raises(hip.S2AmdError):
                s.world_set_tree(2, trees[2].nodes(), trees[2].root)  # no resident world yet
            s.world_upload(*[world[k] for k in world_chain.WORLD_KEYS])
            with pytest.raises(hip.S2AmdError):
                s.world_get_tree(2, trees[2].t.nodeCapacity)  # nothing sent yet
            for t in range(3):
                s.world_set_tree(t, trees[t].nodes(), trees[t].root)
                nodes, root = s.world_get_tree(t, trees[t].t.nodeCapacity)
                assert root == trees[t].root and nodes.tobytes() == trees[t].nodes().tobytes()
            # a tree with a flagged internal node is not what stage 2 leaves behind: refused (the caller keeps its host replay)
            key = int(world["shapes"]["proxyKey"][10])
            trees[2].enlarge(key >> 4, world["shapes"]["fatAABB"][10] + np.float32(3.0))
            with pytest.raises(hip.S2AmdError):
                s.world_set_tree(2, trees[2].nodes(), trees[2].root)
            bad = trees[0].nodes()
            with pytest.raises(hip.S2AmdError):
                s.world_set_tree(0, bad, len(bad))  # root out of range
            # a new upload forgets the trees: the query is sorted by (A, B) again
            s.world_upload(*[world[k] for k in world_chain.WORLD_KEYS])
            with pytest.raises(hip.S2AmdError):
                s.world_get_tree(2, trees[2].t.nodeCapacity)
            got = s.world_find_pairs()
            assert np.array_equal(got, np.array(sorted(map(tuple, got.tolist())), dtype=np.int32).reshape(-1, 2))
    finally:
        for t in trees:
            t.close()


BINDING_CASES = [("mixed", 24, "PGS_NGS_Block", 120), ("tumbler", 150, "SoftStep", 120), ("shapes_zoo", 40, "TGS_Sticky", 150),
                 ("far_ragdoll_pile", 0, "PGS", 100), ("card_house", 0, "XPBD", 60), ("circle_pile", 16, "Jacobi", 80), ("pyramid", 40, "PGS_NGS_Block", 60)]


@pytest.mark.parametrize("scene,p0,solver_name,steps", BINDING_CASES)
def test_binding_host_replay_agrees_with_the_device_trees(scene, p0, solver_name, steps):
    """S2AMD_CHECK_TREES through the product binding: the device orders the pairs and keeps the trees; the binding ALSO replays every
    step into the reference's own trees (round 5's route) and compares -- the order of every query with more than one pair
    against s2amdBinding_OrderPairs, the trees node for node at every query that found a pair."""
    vel, pos = common.DEFAULT_ITERS[solver_name]
    L = refbind.lib()
    L.s2ref_use_amd_world.argtypes = [ctypes.c_char_p, ctypes.c_int]
    L.s2ref_use_amd_world.restype = ctypes.c_int
    out = (ctypes.c_long * 4)()
    with refbind.RefWorld(scene, solver_name, p0, 0) as w:
        assert L.s2ref_use_amd_world(hip.LIB_PATH.encode(), 0) == 0
        L.s2ref_world_device_pairs(1)
        L.s2amdBinding_DeviceTrees(1, 1)
        L.s2amdBinding_TreeCheck(out)
        try:
            for _ in range(steps):
                w.step(1.0 / 60.0, vel, pos, True)
            assert L.s2ref_replace_error() == 0
            L.s2amdBinding_TreeCheck(out)
        finally:
            L.s2amdBinding_DeviceTrees(1, 0)
            L.s2ref_world_device_pairs(0)
            assert L.s2ref_use_amd_world(None, 0) == 0
    queries, order_differs, tree_checks, tree_differs = list(out)
    assert tree_checks > 0 or scene == "pyramid", "no query of this loop found a pair: nothing was compared"
    assert order_differs == 0 and tree_differs == 0, "%d of %d queries ordered differently, %d of %d trees differ" % (order_differs, queries, tree_differs, tree_checks)


@pytest.mark.parametrize("scene,p0,steps", [("tumbler", 150, 150), ("shapes_zoo", 40, 200), ("circle_pile", 20, 120), ("confined", 25, 40), ("rush", 150, 80),
                                            ("pyramid", 20, 60), ("card_house", 0, 60), ("warm_start_energy", 0, 140), ("friction_ramp", 0, 150)])
def test_device_ranked_pool_slots_equal_the_references_own_creation_sequence(scene, p0, steps, monkeypatch):
    """The reference ALONE (its trees, its s2UpdateBroadPhasePairs, its s2CreateContact sequence, its CPU solver) beside the product
    binding with everything on the device (pairs found, ranked by the device's trees, created in that order).  Under s2Solve_Jacobi the
    contact pass does not depend on the sweep order (every body adds its constraints' deltas in pool order), so on worlds whose joints
    do not share bodies the two are the same computation: after EVERY step the contact pools hold the same pairs in the same slots,
    and the bodies the same bits.  (Option `incremental` 0: a created contact placed into the existing structure goes to the END of
    its bodies' lists -- the sum order the device reports and the oracle follows, not the pool's.)"""
    monkeypatch.setenv("S2AMD_OPTIONS", "incremental=0")
    L = refbind.lib()
    L.s2ref_use_amd_world.argtypes = [ctypes.c_char_p, ctypes.c_int]
    L.s2ref_use_amd_world.restype = ctypes.c_int
    with refbind.RefWorld(scene, "Jacobi", p0, 0) as ref, refbind.RefWorld(scene, "Jacobi", p0, 0) as dev:
        created = 0
        assert L.s2ref_use_amd_world(hip.LIB_PATH.encode(), 0) == 0
        L.s2ref_world_device_pairs(1)
        L.s2amdBinding_DeviceTrees(1, 0)
        L.s2ref_plain_world(ref.id.index)  # (this one stays on the reference's own code)
        try:
            for step in range(steps):
                ref.step(1.0 / 60.0, 4, 2, True)
                dev.step(1.0 / 60.0, 4, 2, True)
                assert L.s2ref_replace_error() == 0
                if step % 5 == 4 or step < 3:
                    pa_r, pb_r = ref.contact_pairs()
                    pa_d, pb_d = dev.contact_pairs()
                    assert np.array_equal(pa_r, pa_d) and np.array_equal(pb_r, pb_d), "step %d: %d slots differ" % (step, int((pa_r != pa_d).sum() + (pb_r != pb_d).sum()))
                    created = max(created, int((pa_r >= 0).sum()))
            br, cr, jr = ref.pack()
            bd, cd, jd = dev.pack()
        finally:
            L.s2ref_plain_world(-1)
            L.s2ref_world_device_pairs(0)
            assert L.s2ref_use_amd_world(None, 0) == 0
    assert created > 0
    live = br["type"] >= 0
    assert np.array_equal(br["position"][live].view(np.uint32), bd["position"][live].view(np.uint32)), "%s: the two worlds drifted apart" % scene
    assert np.array_equal(br["rot"][live].view(np.uint32), bd["rot"][live].view(np.uint32))
