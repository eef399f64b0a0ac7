"""Island finder and sharding: integer results are exact, and solving shards separately is
bit-identical to solving the whole world (oracle on CPU here; the GPU variant is in
tests/test_gpu_islands.py)."""
import numpy as np
import pytest

from solver2d_amd import islands, synthetic, wire
from tests import common, oraclebind, refbind


def test_island_membership_pyramids():
    b, c, j = synthetic.pyramid(6, count=5)
    isl, n = islands.find_islands(b, c, j)
    assert n == 5
    per = 6 * 7 // 2 + 1
    for k in range(5):
        seg = isl[k * per:(k + 1) * per]
        assert seg[0] == -1                      # the static ground belongs to no island
        assert (seg[1:] == k).all()              # numbered by lowest body index
    ci, ji = islands.constraint_islands(b, c, j, isl)
    assert (np.bincount(ci) == len(c) // 5).all()


def test_partition_is_balanced_and_deterministic():
    w = [10] * 64
    s = islands.partition(w, 8)
    assert (np.bincount(s) == 8).all()
    assert (islands.partition(w, 8) == s).all()
    s2 = islands.partition([100, 1, 1, 1, 50, 50], 2)
    loads = [sum(x for x, k in zip([100, 1, 1, 1, 50, 50], s2) if k == r) for r in (0, 1)]
    assert abs(loads[0] - loads[1]) <= 3


@pytest.mark.parametrize("solver", ["TGS_Soft", "PGS_NGS_Block", "XPBD", "Jacobi"])
def test_sharded_solve_equals_whole_world(solver):
    world = synthetic.pyramid(7, count=6)
    vel, pos = common.DEFAULT_ITERS[solver]
    params = wire.StepParams.make(solver, 1.0 / 60.0, vel, pos, True)
    whole = common.copy3(world)
    sharded = common.copy3(world)
    for _ in range(3):
        oraclebind.solve(params, *whole)
        shards, _isl, _sh = islands.shard_world(*sharded, n_shards=4)
        for sh in shards:
            oraclebind.solve(params, sh.bodies, sh.contacts, sh.joints)
        # constraintIndex of the whole world = pool-order gather index
        act = sharded[1]["pointCount"] > 0
        sharded[1]["constraintIndex"] = -1
        sharded[1]["constraintIndex"][act] = np.arange(int(act.sum()), dtype=np.int32)
        islands.merge_back(*sharded, shards)
        if solver == "PGS_NGS_Block":
            sharded[1]["constraintIndex"] = whole[1]["constraintIndex"]
        common.compare_exact(sharded, whole, solver)


@pytest.mark.skipif(not refbind.available(), reason="oracle/_ref/libs2ref.so not built")
def test_islands_on_reference_scene_with_joints():
    with refbind.RefWorld("mixed", "TGS_Soft", 24, 0) as w:
        for _ in range(60):
            w.step(1.0 / 60.0, 8, 4, True)
        params, pre, post = w.step_captured(1.0 / 60.0, 8, 4, True)
    isl, n = islands.find_islands(*pre)
    assert n >= 2
    # every constraint joins bodies of one island
    c = pre[1]
    act = c["pointCount"] > 0
    mov = islands.movable_mask(pre[0])
    ia, ib = isl[c["bodyA"][act]], isl[c["bodyB"][act]]
    both = mov[c["bodyA"][act]] & mov[c["bodyB"][act]]   # kinematic bodies are islands of their own
    assert (ia[both] == ib[both]).all()
    shards, _, _ = islands.shard_world(*common.copy3(pre), n_shards=3)
    out = common.copy3(pre)
    for sh in shards:
        oraclebind.solve(params, sh.bodies, sh.contacts, sh.joints)
    islands.merge_back(*out, shards)
    out[1]["constraintIndex"] = post[1]["constraintIndex"]
    # static bodies are owned by no shard and keep their input state (the reference leaves them
    # untouched too); the kinematic platform is an island of its own and is integrated by its owner
    common.compare_exact(out, post, "sharded mixed scene vs reference")


def test_sticky_partition_moves_only_the_smaller_part_of_a_merged_island():
    """islands.sticky_partition (re-sharding): an island goes where most of its bodies were (ties: the lowest shard), an
    island nobody owned goes to the least loaded shard, heaviest first -- integer work, checked exactly."""
    from solver2d_amd import islands as isl
    # five bodies of shard 0 and three of shard 1 are ONE island now (0); island 1 was all on shard 1; island 2 is new; island 3 a tie
    island = np.array([0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 2, 2, 2, 3, 3, -1], dtype=np.int32)
    previous = np.array([0, 0, 0, 0, 0, 1, 1, 1, 1, 1, -1, -1, -1, 0, 1, -1], dtype=np.int32)
    weights = np.array([16, 4, 6, 2])
    shard = isl.sticky_partition(island, 4, weights, previous, 2)
    assert shard.tolist() == [0, 1, 1, 0]  # (island 2: shard 1 carries 4 against shard 0's 16 + 2)
    # nobody owned anything: the plain longest-processing-time packing
    fresh = isl.sticky_partition(island, 4, weights, np.full(len(island), -1, dtype=np.int32), 2)
    assert fresh.tolist() == isl.partition(weights, 2).tolist()


def test_sticky_partition_rebalances_when_islands_have_drifted_onto_one_shard():
    """merged islands always go to the rank that held most of their bodies, so under sustained churn the load drifts.  Past
    REBALANCE_THRESHOLD (heaviest shard / mean) the heaviest shard's lightest islands move to the least loaded shard until it is met."""
    from solver2d_amd import islands as isl
    # eight islands of one body each, all of them previously on shard 0 of 4
    island = np.arange(8, dtype=np.int32)
    previous = np.zeros(8, dtype=np.int32)
    weights = np.array([10, 10, 10, 10, 10, 10, 10, 10])
    shard = isl.sticky_partition(island, 8, weights, previous, 4)
    load = np.bincount(shard, weights=weights, minlength=4)
    assert load.max() <= isl.REBALANCE_THRESHOLD * load.mean()
    assert (shard == 0).sum() >= 2  # ... and no more moves than that takes: shard 0 keeps what it can
    # lightest first, lowest index among equals, to the least loaded shard (lowest index among equals)
    assert shard.tolist()[:3] == [1, 2, 3]
    # below the threshold nothing moves (the case of the test above: 18 against a mean of 14)
    keep = isl.sticky_partition(np.array([0, 1], dtype=np.int32), 2, np.array([18, 10]), np.array([0, 1], dtype=np.int32), 2)
    assert keep.tolist() == [0, 1]
    # one island too heavy for any move to help stays where it is
    heavy = isl.sticky_partition(np.array([0, 1], dtype=np.int32), 2, np.array([100, 1]), np.array([0, 0], dtype=np.int32), 2)
    assert heavy.tolist() == [0, 1]


def test_shard_world_with_previous_owner_keeps_untouched_islands_in_place():
    from solver2d_amd import islands as isl
    world = synthetic.pyramid(5, count=6)
    shards, island, first = isl.shard_world(*world, 3)
    owner = np.full(len(world[0]), -1, dtype=np.int32)
    for r, sh in enumerate(shards):
        owner[sh.body_ids[sh.owned_body]] = r
    # a contact between the top boxes of two islands that live on different shards
    a_island, b_island = 0, next(i for i in range(len(first)) if first[i] != first[0])
    a = int(np.flatnonzero(island == a_island)[-1])
    b = int(np.flatnonzero(island == b_island)[-1])
    c = world[1].copy()
    extra = c[:1].copy()
    extra["bodyA"], extra["bodyB"] = a, b
    joined = np.concatenate([c, extra])
    shards2, island2, second = isl.shard_world(world[0], joined, world[2], 3, previous_owner=owner)
    owner2 = np.full(len(world[0]), -1, dtype=np.int32)
    for r, sh in enumerate(shards2):
        owner2[sh.body_ids[sh.owned_body]] = r
    moved = np.flatnonzero((owner >= 0) & (owner != owner2))
    assert len(moved) > 0 and set(island[moved].tolist()) <= {a_island, b_island}  # only bodies of the two merged islands moved ...
    assert len(set(island[moved].tolist())) == 1  # ... and only ONE of the two
    assert owner2[a] == owner2[b]
    # every constraint lives on exactly one shard, pool order preserved
    ids = np.sort(np.concatenate([sh.contact_ids for sh in shards2]))
    assert np.array_equal(ids, np.flatnonzero(joined["pointCount"] > 0))
