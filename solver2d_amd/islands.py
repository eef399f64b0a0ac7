"""Simulation islands and their sharding across GPUs (host side, numpy/scipy).

The reference has no islands (`islandPool` is dead code), so this module DEFINES
them: an island is a connected component of the graph whose nodes are the movable bodies
(invMass != 0 or invI != 0) and whose edges are the active contact constraints (pointCount > 0)
and live joints between two movable bodies.  Immovable bodies (static, kinematic, massless) never
connect islands -- no impulse propagates through them -- and are replicated into every shard that
touches them.

Because islands share no movable body, solving them separately is arithmetic-identical to solving
them together as long as each shard keeps the pool order of its own constraints; that is what
`extract` guarantees and `tests/test_islands.py` / `tests/test_islands_dist.py` check bit for bit.
One process per GPU solves its shard with no intra-step communication; the only exchange is the
per-step gather of body state (and impulses if the caller wants them back).
"""
import numpy as np
from scipy.sparse import coo_matrix
from scipy.sparse.csgraph import connected_components

from . import wire


def movable_mask(bodies):
    return (bodies["type"] != wire.BODY_FREE) & ((bodies["invMass"] != 0) | (bodies["invI"] != 0))


def find_islands(bodies, contacts, joints):
    """Returns (island_of_body, island_count).  island_of_body is -1 for static/free bodies; island
    ids are numbered by their lowest body index, so the labelling is deterministic.  Kinematic and
    massless bodies never connect anything but still have to be integrated by exactly one owner, so
    each forms an island of its own."""
    nb = len(bodies)
    mov = movable_mask(bodies)
    owned = (bodies["type"] != wire.BODY_FREE) & (bodies["type"] != wire.BODY_STATIC)
    ea, eb = [], []
    act = contacts["pointCount"] > 0
    if act.any():
        a, b = contacts["bodyA"][act], contacts["bodyB"][act]
        both = mov[a] & mov[b]
        ea.append(a[both])
        eb.append(b[both])
    live = joints["type"] == wire.JOINT_REVOLUTE
    if live.any():
        a, b = joints["bodyA"][live], joints["bodyB"][live]
        both = mov[a] & mov[b]
        ea.append(a[both])
        eb.append(b[both])
    if ea:
        ea, eb = np.concatenate(ea), np.concatenate(eb)
    else:
        ea = eb = np.zeros(0, dtype=np.int64)
    g = coo_matrix((np.ones(len(ea), dtype=np.int8), (ea, eb)), shape=(nb, nb))
    _n, label = connected_components(g, directed=False)
    label = label.astype(np.int64)
    # renumber by first body index
    island = np.full(nb, -1, dtype=np.int32)
    if not owned.any():
        return island, 0
    uniq, first = np.unique(label[owned], return_index=True)
    order = np.argsort(np.flatnonzero(owned)[first], kind="stable")
    remap = np.empty(len(uniq), dtype=np.int32)
    remap[order] = np.arange(len(uniq), dtype=np.int32)
    island[owned] = remap[np.searchsorted(uniq, label[owned])]
    return island, len(uniq)


def constraint_islands(bodies, contacts, joints, island):
    """Island of every contact / joint (-1 when inactive or attached to immovable bodies only)."""
    mov = movable_mask(bodies)
    ci = np.full(len(contacts), -1, dtype=np.int32)
    act = contacts["pointCount"] > 0
    ba, bb = contacts["bodyA"][act], contacts["bodyB"][act]
    a = np.where(mov[ba], island[ba], -1)
    b = np.where(mov[bb], island[bb], -1)
    fallback = np.where(island[ba] >= 0, island[ba], island[bb])  # e.g. kinematic vs static
    ci[act] = np.where(a >= 0, a, np.where(b >= 0, b, fallback))
    ji = np.full(len(joints), -1, dtype=np.int32)
    live = joints["type"] >= 0
    if live.any():
        jb = joints["bodyB"][live]
        ja = np.maximum(joints["bodyA"][live], 0)
        rev = joints["type"][live] == wire.JOINT_REVOLUTE
        b = np.where(mov[jb], island[jb], -1)
        a = np.where(rev & mov[ja], island[ja], -1)
        fallback = np.where(island[jb] >= 0, island[jb], np.where(rev, island[ja], -1))
        ji[live] = np.where(a >= 0, a, np.where(b >= 0, b, fallback))
    return ci, ji


def partition(weights, n_shards):
    """Longest-processing-time bin packing of islands by weight; deterministic.
    Returns shard index per island."""
    weights = np.asarray(weights, dtype=np.int64)
    order = np.lexsort((np.arange(len(weights)), -weights))
    load = np.zeros(n_shards, dtype=np.int64)
    shard = np.zeros(len(weights), dtype=np.int32)
    for i in order:
        s = int(np.argmin(load))
        shard[i] = s
        load[s] += max(int(weights[i]), 1)
    return shard


class Shard:
    """One rank's sub-world: wire arrays + the maps back into the full world."""

    def __init__(self, bodies, contacts, joints, body_ids, contact_ids, joint_ids, owned_body):
        self.bodies, self.contacts, self.joints = bodies, contacts, joints
        self.body_ids, self.contact_ids, self.joint_ids = body_ids, contact_ids, joint_ids
        self.owned_body = owned_body  # bool per shard body: movable body owned by this shard


# sticky_partition moves islands again once the heaviest shard carries more than this multiple of the mean load (weights as in
# shard_world: 2 per contact constraint + 1 per joint): the per-step all-gather record and the step time both follow the LARGEST shard,
# so islands that drift onto a few ranks under sustained churn cost every rank.  A move costs the island's constraint state on the
# wire and an upload, so small imbalances stay: 1.75 is above anything longest-processing-time bin packing itself produces for islands
# of unequal size (its bound is 4/3 of the optimum), above what ONE merge of two equal islands among two per rank leaves behind (1.5 x
# the mean: the created contact moves one island, nothing else), and it is reached on two ranks once one of them holds 7/8 of the world.
REBALANCE_THRESHOLD = 1.75


def shard_world(bodies, contacts, joints, n_shards, previous_owner=None, finder=None):
    """Split a world into n_shards sub-worlds along island boundaries.

    finder: callable (bodies, contacts, joints) -> (island_of_body, island_count) used instead of the host's find_islands -- the
    device's union-find (hip.Solver.find_islands = s2amd_find_islands, structure.hip), whose labelling is the same by definition
    (islands numbered by their lowest body index; tests/test_gpu_structure.py compares them exactly).

    previous_owner (int per body, -1 = nobody): the shard that owned each body under the partition before the graph changed.  An
    island then stays where most of its bodies were (ties: the lowest shard), so that a contact created between two islands moves
    ONE of them -- the smaller -- and every island the change did not touch stays put; without it the islands are bin-packed afresh
    (longest-processing-time by constraint count)."""
    if finder is not None:
        island, n_islands = finder(bodies, contacts, joints)
        island = np.asarray(island, dtype=np.int32)
    else:
        island, n_islands = find_islands(bodies, contacts, joints)
    ci, ji = constraint_islands(bodies, contacts, joints, island)
    weights = np.bincount(ci[ci >= 0], minlength=n_islands) * 2 + np.bincount(ji[ji >= 0], minlength=n_islands)
    if previous_owner is None:
        shard_of_island = partition(weights, n_shards)
    else:
        shard_of_island = sticky_partition(island, n_islands, weights, np.asarray(previous_owner), n_shards)
    return [extract(bodies, contacts, joints, island, ci, ji, shard_of_island, s) for s in range(n_shards)], island, shard_of_island


def sticky_partition(island, n_islands, weights, previous_owner, n_shards, threshold=REBALANCE_THRESHOLD):
    """Shard per island: where most of its bodies were; islands of bodies nobody owned go to the least loaded shard (heaviest first).
    Then, while the heaviest shard carries more than `threshold` x the mean load, its LIGHTEST island that still helps (one whose move
    leaves the receiving shard below the giver) goes to the least loaded shard -- few, small moves, deterministic (ties: the lowest
    island / shard index); a merged island too heavy for any move to help stays (one island never splits)."""
    shard = np.full(n_islands, -1, dtype=np.int32)
    has = (island >= 0) & (previous_owner >= 0)
    if has.any():
        votes = np.zeros((n_islands, n_shards), dtype=np.int64)
        np.add.at(votes, (island[has], previous_owner[has]), 1)
        seen = votes.sum(axis=1) > 0
        shard[seen] = np.argmax(votes[seen], axis=1).astype(np.int32)  # (argmax takes the lowest shard among equals)
    load = np.zeros(n_shards, dtype=np.int64)
    np.add.at(load, shard[shard >= 0], np.maximum(np.asarray(weights, dtype=np.int64)[shard >= 0], 1))
    rest = np.flatnonzero(shard < 0)
    for i in rest[np.lexsort((rest, -np.asarray(weights, dtype=np.int64)[rest]))]:
        s = int(np.argmin(load))
        shard[i] = s
        load[s] += max(int(weights[i]), 1)
    w = np.maximum(np.asarray(weights, dtype=np.int64), 1)
    for _ in range(n_islands):
        mean = load.sum() / max(n_shards, 1)
        heavy, light = int(np.argmax(load)), int(np.argmin(load))
        if n_shards < 2 or mean <= 0 or load[heavy] <= threshold * mean:
            break
        mine = np.flatnonzero(shard == heavy)
        mine = mine[np.lexsort((mine, w[mine]))]  # lightest first
        movable = [int(i) for i in mine if load[light] + w[i] < load[heavy]]
        if not movable:
            break
        i = movable[0]
        shard[i] = light
        load[heavy] -= w[i]
        load[light] += w[i]
    return shard


def extract(bodies, contacts, joints, island, ci, ji, shard_of_island, s):
    csel = np.flatnonzero((ci >= 0) & (shard_of_island[np.maximum(ci, 0)] == s))
    jsel = np.flatnonzero((ji >= 0) & (shard_of_island[np.maximum(ji, 0)] == s))
    own = (island >= 0) & (shard_of_island[np.maximum(island, 0)] == s)
    used = own.copy()
    if len(csel):
        used[contacts["bodyA"][csel]] = True
        used[contacts["bodyB"][csel]] = True
    if len(jsel):
        used[joints["bodyB"][jsel]] = True
        rev = joints["type"][jsel] == wire.JOINT_REVOLUTE
        used[joints["bodyA"][jsel][rev]] = True
    body_ids = np.flatnonzero(used)  # ascending => pool order is preserved inside the shard
    remap = np.full(len(bodies), -1, dtype=np.int32)
    remap[body_ids] = np.arange(len(body_ids), dtype=np.int32)
    sb = bodies[body_ids].copy()
    sc = contacts[csel].copy()
    sj = joints[jsel].copy()
    sc["bodyA"] = remap[sc["bodyA"]]
    sc["bodyB"] = remap[sc["bodyB"]]
    sj["bodyB"] = remap[sj["bodyB"]]
    if len(sj):
        a = sj["bodyA"].copy()
        ok = a >= 0
        a[ok] = remap[a[ok]]
        sj["bodyA"] = a
    return Shard(sb, sc, sj, body_ids, csel, jsel, own[body_ids])


def merge_back(bodies, contacts, joints, shards):
    """Scatter solved shards into the full-world arrays (only what each shard owns)."""
    for sh in shards:
        o = sh.owned_body
        bodies[sh.body_ids[o]] = sh.bodies[o]
        ci_global = contacts["constraintIndex"][sh.contact_ids].copy()
        contacts[sh.contact_ids] = _restore_indices(sh.contacts, contacts[sh.contact_ids])
        contacts["constraintIndex"][sh.contact_ids] = ci_global  # gather index is a whole-world property
        joints[sh.joint_ids] = _restore_joint_indices(sh.joints, joints[sh.joint_ids])
    return bodies, contacts, joints


def _restore_indices(solved, original):
    out = solved.copy()
    out["bodyA"] = original["bodyA"]
    out["bodyB"] = original["bodyB"]
    return out


def _restore_joint_indices(solved, original):
    out = solved.copy()
    out["bodyA"] = original["bodyA"]
    out["bodyB"] = original["bodyB"]
    return out
