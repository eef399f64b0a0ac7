"""tests/tree_parallel.py (the order-free statement the device executes) against the reference's own s2DynamicTree
compiled here (oracle/_ref): node arrays, roots and free lists equal after every enlarge pass and every rebuild."""
import numpy as np
import pytest

from tests import refbind, tree_parallel as tp

pytestmark = pytest.mark.skipif(not refbind.available(), reason="oracle/_ref/libs2ref.so not built")


def random_boxes(rng, n, extent, size, lattice):
    c = rng.uniform(-extent, extent, size=(n, 2)).astype(np.float32)
    if lattice:  # many equal centre coordinates: the degenerate splits (count / 2) and ties at the pivot
        c = np.round(c / lattice) * lattice
    h = rng.uniform(0.5 * size, size, size=(n, 2)).astype(np.float32)
    return np.concatenate([c - h, c + h], axis=1).astype(np.float32)


def run_world(seed, n, steps, move_fraction, lattice=0.0, churn=False):
    from tests import treebind
    rng = np.random.default_rng(seed)
    ref = treebind.RefTree()
    try:
        boxes = random_boxes(rng, n, 40.0, 1.0, lattice)
        proxies = [ref.create_proxy(b, category=1 << int(rng.integers(0, 4)), user=i) for i, b in enumerate(boxes)]
        ref.rebuild()  # (the first step's stage 2: nothing is flagged -- a tree made by insertions is kept as it is)
        nodes = ref.nodes()
        root = ref.root
        assert tp.flags_are_closed(nodes)
        for step in range(steps):
            # stage 4: some proxies left their fat boxes
            k = max(1, int(move_fraction * n)) if step % 3 != 2 else int(rng.integers(0, 3))
            moved = rng.choice(n, size=min(k, n), replace=False)
            for i in moved:
                boxes[i] += np.tile(rng.normal(0.0, 1.5, size=2).astype(np.float32), 2)
                if lattice:
                    boxes[i] = np.round(boxes[i] / lattice) * lattice
            order = rng.permutation(len(moved))  # (the device's order is not the reference's)
            for i in moved:
                ref.enlarge(proxies[i], boxes[i])
            tp.enlarge(nodes, [proxies[i] for i in moved[order]], [boxes[i] for i in moved[order]])
            treebind.same_nodes(ref.nodes(), nodes, "seed %d step %d after the enlarges" % (seed, step))
            counts = tp.leaf_counts(nodes, root)
            assert counts[root] == n
            # stage 2 of the next step
            ref.rebuild()
            root = tp.rebuild(nodes, root)
            assert root == ref.root, (seed, step)
            treebind.same_nodes(ref.nodes(), nodes, "seed %d step %d after the rebuild" % (seed, step))
            assert not np.any(nodes["enlarged"][nodes["height"] >= 0])
            if churn and step % 4 == 1:
                # a proxy destroyed and one created on the host between steps: the free list hands out what the rebuilds left
                i = int(rng.integers(0, n))
                ref.destroy_proxy(proxies[i])
                boxes[i] = random_boxes(rng, 1, 40.0, 1.0, lattice)[0]
                proxies[i] = ref.create_proxy(boxes[i], user=i)
                nodes, root = ref.nodes(), ref.root
                if not tp.flags_are_closed(nodes):
                    ref.rebuild()
                    nodes, root = ref.nodes(), ref.root
        return ref.free_list
    finally:
        ref.close()


@pytest.mark.parametrize("seed", range(6))
def test_rebuild_matches_reference(seed):
    run_world(seed, n=[3, 17, 200, 700, 64, 1500][seed], steps=8, move_fraction=[0.5, 0.3, 0.1, 0.05, 1.0, 0.02][seed])


@pytest.mark.parametrize("seed", range(3))
def test_rebuild_on_a_lattice(seed):
    # equal centres: splits that fall back to count / 2, predicates with ties
    run_world(100 + seed, n=[40, 300, 900][seed], steps=6, move_fraction=0.3, lattice=[4.0, 2.0, 1.0][seed])


def test_rebuild_with_proxies_created_between_steps():
    run_world(7, n=300, steps=16, move_fraction=0.2, churn=True)


def test_traversal_rank_is_the_query_order():
    """rank from leaf counts == position in the reference's s2DynamicTree_Query callback sequence"""
    import ctypes
    from tests import treebind
    rng = np.random.default_rng(5)
    ref = treebind.RefTree()
    try:
        boxes = random_boxes(rng, 400, 30.0, 1.0, 0.0)
        proxies = [ref.create_proxy(b, user=i) for i, b in enumerate(boxes)]
        for i in rng.choice(400, size=150, replace=False):
            boxes[i] += np.float32(2.0)
            ref.enlarge(proxies[i], boxes[i])
        ref.rebuild()
        nodes, root = ref.nodes(), ref.root
        counts = tp.leaf_counts(nodes, root)
        seen = []
        CB = ctypes.CFUNCTYPE(ctypes.c_bool, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p)

        def cb(proxy, user, ctx):
            seen.append(proxy)
            return True
        L = refbind.lib()
        L.s2DynamicTree_Query.argtypes = [ctypes.POINTER(treebind.DynamicTree), treebind.Box, CB, ctypes.c_void_p]
        L.s2DynamicTree_Query.restype = None
        for q in ((-100, -100, 100, 100), (-5, -5, 8, 9), (10, -20, 30, 5)):
            del seen[:]
            L.s2DynamicTree_Query(ctypes.byref(ref.t), treebind.box(q), CB(cb), None)
            ranks = [tp.traversal_rank(nodes, counts, p) for p in seen]
            assert ranks == sorted(ranks) and len(set(ranks)) == len(ranks)
            if q[0] == -100:
                assert ranks == list(range(400))
    finally:
        ref.close()
