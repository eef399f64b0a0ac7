"""SURVEY 8f row 1 (creation order): s2amd_world_find_pairs returns the new pairs as a SET; the binding
(shim/s2_amd_binding.c: s2amdBinding_OrderPairs) puts them into the sequence the reference's stage 1 creates contacts in
-- move-array order, reverse callback order, trees in query order, s2DynamicTree_Query's traversal order
(/root/reference/src/broad_phase.c:253-254, :288-357, src/dynamic_tree.c:1171-1210) -- so s2CreateContact hands out the
reference's pool slots.  Checked here on the CPU against the unmodified reference: its real s2CreateContact call sequence
is recorded step by step (oracle/ref_hook.c: __wrap_s2CreateContact), handed to the ordering function as a sorted set,
and must come back as recorded."""
import ctypes

import pytest

from tests import refbind

pytestmark = pytest.mark.skipif(not refbind.available(), reason="oracle/_ref/libs2ref.so not built")

WORLDS = [("pyramid", 20, 0, 60), ("mixed", 24, 0, 150), ("tumbler", 150, 0, 200), ("circle_pile", 16, 0, 120), ("shapes_zoo", 40, 0, 200),
          ("far_ragdoll_pile", 0, 0, 120), ("card_house", 0, 0, 60), ("ragdoll", 0, 0, 90), ("overlap_recovery", 0, 0, 40), ("arch", 0, 0, 40)]


@pytest.mark.parametrize("scene,p0,p1,steps", WORLDS)
def test_binding_orders_pairs_like_the_reference(scene, p0, p1, steps):
    L = refbind.lib()
    out = (ctypes.c_long * 3)()
    L.s2ref_order_check(1)
    try:
        with refbind.RefWorld(scene, "TGS_Soft", p0, p1) as world:
            for _ in range(steps):
                world.step(1.0 / 60.0, 8, 4, True)
        L.s2ref_order_check_result(out)
    finally:
        L.s2ref_order_check(0)
    checked, mismatched, with_pairs = out[0], out[1], out[2]
    assert with_pairs > 0 and checked >= with_pairs, "the scene created no contact"
    assert mismatched == 0, "%d of %d steps with new pairs came back in another order (%d pairs)" % (mismatched, with_pairs, checked)
