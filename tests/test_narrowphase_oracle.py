"""Parity link L1 for the narrow phase: the oracle's restatement of Stage 3 of s2World_Step
(s2UpdateContact: manifold functions, GJK with its cache, SAT, clipping, id matching) == the unmodified reference,
BIT FOR BIT, on states captured at the end of Stage 2 and at solver entry (oracle/ref_hook.c)."""
import numpy as np
import pytest

from solver2d_amd import wire
from tests import common, oraclebind, refbind

pytestmark = pytest.mark.ref

SCENES = [("pyramid", 12, 0, 6), ("mixed", 24, 0, 120), ("tumbler", 60, 0, 90), ("circle_pile", 30, 0, 80), ("vertical_stack", 8, 0, 40),
          ("multi_pyramid", 4, 6, 20), ("shapes_zoo", 60, 0, 200)]


def compare_narrowphase(cap, pairs, contacts, status, what):
    post_p, post_c = cap["pairs_post"], cap["contacts_post"]
    live_pre = cap["pairs_pre"]["shapeA"] >= 0
    live_post = post_p["shapeA"] >= 0
    # a contact Stage 3 destroyed is free at solver entry
    assert np.array_equal(status == wire.PAIR_SEPARATED, live_pre & ~live_post), what
    assert np.array_equal(status == wire.PAIR_FREE, ~live_pre), what
    upd = status == wire.PAIR_UPDATED
    for f in ("cacheMetric", "cacheCount", "id", "cacheIndexA", "cacheIndexB", "persisted"):
        a, b = pairs[f][upd], post_p[f][upd]
        if a.dtype == np.float32:
            a, b = a.view(np.uint32), b.view(np.uint32)
        assert np.array_equal(a, b), "%s: pair field %s" % (what, f)
    for f in ("pointCount", "frictionPersisted", "constraintIndex"):
        assert np.array_equal(contacts[f][upd], post_c[f][upd]), "%s: contact field %s" % (what, f)
    for f in ("normal", "friction"):
        assert np.array_equal(contacts[f][upd].view(np.uint32), post_c[f][upd].view(np.uint32)), "%s: contact field %s" % (what, f)
    got, want = contacts["points"][upd], post_c["points"][upd]
    for f in got.dtype.names:
        assert np.array_equal(got[f].view(np.uint32), want[f].view(np.uint32)), "%s: point field %s" % (what, f)
    return int(upd.sum())


@pytest.mark.parametrize("scene,p0,p1,steps", SCENES, ids=[s[0] for s in SCENES])
def test_oracle_narrow_phase_equals_reference_every_step(scene, p0, p1, steps):
    updated = 0
    with refbind.RefWorld(scene, "TGS_Soft", p0, p1) as world:
        for step in range(steps):
            world.step_captured(1.0 / 60.0, 4, 2, True)
            cap = refbind.narrowphase_capture()
            if len(cap["contacts_pre"]) == 0:
                continue
            pairs, contacts = cap["pairs_pre"].copy(), cap["contacts_pre"].copy()
            status = oraclebind.update_contacts(cap["bodies"], cap["origins"], cap["shapes"], pairs, contacts)
            updated += compare_narrowphase(cap, pairs, contacts, status, "%s step %d" % (scene, step))
    assert updated > 0


def test_captures_cover_every_manifold_function():
    """The scenes above must exercise every primary shape-type pair of src/contact.c:139-151 and 1- and 2-point manifolds."""
    seen, counts = set(), set()
    for scene, p0, steps in (("shapes_zoo", 60, 200), ("mixed", 24, 120), ("circle_pile", 30, 80)):
        with refbind.RefWorld(scene, "TGS_Soft", p0, 0) as world:
            for step in range(steps):
                world.step_captured(1.0 / 60.0, 4, 2, True)
                if step % 10:
                    continue
                cap = refbind.narrowphase_capture()
                live = cap["pairs_post"]["shapeA"] >= 0
                ta = cap["shapes"]["type"][cap["pairs_post"]["shapeA"][live]]
                tb = cap["shapes"]["type"][cap["pairs_post"]["shapeB"][live]]
                seen |= set(zip(ta.tolist(), tb.tolist()))
                counts |= set(cap["contacts_post"]["pointCount"][live].tolist())
    C, O, P, S = wire.SHAPE_CAPSULE, wire.SHAPE_CIRCLE, wire.SHAPE_POLYGON, wire.SHAPE_SEGMENT
    assert {(O, O), (C, O), (C, C), (P, O), (P, C), (P, P), (S, O), (S, C), (S, P)} <= seen, seen
    assert {0, 1, 2} <= counts
