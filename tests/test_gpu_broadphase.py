"""Stage 4 (AABB refit) and Stage 1 (pair discovery) on the GPU against the oracle and, through the
capture hooks, against the unmodified reference: AABBs bit-exact, pair lists exactly equal (set,
orientation, sorted order)."""
import numpy as np
import pytest

from solver2d_amd import hip, wire
from tests import oraclebind, refbind

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not refbind.available(), reason="oracle/_ref/libs2ref.so not built")]

SCENES = [("pyramid", 25, 0, 30), ("mixed", 24, 0, 120), ("joint_grid", 8, 8, 30), ("tumbler", 200, 0, 90), ("circle_pile", 20, 0, 60)]


@pytest.mark.parametrize("scene,p0,p1,steps", SCENES)
def test_gpu_refit_and_pairs_match_reference(scene, p0, p1, steps):
    pairs_seen = 0
    with hip.Solver(0) as gpu, refbind.RefWorld(scene, "TGS_Soft", p0, p1) as w:
        for step in range(steps):
            shapes_before, origins_before = w.pack_shapes()
            _params, pre, post = w.step_captured(1.0 / 60.0, 8, 4, True)
            shapes_after, origins_after = w.pack_shapes()
            bp_shapes, moved, existing, created = refbind.broadphase_capture()

            got = gpu.find_pairs(pre[0], bp_shapes, moved, existing, pre[2])
            want = created[np.lexsort((created[:, 1], created[:, 0]))] if len(created) else created.reshape(0, 2)
            assert got.tolist() == want.tolist(), "step %d" % step
            assert got.tolist() == oraclebind.find_pairs(pre[0], bp_shapes, moved, existing, pre[2]).tolist()
            pairs_seen += len(got)

            shapes, origins = shapes_before.copy(), origins_before.copy()
            gpu.refit_shapes(post[0], shapes, origins)
            oshapes, oorigins = shapes_before.copy(), origins_before.copy()
            oraclebind.refit_shapes(post[0], oshapes, oorigins)
            live = shapes_after["type"] >= 0
            for f in ("aabb", "fatAABB"):
                assert np.array_equal(shapes[f][live].view(np.uint32), shapes_after[f][live].view(np.uint32)), (step, f)
                assert np.array_equal(shapes[f].view(np.uint32), oshapes[f].view(np.uint32))
            assert np.array_equal(shapes["enlarged"], oshapes["enlarged"])
            moving = (post[0]["type"] == wire.BODY_DYNAMIC) | (post[0]["type"] == wire.BODY_KINEMATIC)
            assert np.array_equal(origins[moving].view(np.uint32), origins_after[moving].view(np.uint32))
    if scene != "joint_grid":
        assert pairs_seen > 0


def test_gpu_pairs_large_first_step():
    """Every proxy has moved on the first step: base-100 pyramid, 5,051 shapes, 14,950 new pairs."""
    with hip.Solver(0) as gpu, refbind.RefWorld("pyramid", "TGS_Soft", 100, 0) as w:
        _params, pre, _post = w.step_captured(1.0 / 60.0, 8, 4, True)
        bp_shapes, moved, existing, created = refbind.broadphase_capture()
        got = gpu.find_pairs(pre[0], bp_shapes, moved, existing, pre[2])
        want = created[np.lexsort((created[:, 1], created[:, 0]))]
        assert len(want) >= 14950 and got.tolist() == want.tolist()
