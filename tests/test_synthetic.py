"""The synthetic benchmark snapshots are the reference's own step-0 solver inputs."""
import numpy as np
import pytest

from solver2d_amd import synthetic, wire
from tests import common, oraclebind, refbind


def test_pyramid_counts():
    b, c, j = synthetic.pyramid(200)
    assert len(b) == 20101 and len(c) == 59900 and (c["pointCount"] == 2).all()
    b, c, j = synthetic.pyramid(40, count=4)
    assert (b["type"] == wire.BODY_DYNAMIC).sum() == 4 * 820 and len(c) == 4 * 2380


def test_oracle_runs_on_synthetic_and_rests():
    st = synthetic.pyramid(12)
    p = wire.StepParams.make("TGS_Soft", 1.0 / 60.0, 8, 4, True)
    for _ in range(5):
        oraclebind.solve(p, *st)
    assert np.abs(st[0]["linearVelocity"]).max() < 0.5
    assert st[1]["points"]["normalImpulse"].max() > 0


@pytest.mark.skipif(not refbind.available(), reason="oracle/_ref/libs2ref.so not built")
@pytest.mark.parametrize("base", [4, 10, 25])
def test_pyramid_equals_reference_step0(base):
    with refbind.RefWorld("pyramid", "TGS_Soft", base, 0) as w:
        _params, pre, _post = w.step_captured(1.0 / 60.0, 8, 4, True)
    rb, rc, _ = pre
    sb, sc, _ = synthetic.pyramid(base)
    live = rb["type"] >= 0
    for f in rb.dtype.names:
        assert np.array_equal(rb[f][live], sb[f]), f

    def key(c):
        return sorted((int(x["bodyA"]), int(x["bodyB"]), x["normal"].tobytes(), x["friction"].tobytes(), x["points"].tobytes())
                      for x in c if x["pointCount"] > 0)
    assert key(rc) == key(sc)


@pytest.mark.skipif(not refbind.available(), reason="oracle/_ref/libs2ref.so not built")
def test_joint_grid_equals_reference_step0():
    with refbind.RefWorld("joint_grid", "PGS_NGS", 12, 12) as w:
        _params, pre, _post = w.step_captured(1.0 / 60.0, 4, 2, True)
    rb, rc, rj = pre
    sb, sc, sj = synthetic.joint_grid(12)
    live = rb["type"] >= 0
    for f in rb.dtype.names:
        assert np.array_equal(rb[f][live], sb[f]), f
    livej = rj["type"] >= 0
    for f in rj.dtype.names:
        assert np.array_equal(rj[f][livej], sj[f]), f
    assert (rc["pointCount"] > 0).sum() == 0
