"""BASELINE.json configs 3, 4 and 5 at FULL size, parity-checked (VERDICT r1 "weak" 1): the regimes the small sizes of
the other test files do not reach -- a 230-contact hub body walked by a whole wave (Tumbler drum), 19,800 revolute
joints in four colours (JointGrid 100x100), 512 LDS groups / 1.2 M constraints (512 x base-40).

Same gate as tests/test_gpu_parity.py: the C-ABI result equals the oracle BIT FOR BIT when the oracle sweeps in the
order the library reports.  The oracle is only the checker (0.95 s per step at config 5).
"""
import numpy as np
import pytest

from solver2d_amd import hip, synthetic, wire
from tests import common, oraclebind, refbind, world_chain
from tests.test_gpu_parity import check_order_is_valid, gpu_vs_oracle, gpu_vs_oracle_loose

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tumbler_10k():
    """Config 3's solver input: the Tumbler with 10,000 boxes stepped by the UNMODIFIED reference until the boxes have settled in the
    turning drum, captured at s2Solve_* entry -- a committed fixture (tests/golden/make_golden.py: big_inputs), so the test runs (and
    fails, not skips) wherever the repository is."""
    import os
    from tests import golden_util
    path = os.path.join(golden_util.GOLDEN_DIR, "big_tumbler10000_input.npz")
    assert os.path.exists(path), "%s is missing: python tests/golden/make_golden.py (needs the reference)" % path
    z = np.load(path)
    pre = (z["pre_bodies"].copy(), z["pre_contacts"].copy(), z["pre_joints"].copy())
    assert pre[0].dtype == wire.body_dtype and pre[1].dtype == wire.contact_dtype and pre[2].dtype == wire.joint_dtype
    return pre


def test_config3_tumbler_10k_jacobi(tumbler_10k):
    pre = tumbler_10k
    active = int((pre[1]["pointCount"] > 0).sum())
    assert active > 20000 and int((pre[0]["type"] == wire.BODY_DYNAMIC).sum()) == 10001
    # the drum is the hub: hundreds of boxes lean on one body
    drum_degree = int(np.bincount(np.concatenate([pre[1]["bodyA"], pre[1]["bodyB"]])[np.tile(pre[1]["pointCount"] > 0, 2)]).max())
    assert drum_degree > 100
    params = wire.StepParams.make("Jacobi", 1.0 / 60.0, 4, 2, True)
    with hip.Solver(0) as s:
        state = common.copy3(pre)
        for step in range(3):
            state = gpu_vs_oracle_loose(s, params, state, "Tumbler 10k / Jacobi step %d" % step)
        assert s.stats()["constraintCount"] == active


def test_config3b_tumbler_10k_tgs_soft(tumbler_10k):
    """The same captured input under the headline solver: > 200 colours, parallel batches + the sequential LDS tail."""
    params = wire.StepParams.make("TGS_Soft", 1.0 / 60.0, 8, 4, True)
    with hip.Solver(0) as s:
        state = gpu_vs_oracle_loose(s, params, tumbler_10k, "Tumbler 10k / TGS_Soft step 0")
        gpu_vs_oracle_loose(s, params, state, "Tumbler 10k / TGS_Soft step 1")
        assert s.stats()["contactColors"] > 100


@pytest.mark.parametrize("solver_name", ["PGS_NGS", "TGS_Soft", "PGS_NGS_Block"])
def test_config4_joint_grid_100x100(solver_name):
    """JointGrid exactly as the reference sample sizes it (samples/collection/sample_joints.cpp:377-446, numi = numk = 100):
    10,000 circles, 19,800 revolute joints, 7 static anchors; PGS_NGS is BASELINE's solver, the other two are the headline
    and the reference's default."""
    pre = synthetic.joint_grid(100)
    assert len(pre[2]) == 19800 and int((pre[0]["type"] == wire.BODY_STATIC).sum()) == 7
    vel, pos = common.DEFAULT_ITERS[solver_name]
    params = wire.StepParams.make(solver_name, 1.0 / 60.0, vel, pos, True)
    with hip.Solver(0) as s:
        state = common.copy3(pre)
        for step in range(4):
            state = gpu_vs_oracle(s, params, state, "JointGrid 100x100 / %s step %d" % (solver_name, step))
        st = s.stats()
        # one island of 10,000 bodies: strips on the op interpreter (generic_kernel.hip), prologue + one step kernel + epilogue
        assert st["jointCount"] == 19800 and st["stripCount"] >= 50 and st["persistent"] == 1 and st["kernelLaunches"] <= 5, st
    assert np.isfinite(state[0]["position"]).all()


def test_config5_512_pyramids_tgs_soft():
    """512 independent base-40 pyramids in ONE world (BASELINE configs[4] on one GPU): 419,840 boxes, 1,218,560 two-point
    constraints, 512 islands packed into LDS groups; two consecutive resident steps against the oracle."""
    pre = synthetic.pyramid(40, count=512)
    assert int((pre[1]["pointCount"] > 0).sum()) == 1218560
    params = wire.StepParams.make("TGS_Soft", 1.0 / 60.0, 8, 4, True)
    with hip.Solver(0) as s:
        s.upload(*pre)
        want = common.copy3(pre)
        for step in range(2):
            s.step_resident(params)
            order, offsets = s.contact_order()
            if step == 0:
                check_order_is_valid(order, offsets, pre[1], pre[0])
            oraclebind.solve(params, *want, contact_order=order)
            got = common.copy3(pre)
            s.download(*got)
            common.compare_exact(got, want, "512 x pyramid40 step %d" % step)
        st = s.stats()
        assert st["groupCount"] == 512 and st["constraintCount"] == 1218560 and st["kernelLaunches"] <= 4, st


def test_config5_as_a_resident_world_chain():
    """The same world through the whole step (stage 3 narrow phase on 1.2 M pairs -> solve -> stage 4) against the oracle
    chain: one step."""
    params = wire.StepParams.make("TGS_Soft", 1.0 / 60.0, 8, 4, True)
    world = synthetic.pyramid_world(40, count=512)
    ref = world_chain.copy_world(world)
    with hip.Solver(0) as s:
        s.world_upload(*[world[k] for k in world_chain.WORLD_KEYS])
        info = s.world_step(params)
        order, _ = s.contact_order()
        world_chain.oracle_world_step(params, ref, contact_order=order)
        out = world_chain.copy_world(world)
        res = s.world_download(*[out[k] for k in world_chain.WORLD_KEYS])
        world_chain.assert_device_equals_oracle(dict(zip(world_chain.WORLD_KEYS, res[:6])), ref, "512 x pyramid40 world step")
        assert info["activeContacts"] == 1218560 and info["separatedCount"] == 0


def test_config2_as_a_whole_loop_at_full_size():
    """BASELINE configs[1] as a WORLD at full size: LargePyramid base-200 (20,101 boxes, 59,900 manifolds), the whole loop of
    s2World_Step -- device pair query after every step that re-inflated a box (== the oracle's, nothing new), stage 3 on every
    pair, s2Solve_TGS_Soft on the persistent strip kernel from the second step on, stage 4 -- six steps, every array
    bit-exact against the oracle chain after each."""
    params = wire.StepParams.make("TGS_Soft", 1.0 / 60.0, 8, 4, True)
    world = synthetic.pyramid_world(200)
    ref = world_chain.copy_world(world)
    with hip.Solver(0) as s:
        s.world_upload(*[world[k] for k in world_chain.WORLD_KEYS])
        persistent = 0
        for step in range(6):
            if world_chain.moved_any(ref):
                got = s.world_find_pairs()
                want = world_chain.oracle_find_pairs(ref)
                assert np.array_equal(got, want) and len(got) == 0, "step %d: new pairs" % step
            info = s.world_step(params)
            order, _ = s.contact_order()
            world_chain.oracle_world_step(params, ref, contact_order=order)
            out = world_chain.copy_world(world)
            res = s.world_download(*[out[k] for k in world_chain.WORLD_KEYS])
            world_chain.assert_device_equals_oracle(dict(zip(world_chain.WORLD_KEYS, res[:6])), ref, "pyramid200 world step %d" % step)
            assert info["activeContacts"] == 59900 and info["separatedCount"] == 0
            persistent += s.stats()["persistent"]
        assert persistent >= 4, persistent
