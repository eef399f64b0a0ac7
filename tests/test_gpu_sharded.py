"""SURVEY.md 8e behind the C-ABI (include/solver2d_amd.h: s2amd_sharded_*; csrc/sharded.hip): one process, N shards -- here N logical
shards on this box's one GPU (the peer copies of the exchange become device copies; everything else is the code N GPUs run).

  * a sharded world == the unsharded world, BIT FOR BIT, body state and impulses (islands share no movable body and every shard keeps
    the pool order of its own constraints: the property tests/test_islands.py proves on the CPU against the reference);
  * ... and == the oracle, every shard's solve swept in that shard's own device order;
  * every shard's device ends a step with the whole world's body records (the exchange);
  * the partition is solver2d_amd/islands.py's, integer for integer -- the Python statement stays the reference for it;
  * re-sharding when a created contact joins islands of two shards and is destroyed again == one solver through the same script, and
    only the smaller island moves.
"""
import os

import numpy as np
import pytest

from solver2d_amd import hip, islands as isl, synthetic, wire
from tests import common, oraclebind
from tests.test_islands_dist import _merging_script, _spare

pytestmark = pytest.mark.gpu


def _mixed_world():
    """twelve pyramids of three sizes: islands of unequal weight, so that the bin packing has something to decide"""
    parts = [synthetic.pyramid(b, count=4) for b in (4, 7, 10)]
    bodies, contacts, joints = [], [], []
    offset = 0
    for k, (b, c, j) in enumerate(parts):
        b = b.copy()
        b["position"][:, 0] += 400.0 * k
        c = c.copy()
        c["bodyA"] += offset
        c["bodyB"] += offset
        bodies.append(b), contacts.append(c), joints.append(j)
        offset += len(b)
    return np.concatenate(bodies), np.concatenate(contacts), np.concatenate(joints)


def _unsharded(world, params, steps):
    b, c, j = common.copy3(world)
    with hip.Solver(0) as s:
        s.upload(b, c, j)
        for _ in range(steps):
            s.step_resident(params)
        s.download(b, c, j)
    return b, c, j


@pytest.mark.parametrize("shards", [1, 2, 3, 4])
@pytest.mark.parametrize("solver_name", ["TGS_Soft", "PGS_NGS_Block", "Jacobi"])
def test_sharded_world_equals_the_unsharded_world(shards, solver_name):
    vel, pos = common.DEFAULT_ITERS[solver_name]
    params = wire.StepParams.make(solver_name, 1.0 / 60.0, vel, pos, True)
    world = _mixed_world()
    want = _unsharded(world, params, 4)
    got = common.copy3(world)
    with hip.ShardedSolver([0] * shards) as sh:
        sh.upload(*world)
        owner, n_islands, reshards = sh.partition()
        for _ in range(4):
            sh.step(params)
        records = [sh.read_bodies(k) for k in range(shards)]
        sh.download(*got)
    assert n_islands == 12 and reshards == 0
    common.compare_exact(got, want, "%d shards, %s" % (shards, solver_name))
    # the partition: islands.py's, integer for integer
    pshards, island, shard_of_island = isl.shard_world(*world, shards)
    expect = np.where(island >= 0, shard_of_island[np.maximum(island, 0)], -1)
    assert np.array_equal(owner, expect)
    # the exchange: every shard's device holds every body's records
    b = got[0]
    for k in range(shards):
        r = records[k]
        assert np.array_equal(r[:, 0:2].view(np.uint32), b["position"].view(np.uint32)), k
        assert np.array_equal(r[:, 2:4].view(np.uint32), b["rot"].view(np.uint32)), k
        assert np.array_equal(r[:, 4:6].view(np.uint32), b["linearVelocity"].view(np.uint32)), k
        assert np.array_equal(r[:, 6].view(np.uint32), b["angularVelocity"].view(np.uint32)), k


def test_every_shard_against_the_oracle_in_its_own_order():
    """One step: the shards' sub-worlds as islands.py extracts them, each solved by the oracle in the sweep order its shard's solver
    reports (s2amd_sharded_solver -> s2amd_get_contact_order), merged back -- against the sharded solver's download."""
    params = wire.StepParams.make("TGS_Soft", 1.0 / 60.0, 8, 4, True)
    world = _mixed_world()
    got = common.copy3(world)
    with hip.ShardedSolver([0, 0, 0]) as sh:
        sh.upload(*world)
        sh.step(params)
        orders = []
        for k in range(3):
            order, _ = sh.shard(k).contact_order()
            jorder, _ = sh.shard(k).joint_order()
            orders.append((order, jorder))
        sh.download(*got)
    pshards, _island, _soi = isl.shard_world(*common.copy3(world), 3)
    for k, ps in enumerate(pshards):
        oraclebind.solve(params, ps.bodies, ps.contacts, ps.joints, contact_order=orders[k][0], joint_order=orders[k][1])
    want = common.copy3(world)
    isl.merge_back(*want, pshards)
    rank = np.cumsum(want[1]["pointCount"] > 0) - 1
    want[1]["constraintIndex"] = np.where(want[1]["pointCount"] > 0, rank, -1)
    common.compare_exact(got, want, "three shards against the oracle")


def test_resharding_when_a_contact_joins_islands_of_two_shards():
    """tests/test_gpu_reshard.py's script through the C-ABI: after step 1 a contact appears between island 0 and island 1 (different
    shards), after step 3 it is destroyed again."""
    steps = 6
    params = wire.StepParams.make("TGS_Soft", 1.0 / 60.0, 8, 4, True)
    world = _spare(synthetic.pyramid(6, count=4), 1)
    # ONE solver through the script
    b, c, j = common.copy3(world)
    with hip.Solver(0) as s:
        s.upload(b, c, j)
        for step in range(steps):
            s.step_resident(params)
            new = _merging_script((b, c, j), step)
            if new is not None:
                s.download(b, c, j)
                c = _merging_script((b, c, j), step)
                s.upload(b, c, j)
        s.download(b, c, j)
    # two shards
    gb, gc, gj = common.copy3(world)
    owners = []
    with hip.ShardedSolver([0, 0]) as sh:
        sh.upload(gb, gc, gj)
        owners.append(sh.partition()[0].copy())
        for step in range(steps):
            sh.step(params)
            if _merging_script((gb, gc, gj), step) is not None:
                sh.download(gb, gc, gj)
                new = _merging_script((gb, gc, gj), step)
                sh.reshard(contacts=new)
                gc = new
                owners.append(sh.partition()[0].copy())
        assert sh.partition()[2] == 2
        sh.download(gb, gc, gj)
        records = sh.read_bodies(1)
    for f in common.BODY_OUT:
        assert np.array_equal(common.bits(gb[f]), common.bits(b[f])), f
    for f in ("normalImpulse", "tangentImpulse"):
        assert np.array_equal(common.bits(gc["points"][f]), common.bits(c["points"][f])), f
    assert np.array_equal(records[:, 0:2].view(np.uint32), b["position"].view(np.uint32))
    island, n = isl.find_islands(*synthetic.pyramid(6, count=4))
    first, joined, after = ([int(o[np.flatnonzero(island == i)[0]]) for i in range(n)] for o in owners)
    assert first[0] != first[1] and joined[0] == joined[1] and joined[2:] == first[2:] and after[2:] == first[2:], (first, joined, after)


def test_sharded_api_state_errors():
    params = wire.StepParams.make("TGS_Soft", 1.0 / 60.0, 8, 4, True)
    with hip.ShardedSolver([0, 0]) as sh:
        with pytest.raises(hip.S2AmdError):
            sh.step(params)  # nothing uploaded
        with pytest.raises(hip.S2AmdError):
            sh.shard(2)
    with pytest.raises(hip.S2AmdError):
        hip.ShardedSolver([])


@pytest.mark.parametrize("mode,shards", [("stores", 3), ("copies", 3), ("copies", 2), ("rccl", 1)])
def test_pipelined_steps_in_every_form_of_the_exchange(mode, shards, monkeypatch):
    """s2amd_sharded_step_async x 5, one s2amd_sharded_wait: the exchange as stores into every world copy (shards of one device), as
    round 5's peer copies (forced), and -- one rank, all this box has -- through RCCL's all-gather; every form ends with the unsharded
    world's bits on the host and in every shard's copy of the body records, and enqueues what s2amd_sharded_count_ops says it does."""
    import ctypes
    monkeypatch.setenv("S2AMD_SHARDED_EXCHANGE", mode)
    params = wire.StepParams.make("TGS_Soft", 1.0 / 60.0, 8, 4, True)
    world = _mixed_world()
    want = _unsharded(world, params, 5)
    got = common.copy3(world)
    with hip.ShardedSolver([0] * shards) as sh:
        sh.upload(*world)
        for _ in range(5):
            sh.step_async(params)
        ops, waits, form = sh.step_ops()
        assert waits == 0
        sh.wait()
        assert sh.step_ops()[1] == 1
        records = [sh.read_bodies(k) for k in range(shards)]
        sh.download(*got)
    if mode == "rccl" and form != 1 and not os.path.exists("/opt/rocm/lib/librccl.so"):
        pytest.skip("no librccl.so on this box: the exchange fell back to peer copies")
    assert form == {"stores": 0, "copies": 2, "rccl": 1}[mode]
    counted = ctypes.c_int32()
    assert hip.load().s2amd_sharded_count_ops(shards, form, ctypes.byref(counted)) == 0
    assert ops == counted.value
    common.compare_exact(got, want, "%s, %d shards, pipelined" % (mode, shards))
    live = want[0]["type"] >= 0
    for r in records:
        assert np.array_equal(r[live, 0:2], want[0]["position"][live]) and np.array_equal(r[live, 4:6], want[0]["linearVelocity"][live])


def test_sharded_upload_refuses_constraints_that_name_no_body():
    world = _mixed_world()
    bad = common.copy3(world)
    k = int(np.flatnonzero(bad[1]["pointCount"] > 0)[0])
    bad[1]["bodyB"][k] = len(bad[0]) + 5
    with hip.ShardedSolver([0, 0]) as sh:
        with pytest.raises(hip.S2AmdError):
            sh.upload(*bad)
        with pytest.raises(hip.S2AmdError):
            sh.upload(np.zeros(0, dtype=wire.body_dtype), world[1], world[2])  # constraints with points and no bodies at all
        sh.upload(*world)
        with pytest.raises(hip.S2AmdError):
            sh.reshard(contacts=bad[1])
        # ... and the refused reshard left the solver as it was
        params = wire.StepParams.make("TGS_Soft", 1.0 / 60.0, 8, 4, True)
        sh.step(params)
        got = common.copy3(world)
        sh.download(*got)
    common.compare_exact(got, _unsharded(world, params, 1), "after a refused reshard")
