"""N > 1 path on CPU: two processes over gloo, islands sharded across ranks, per-step all-gather of
the owned body records.  The oracle stands in for the HIP solver (this is a test of the sharding /
exchange logic; the kernels are covered by the -m gpu tests).  Result must equal the single-process
solve of the whole world bit for bit."""
import os
import socket

import numpy as np
import pytest

from solver2d_amd import distributed, synthetic, wire
from tests import common, oraclebind


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world_size, port, steps, q, islands=7):
    try:
        _worker_body(rank, world_size, port, steps, q, islands)
    except Exception as e:  # surface the failure instead of leaving the parent waiting
        q.put(("error", repr(e)))
        raise


def _worker_body(rank, world_size, port, steps, q, islands=7):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    world = synthetic.pyramid(6, count=islands)
    params = wire.StepParams.make("TGS_Soft", 1.0 / 60.0, 8, 4, True)
    sw = distributed.ShardedWorld(*world, rank=rank, world_size=world_size)
    bodies = None
    for _ in range(steps):
        bodies = distributed.step_sharded(sw, lambda p, b, c, j: oraclebind.solve(p, b, c, j), params, dist=dist, torch=torch)
    if rank == 0:
        q.put(("ok", bodies.tobytes()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_ranks_gloo_equal_single_process():
    import torch.multiprocessing as mp
    steps = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, steps, q)) for r in range(2)]
    for p in procs:
        p.start()
    status, payload = q.get(timeout=120)
    assert status == "ok", payload
    got = np.frombuffer(payload, dtype=wire.body_dtype)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    whole = synthetic.pyramid(6, count=7)
    params = wire.StepParams.make("TGS_Soft", 1.0 / 60.0, 8, 4, True)
    for _ in range(steps):
        oraclebind.solve(params, *whole)
    for f in common.BODY_OUT:
        assert np.array_equal(got[f].view(np.uint32), whole[0][f].view(np.uint32)), f


def test_single_rank_path():
    world = synthetic.pyramid(5, count=3)
    params = wire.StepParams.make("TGS_Soft", 1.0 / 60.0, 8, 4, True)
    sw = distributed.ShardedWorld(*common.copy3(world), rank=0, world_size=1)
    out = distributed.step_sharded(sw, lambda p, b, c, j: oraclebind.solve(p, b, c, j), params)
    oraclebind.solve(params, *world)
    for f in common.BODY_OUT:
        assert np.array_equal(out[f], world[0][f])


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world_size,islands", [(4, 7), (4, 3)])
def test_more_ranks_gloo_equal_single_process(world_size, islands):
    """Four ranks: seven islands (uneven shards) and three islands (one rank owns NOTHING: its shard is empty, it still takes
    part in every all-gather)."""
    import torch.multiprocessing as mp
    steps = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world_size, port, steps, q, islands)) for r in range(world_size)]
    for p in procs:
        p.start()
    status, payload = q.get(timeout=200)
    assert status == "ok", payload
    got = np.frombuffer(payload, dtype=wire.body_dtype)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    whole = synthetic.pyramid(6, count=islands)
    params = wire.StepParams.make("TGS_Soft", 1.0 / 60.0, 8, 4, True)
    for _ in range(steps):
        oraclebind.solve(params, *whole)
    for f in common.BODY_OUT:
        assert np.array_equal(got[f].view(np.uint32), whole[0][f].view(np.uint32)), f


def _joining_contact(world, islands_of, a_island, b_island):
    """A two-point manifold between the top boxes of two pyramids (made-up geometry: a solver input, not a scene)."""
    bodies, contacts, _ = world
    a = int(np.flatnonzero(islands_of == a_island)[-1])
    b = int(np.flatnonzero(islands_of == b_island)[-1])
    c = contacts[int(np.flatnonzero(contacts["pointCount"] == 2)[0])].copy()
    c["bodyA"], c["bodyB"] = a, b
    c["normal"] = (1.0, 0.0)
    for p in range(2):
        c["points"][p]["localAnchorA"] = (0.5, 0.25 - 0.5 * p)
        c["points"][p]["localAnchorB"] = (-0.5, 0.25 - 0.5 * p)
        c["points"][p]["separation"] = -0.004
        c["points"][p]["normalImpulse"] = 0.0
        c["points"][p]["tangentImpulse"] = 0.0
    return c


def _merging_script(world, step):
    """The collision phase of the test, the same on every rank and in the single process: after step 1 a contact appears between
    island 0 and island 1 (different ranks under the first partition), after step 3 it is destroyed again."""
    from solver2d_amd import islands as isl
    bodies, contacts, joints = world
    slot = len(contacts) - 1
    if step == 1:
        island, _n = isl.find_islands(bodies, contacts, joints)
        new = contacts.copy()
        new[slot] = _joining_contact(world, island, 0, 1)
        return new
    if step == 3:
        new = contacts.copy()
        new[slot]["bodyA"], new[slot]["bodyB"], new[slot]["pointCount"] = -1, -1, 0
        return new
    return None


def _spare(world, n):
    b, c, j = world
    free = np.zeros(n, dtype=wire.contact_dtype)
    free["bodyA"], free["bodyB"], free["constraintIndex"] = -1, -1, -1
    return b, np.concatenate([c, free]), j


def _reshard_worker(rank, world_size, port, steps, q):
    try:
        import torch
        import torch.distributed as dist
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world_size)
        world = _spare(synthetic.pyramid(6, count=4), 1)
        params = wire.StepParams.make("TGS_Soft", 1.0 / 60.0, 8, 4, True)
        sw = distributed.ShardedWorld(*world, rank=rank, world_size=world_size)
        owners = [sw.owner_of_body().copy()]
        joined = None
        bodies = None
        for step in range(steps):
            bodies = distributed.step_sharded(sw, lambda p, b, c, j: oraclebind.solve(p, b, c, j), params, dist=dist, torch=torch)
            new = _merging_script((sw.bodies, sw.contacts, sw.joints), step)
            if new is not None:
                if step == 1:
                    joined = sw.joins_shards(int(new[-1]["bodyA"]), int(new[-1]["bodyB"]))
                sw.reshard(contacts=new, dist=dist, torch=torch)
                owners.append(sw.owner_of_body().copy())
        # the whole world's solver state, for the comparison: one more exchange of the constraint records
        sw.reshard(dist=dist, torch=torch)
        if rank == 0:
            q.put(("ok", bodies.tobytes(), sw.contacts.tobytes(), joined, [o.tobytes() for o in owners]))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:
        q.put(("error", repr(e), None, None, None))
        raise


@pytest.mark.timeout(300)
def test_islands_that_merge_across_ranks_are_resharded():
    """a contact created between two pyramids that live on different ranks joins their islands; every rank finds
    the islands again, the smaller island moves to the rank of the larger (everything else stays), the moved constraints take their
    impulses along -- and the sharded world stays bit-identical to the single process, also after the contact is destroyed again."""
    import torch.multiprocessing as mp
    steps = 6
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_reshard_worker, args=(r, 2, port, steps, q)) for r in range(2)]
    for p in procs:
        p.start()
    status, body_bytes, contact_bytes, joined, owners = q.get(timeout=200)
    assert status == "ok", body_bytes
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    got_b = np.frombuffer(body_bytes, dtype=wire.body_dtype)
    got_c = np.frombuffer(contact_bytes, dtype=wire.contact_dtype)
    whole = common.copy3(_spare(synthetic.pyramid(6, count=4), 1))
    params = wire.StepParams.make("TGS_Soft", 1.0 / 60.0, 8, 4, True)
    for step in range(steps):
        oraclebind.solve(params, *whole)
        new = _merging_script(whole, step)
        if new is not None:
            # (the single process keeps the solver state of every slot that kept its pair, like ShardedWorld.reshard)
            same = (new["bodyA"] == whole[1]["bodyA"]) & (new["bodyB"] == whole[1]["bodyB"]) & (whole[1]["bodyA"] >= 0)
            for f in ("normalImpulse", "tangentImpulse"):
                new["points"][f][same] = whole[1]["points"][f][same]
            whole = (whole[0], new, whole[2])
    assert joined is True, "the created contact joined islands of two ranks"
    for f in common.BODY_OUT:
        assert np.array_equal(got_b[f].view(np.uint32), whole[0][f].view(np.uint32)), f
    live = whole[1]["pointCount"] > 0
    for f in ("normalImpulse", "tangentImpulse"):
        assert np.array_equal(got_c["points"][f][live].view(np.uint32), whole[1]["points"][f][live].view(np.uint32)), f
    # the partition: islands 0 and 1 on different ranks first, on one rank while joined; the islands the contact never touched stay put
    o0, o1, o2 = (np.frombuffer(o, dtype=np.int32) for o in owners)
    from solver2d_amd import islands as isl
    island, n = isl.find_islands(*synthetic.pyramid(6, count=4))
    first = [int(o0[np.flatnonzero(island == i)[0]]) for i in range(n)]
    joined_at = [int(o1[np.flatnonzero(island == i)[0]]) for i in range(n)]
    after = [int(o2[np.flatnonzero(island == i)[0]]) for i in range(n)]
    assert first[0] != first[1] and joined_at[0] == joined_at[1] and joined_at[2:] == first[2:] and after[2:] == first[2:], (first, joined_at, after)
