"""SURVEY.md 8e on the device: the resident island-sharded world (solver2d_amd/distributed.py: ResidentShardedWorld) when a created
contact joins islands that live on different ranks and is destroyed again -- ResidentShardedWorld.reshard: the shards' solver state
comes down, the constraint state is exchanged once, islands are found and partitioned again (the smaller part of a merged island
moves, everything else stays), the new shards go up.  Two ranks share this box's one GPU and exchange through gloo (the collective
path of the tests; RCCL needs two GPUs); the result must equal ONE solver stepping the whole world through the same script, bit for
bit.  (The CPU form of the same scenario, against the oracle: tests/test_islands_dist.py.)"""
import os

import numpy as np
import pytest

from solver2d_amd import wire
from tests import common
from tests.test_islands_dist import _free_port, _merging_script, _spare

pytestmark = pytest.mark.gpu

STEPS = 6


def _world():
    from solver2d_amd import synthetic
    return _spare(synthetic.pyramid(6, count=4), 1)


def _worker(rank, world_size, port, q):
    try:
        import torch
        import torch.distributed as dist
        from solver2d_amd import distributed, hip
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world_size)
        torch.cuda.set_device(0)
        params = wire.StepParams.make("TGS_Soft", 1.0 / 60.0, 8, 4, True)
        sw = distributed.ShardedWorld(*_world(), rank=rank, world_size=world_size)
        owners = [sw.owner_of_body().copy()]
        with hip.Solver(0) as s:
            s.set_option("async", 1)
            rw = distributed.ResidentShardedWorld(sw, s, torch, dist=dist, backend="gloo")
            for step in range(STEPS):
                rw.run(params, 1)
                new = _merging_script((rw.sw.bodies, rw.sw.contacts, rw.sw.joints), step)
                if new is not None:
                    rw.reshard(contacts=new)
                    owners.append(rw.sw.owner_of_body().copy())
            bodies = rw.world_bodies()
            rw.close()
        if rank == 0:
            q.put(("ok", bodies.tobytes(), [o.tobytes() for o in owners]))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:  # surface the failure instead of leaving the parent waiting
        import traceback
        q.put(("error", repr(e) + "\n" + traceback.format_exc()[-1500:], None))
        raise


@pytest.mark.timeout(600)
def test_resident_shards_are_resharded_when_a_contact_joins_islands_of_two_ranks():
    import torch.multiprocessing as mp
    from solver2d_amd import hip, islands as isl
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    status, payload, owners = q.get(timeout=400)
    assert status == "ok", payload
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    got = np.frombuffer(payload, dtype=np.float32).reshape(-1, 8)
    # the same world, the same script, ONE solver: the created contact goes into the downloaded arrays (every other slot keeps its
    # impulses, as reshard keeps them), the arrays go up again
    b, c, j = common.copy3(_world())
    params = wire.StepParams.make("TGS_Soft", 1.0 / 60.0, 8, 4, True)
    with hip.Solver(0) as s:
        s.upload(b, c, j)
        for step in range(STEPS):
            s.step_resident(params)
            s.synchronize()
            new = _merging_script((b, c, j), step)
            if new is not None:
                s.download(b, c, j)
                new = _merging_script((b, c, j), step)
                c = new
                s.upload(b, c, j)
        s.download(b, c, j)
    assert np.array_equal(got[:, 0:2].view(np.uint32), b["position"].view(np.uint32))
    assert np.array_equal(got[:, 2:4].view(np.uint32), b["rot"].view(np.uint32))
    assert np.array_equal(got[:, 4:6].view(np.uint32), b["linearVelocity"].view(np.uint32))
    assert np.array_equal(got[:, 6].view(np.uint32), b["angularVelocity"].view(np.uint32))
    # the partition: islands 0 and 1 on different ranks first, on ONE while the contact joins them; the others never move
    o0, o1, o2 = (np.frombuffer(o, dtype=np.int32) for o in owners)
    from solver2d_amd import synthetic
    island, n = isl.find_islands(*synthetic.pyramid(6, count=4))
    first = [int(o0[np.flatnonzero(island == i)[0]]) for i in range(n)]
    joined = [int(o1[np.flatnonzero(island == i)[0]]) for i in range(n)]
    after = [int(o2[np.flatnonzero(island == i)[0]]) for i in range(n)]
    assert first[0] != first[1] and joined[0] == joined[1] and joined[2:] == first[2:] and after[2:] == first[2:], (first, joined, after)
