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
