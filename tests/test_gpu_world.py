"""Resident world chain on the device (s2amd_world_upload / _step / _download, solver2d_amd/csrc/world.hip): stage 3
update contacts -> s2Solve_* -> stage 4 refit on arrays that stay in HBM, against the same chain through the CPU oracle
(tests/world_chain.py, pinned to the reference by tests/test_world_chain.py) with the solve swept in the order the
device reports.  Bit for bit, every array, every step."""
import glob
import os

import numpy as np
import pytest

from solver2d_amd import hip, synthetic, wire
from tests import world_chain

pytestmark = pytest.mark.gpu

FILES = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "world_*.npz")))


def run_chain(s, params, world, steps, what):
    """Uploads `world`, steps both sides, compares after every step; returns the device's last state and the infos."""
    ref = world_chain.copy_world(world)
    s.world_upload(*[world[k] for k in world_chain.WORLD_KEYS])
    infos = []
    got = None
    for step in range(steps):
        info = s.world_step(params)
        order, _ = s.contact_order()
        jorder, _ = s.joint_order()
        status = world_chain.oracle_world_step(params, ref, contact_order=order, joint_order=jorder)
        out = world_chain.copy_world(world)
        res = s.world_download(*[out[k] for k in world_chain.WORLD_KEYS])
        got = dict(zip(world_chain.WORLD_KEYS, res[:6]))
        world_chain.assert_device_equals_oracle(got, ref, "%s step %d" % (what, step))
        assert np.array_equal(res[6], status), "%s step %d: stage-3 status" % (what, step)
        assert info["separatedCount"] == int((status == wire.PAIR_SEPARATED).sum())
        assert info["activeContacts"] == int((ref["contacts"]["pointCount"] > 0).sum())
        live_shapes = ref["shapes"]["type"] != wire.SHAPE_FREE
        assert info["movedCount"] == int((ref["shapes"]["enlarged"][live_shapes] != 0).sum())
        infos.append(info)
    return got, infos


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(p)[6:-4] for p in FILES])
def test_golden_world_chains(path):
    d = np.load(path)
    params = world_chain.params_of(d)
    world = world_chain.load_world(d)
    with hip.Solver(0) as s:
        run_chain(s, params, world, int(d["steps"][0]) + 3, os.path.basename(path))


@pytest.mark.parametrize("solver_name", wire.SOLVER_NAMES)
def test_golden_world_every_solver(solver_name):
    """The shapes-zoo world (all shape types, joints) through every driver."""
    from tests import common
    path = [f for f in FILES if "shapes_zoo" in f][0]
    d = np.load(path)
    vel, pos = common.DEFAULT_ITERS[solver_name]
    params = wire.StepParams.make(solver_name, 1.0 / 60.0, vel, pos, True)
    with hip.Solver(0) as s:
        run_chain(s, params, world_chain.load_world(d), 4, "shapes_zoo/" + solver_name)
