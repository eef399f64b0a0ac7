"""Islands on the GPU: a sharded solve through the C-ABI equals the whole-world solve on the
oracle (each in the device's sweep order)."""
import numpy as np
import pytest

from solver2d_amd import hip, islands, synthetic, wire
from tests import common, oraclebind

pytestmark = pytest.mark.gpu


def test_sharded_gpu_solve_matches_oracle_per_shard():
    world = synthetic.pyramid(10, count=8)
    params = wire.StepParams.make("TGS_Soft", 1.0 / 60.0, 8, 4, True)
    shards, isl, _ = islands.shard_world(*world, n_shards=4)
    assert isl.max() + 1 == 8
    with hip.Solver(0) as gpu:
        for sh in shards:
            want = (sh.bodies.copy(), sh.contacts.copy(), sh.joints.copy())
            gpu.solve(params, sh.bodies, sh.contacts, sh.joints)
            order, _ = gpu.contact_order()
            jorder, _ = gpu.joint_order()
            oraclebind.solve(params, *want, contact_order=order, joint_order=jorder)
            common.compare_exact((sh.bodies, sh.contacts, sh.joints), want, "shard")
    out = common.copy3(world)
    islands.merge_back(*out, shards)
    assert np.isfinite(out[0]["position"]).all()
