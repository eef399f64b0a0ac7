"""Resident world chain (stage 3 update contacts -> s2Solve_* -> stage 4 refit), CPU side: the oracle's three stage
restatements chained by tests/world_chain.py must reproduce the UNMODIFIED reference's world k steps later, bit for bit
(identity constraint order), on windows where the reference's stage 1 created no contact.  Fixtures:
tests/golden/world_*.npz (tests/golden/make_golden.py).  This pins the checker of tests/test_gpu_world.py."""
import glob
import os

import numpy as np
import pytest

from solver2d_amd import wire
from tests import world_chain

# "_k0" fixtures are inputs only (for the GPU loop tests): nothing to replay here
FILES = [f for f in sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "world_*.npz"))) if not f.endswith("_k0.npz")]


def test_fixtures_present():
    assert len(FILES) >= 7


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(p)[6:-4] for p in FILES])
def test_oracle_chain_equals_reference(path):
    d = np.load(path)
    params = world_chain.params_of(d)
    world = world_chain.load_world(d)
    want = world_chain.load_world(d, "out_")
    separated = 0
    for _ in range(int(d["steps"][0])):
        status = world_chain.oracle_world_step(params, world)
        separated += int((status == wire.PAIR_SEPARATED).sum())
    world_chain.assert_worlds_equal(world, want, os.path.basename(path))
    if "pyramid" not in path and "joint_grid" not in path:
        assert separated > 0, "the window was chosen to contain a separation"
