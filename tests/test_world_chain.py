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
    # the windows of the mixed / shapes_zoo / circle_pile worlds were chosen to contain a separation
    if any(name in os.path.basename(path) for name in ("mixed", "shapes_zoo", "circle_pile")):
        assert separated > 0, "the window was chosen to contain a separation"


def test_fuzz_world_generators_run_through_the_oracle_chain():
    """The generators behind the GPU loop fuzz tests (tests/world_chain.py: rain_world, wreck_world) on the CPU: the whole
    loop -- pair query, contact creation, stage 3, solve, stage 4 -- through the oracle alone is deterministic, stays
    finite, and creates and destroys contacts (so the GPU tests that use them do compare something)."""
    from solver2d_amd import wire
    from tests import oraclebind, world_chain
    from tests.test_gpu_world import _create_contacts, _live_pairs

    def loop(world, steps):
        params = wire.StepParams.make("TGS_Soft", 1.0 / 60.0, 8, 4, True)
        created = separated = 0
        for _ in range(steps):
            if world_chain.moved_any(world):
                new = world_chain.oracle_find_pairs(world)
                if len(new):
                    created += len(new)
                    _create_contacts(world, new)
            status = world_chain.oracle_world_step(params, world)
            separated += int((status == wire.PAIR_SEPARATED).sum())
        return created, separated

    for make in (lambda: world_chain.rain_world(3, 60), lambda: world_chain.wreck_world(2, 14)):
        a, b = make(), make()
        ca, sa = loop(a, 40)
        cb, sb = loop(b, 40)
        assert (ca, sa) == (cb, sb) and ca > 20 and sa > 0, (ca, sa, cb, sb)
        for key in world_chain.WORLD_KEYS:
            assert a[key].tobytes() == b[key].tobytes(), key
        live = a["bodies"]["type"] != wire.BODY_FREE
        assert np.isfinite(a["bodies"]["position"][live]).all()
