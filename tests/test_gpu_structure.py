"""Device-side constraint-graph structure: islands == solver2d_amd/islands.py, colours == the
greedy colouring in descending priority-hash order (tests/structure_ref.py).  Integer results, compared exactly."""
import numpy as np
import pytest

from solver2d_amd import hip, islands, synthetic, wire
from tests import fuzz_worlds, golden_util, structure_ref

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def solver():
    s = hip.Solver(0)
    yield s
    s.close()


def worlds():
    out = [("pyramid40", synthetic.pyramid(40)), ("jointgrid12", synthetic.joint_grid(12)), ("platform", synthetic.platform(20, layers=6))]
    for seed in range(6):
        out.append(("fuzz%d" % seed, fuzz_worlds.random_world(seed + 300, n_bodies=80 + 40 * seed, n_contacts=150 + 90 * seed, n_joints=5 * seed)))
    for path in golden_util.golden_files()[::9]:
        _params, pre, _post = golden_util.load(path)
        out.append((path.split("/")[-1][:-4], pre))
    return out


WORLDS = worlds()


@pytest.mark.parametrize("name,world", WORLDS, ids=[w[0] for w in WORLDS])
def test_islands_equal_the_host_definition(solver, name, world):
    bodies, contacts, joints = world
    want, count = islands.find_islands(bodies, contacts, joints)
    got, n = solver.find_islands(bodies, contacts, joints)
    assert n == count
    assert np.array_equal(got, want)


@pytest.mark.parametrize("name,world", WORLDS, ids=[w[0] for w in WORLDS])
def test_colours_equal_priority_greedy(solver, name, world):
    bodies, contacts, _joints = world
    got, n, rounds = solver.color_constraints(bodies, contacts)
    structure_ref.check_proper(bodies, contacts, got)
    want, count = structure_ref.color_constraints(bodies, contacts)
    assert n == count and np.array_equal(got, want)
    assert rounds >= 1 or count == 0


def test_many_islands_and_a_big_one(solver):
    """512 small pyramids next to one big one: islands numbered by lowest body index across all of them."""
    small = synthetic.pyramid(6, count=40)
    got, n = solver.find_islands(*small)
    want, count = islands.find_islands(*small)
    assert n == count == 40 and np.array_equal(got, want)
    big = synthetic.pyramid(120)
    got, n = solver.find_islands(*big)
    assert n == 1 and (got[big[0]["type"] == wire.BODY_DYNAMIC] == 0).all() and (got[big[0]["type"] == wire.BODY_STATIC] == -1).all()


def test_base_200_colours_and_timing(solver):
    bodies, contacts, joints = synthetic.pyramid(200)
    colour, n, rounds = solver.color_constraints(bodies, contacts)
    colour_ms = solver.stats()["deviceMs"]
    structure_ref.check_proper(bodies, contacts, colour)
    assert 6 <= n <= 12, n  # random-priority greedy needs a few more colours than the host's pool-order greedy (6 here)
    island, count = solver.find_islands(bodies, contacts, joints)
    island_ms = solver.stats()["deviceMs"]
    assert count == 1
    print("base-200: %d colours in %d rounds, %.3f ms; islands %.3f ms" % (n, rounds, colour_ms, island_ms))


def test_out_of_range_bodies_are_rejected(solver):
    bodies, contacts, joints = synthetic.pyramid(6)
    contacts = contacts.copy()
    contacts["bodyB"][3] = len(bodies) + 9
    with pytest.raises(hip.S2AmdError):
        solver.find_islands(bodies, contacts, joints)
    with pytest.raises(hip.S2AmdError):
        solver.color_constraints(bodies, contacts)
