"""s2amd_update_contacts (solver2d_amd/csrc/narrowphase.hip) == Stage 3 of s2World_Step, BIT FOR BIT: manifolds,
feature ids, simplex caches, matched impulses.  Checked against the committed captures of the unmodified reference
(tests/golden/np_*.npz), against the oracle on the same inputs and -- when oracle/_ref is there -- against live
captures of every step of several worlds, through the C-ABI."""
import os

import numpy as np
import pytest

from solver2d_amd import hip, wire
from tests import golden_util, oraclebind, refbind
from tests.test_narrowphase_oracle import compare_narrowphase

pytestmark = pytest.mark.gpu

NP_FILES = golden_util.narrowphase_files()


@pytest.fixture(scope="module")
def solver():
    s = hip.Solver(0)
    yield s
    s.close()


@pytest.mark.parametrize("path", NP_FILES, ids=[os.path.basename(p)[:-4] for p in NP_FILES])
def test_gpu_narrow_phase_matches_reference_captures(solver, path):
    cap = dict(np.load(path))
    pairs, contacts = cap["pairs_pre"].copy(), cap["contacts_pre"].copy()
    status = solver.update_contacts(cap["bodies"], cap["origins"], cap["shapes"], pairs, contacts)
    assert compare_narrowphase(cap, pairs, contacts, status, os.path.basename(path)) > 0
    # and the oracle agrees on every byte the library wrote, free slots included
    opairs, ocontacts = cap["pairs_pre"].copy(), cap["contacts_pre"].copy()
    ostatus = oraclebind.update_contacts(cap["bodies"], cap["origins"], cap["shapes"], opairs, ocontacts)
    assert np.array_equal(status, ostatus)
    upd = status == wire.PAIR_UPDATED
    assert pairs[upd].tobytes() == opairs[upd].tobytes()
    assert contacts[upd].tobytes() == ocontacts[upd].tobytes()
    assert pairs[~upd].tobytes() == cap["pairs_pre"][~upd].tobytes() and contacts[~upd].tobytes() == cap["contacts_pre"][~upd].tobytes()


@pytest.mark.skipif(not refbind.available(), reason="oracle/_ref/libs2ref.so not built")
@pytest.mark.parametrize("scene,p0,steps", [("shapes_zoo", 60, 160), ("mixed", 24, 100), ("pyramid", 14, 8), ("tumbler", 80, 60)])
def test_gpu_narrow_phase_every_step_of_a_reference_world(solver, scene, p0, steps):
    updated = 0
    with refbind.RefWorld(scene, "TGS_Soft", p0, 0) as world:
        for step in range(steps):
            world.step_captured(1.0 / 60.0, 4, 2, True)
            cap = refbind.narrowphase_capture()
            if len(cap["contacts_pre"]) == 0:
                continue
            pairs, contacts = cap["pairs_pre"].copy(), cap["contacts_pre"].copy()
            status = solver.update_contacts(cap["bodies"], cap["origins"], cap["shapes"], pairs, contacts)
            updated += compare_narrowphase(cap, pairs, contacts, status, "%s step %d" % (scene, step))
    assert updated > 0


def test_bad_arguments_fail_loudly(solver):
    cap = dict(np.load(NP_FILES[0]))
    pairs, contacts = cap["pairs_pre"].copy(), cap["contacts_pre"].copy()
    pairs["shapeA"][np.flatnonzero(pairs["shapeA"] >= 0)[0]] = len(cap["shapes"]) + 5
    with pytest.raises(hip.S2AmdError):
        solver.update_contacts(cap["bodies"], cap["origins"], cap["shapes"], pairs, contacts)
