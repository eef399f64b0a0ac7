"""One s2World_Step minus pair creation (stage 1) on wire arrays, through the CPU oracle:
stage 3 update contacts (src/world.c:132-168) -> s2Solve_* (src/world.c:206-256) -> stage 4 refit (src/world.c:259-301).
The same chain s2amd_world_step runs on the device (solver2d_amd/csrc/world.hip); test infrastructure only."""
import numpy as np

from solver2d_amd import wire
from tests import oraclebind

WORLD_KEYS = ("bodies", "contacts", "joints", "shapes", "pairs", "origins")


def copy_world(world):
    return {k: world[k].copy() for k in WORLD_KEYS}


def oracle_world_step(params, world, contact_order=None, joint_order=None):
    """In place; returns the stage-3 status.  A separated pair is destroyed as src/world.c:149-167 does: no manifold,
    free pair slot."""
    w = world
    w["origins"] = np.ascontiguousarray(w["origins"], dtype=np.float32)
    status = oraclebind.update_contacts(w["bodies"], w["origins"], w["shapes"], w["pairs"], w["contacts"])
    sep = status == wire.PAIR_SEPARATED
    w["contacts"]["pointCount"][sep] = 0
    w["pairs"]["shapeA"][sep] = -1
    w["pairs"]["shapeB"][sep] = -1
    oraclebind.solve(params, w["bodies"], w["contacts"], w["joints"], contact_order=contact_order, joint_order=joint_order)
    oraclebind.refit_shapes(w["bodies"], w["shapes"], w["origins"])
    # stage 4 also consumes the applied forces of every non-static body (src/world.c:274-275)
    moving = (w["bodies"]["type"] != wire.BODY_FREE) & (w["bodies"]["type"] != wire.BODY_STATIC)
    w["bodies"]["force"][moving] = 0.0
    w["bodies"]["torque"][moving] = 0.0
    return status


def load_world(npz, prefix=""):
    return {k: np.ascontiguousarray(npz[prefix + k]) for k in WORLD_KEYS}


def params_of(npz):
    p, f = npz["params"], npz["params_f"]
    import ctypes
    return wire.StepParams(int(p[0]), float(f[0]), int(p[1]), int(p[2]), int(p[3]), (ctypes.c_float * 2)(float(f[1]), float(f[2])))


def live_view(world):
    """The parts of a world the reference defines: manifolds and pair states of live pair slots only (a destroyed
    contact's pool slot holds whatever the pool left there), everything else in full."""
    live = world["pairs"]["shapeA"] >= 0
    out = {k: world[k] for k in ("bodies", "joints", "shapes", "origins")}
    out["contacts"] = world["contacts"][live]
    out["pairs"] = world["pairs"][live]
    out["live"] = live
    return out


def assert_worlds_equal(got, want, what):
    g, w = live_view(got), live_view(want)
    assert np.array_equal(g["live"], w["live"]), what + ": different live pair slots"
    for k in ("bodies", "joints", "shapes", "origins", "contacts", "pairs"):
        a, b = np.ascontiguousarray(g[k]), np.ascontiguousarray(w[k])
        if a.tobytes() != b.tobytes():
            if a.dtype.names:
                skip = {"pad", "enlarged"}  # enlarged: an output of the refit only (the reference keeps a move buffer instead)
                if k == "contacts":
                    # the reference leaves a stale constraintIndex in manifolds without points; the wire format says -1
                    act = b["pointCount"] > 0
                    if a["constraintIndex"][act].tobytes() != b["constraintIndex"][act].tobytes():
                        raise AssertionError("%s: contacts differ in constraintIndex" % what)
                    skip.add("constraintIndex")
                bad = [n for n in a.dtype.names if n not in skip and a[n].tobytes() != b[n].tobytes()]
                if not bad:
                    continue
                rows = np.flatnonzero([a[i].tobytes() != b[i].tobytes() for i in range(len(a))])
                raise AssertionError("%s: %s differ in fields %s, first rows %s" % (what, k, bad, rows[:5].tolist()))
            raise AssertionError("%s: %s differ" % (what, k))


def assert_device_equals_oracle(got, want, what):
    """Both sides are wire arrays of the same chain: every byte must agree, except constraintIndex of manifolds
    without points (see tests/common.py: compare_exact)."""
    from tests import common
    common.compare_exact((got["bodies"], got["contacts"], got["joints"]), (want["bodies"], want["contacts"], want["joints"]), what)
    g, w = got["contacts"].copy(), want["contacts"].copy()
    idle = w["pointCount"] <= 0
    g["constraintIndex"][idle] = 0
    w["constraintIndex"][idle] = 0
    for k, a, b in (("contacts", g, w), ("bodies", got["bodies"], want["bodies"]), ("joints", got["joints"], want["joints"]),
                    ("shapes", got["shapes"], want["shapes"]), ("pairs", got["pairs"], want["pairs"]),
                    ("origins", np.asarray(got["origins"], dtype=np.float32), np.asarray(want["origins"], dtype=np.float32))):
        a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
        if a.tobytes() == b.tobytes():
            continue
        if a.dtype.names:
            bad = [n for n in a.dtype.names if a[n].tobytes() != b[n].tobytes()]
            rows = np.flatnonzero([a[i].tobytes() != b[i].tobytes() for i in range(len(a))])
            raise AssertionError("%s: %s differ in fields %s, first rows %s" % (what, k, bad, rows[:5].tolist()))
        raise AssertionError("%s: %s differ" % (what, k))
