"""One s2World_Step minus pair creation (stage 1) on wire arrays, through the CPU oracle:
stage 3 update contacts (src/world.c:132-168) -> s2Solve_* (src/world.c:206-256) -> stage 4 refit (src/world.c:259-301).
The same chain s2amd_world_step runs on the device (solver2d_amd/csrc/world.hip); test infrastructure only."""
import numpy as np

from solver2d_amd import wire
from tests import oraclebind

WORLD_KEYS = ("bodies", "contacts", "joints", "shapes", "pairs", "origins")


def copy_world(world):
    return {k: world[k].copy() for k in WORLD_KEYS}


def oracle_world_step(params, world, contact_order=None, joint_order=None, reverse=False):
    """In place; returns the stage-3 status.  A separated pair is destroyed as src/world.c:149-167 does: no manifold,
    free pair slot.  reverse: sweep the active contacts in reversed pool order (the yardstick of order sensitivity)."""
    w = world
    w["origins"] = np.ascontiguousarray(w["origins"], dtype=np.float32)
    status = oraclebind.update_contacts(w["bodies"], w["origins"], w["shapes"], w["pairs"], w["contacts"])
    sep = status == wire.PAIR_SEPARATED
    w["contacts"]["pointCount"][sep] = 0
    w["pairs"]["shapeA"][sep] = -1
    w["pairs"]["shapeB"][sep] = -1
    if reverse:
        contact_order = np.flatnonzero(w["contacts"]["pointCount"] > 0)[::-1].astype(np.int32)
    oraclebind.solve(params, w["bodies"], w["contacts"], w["joints"], contact_order=contact_order, joint_order=joint_order)
    oraclebind.refit_shapes(w["bodies"], w["shapes"], w["origins"])
    # stage 4 also consumes the applied forces of every non-static body (src/world.c:274-275)
    moving = (w["bodies"]["type"] != wire.BODY_FREE) & (w["bodies"]["type"] != wire.BODY_STATIC)
    w["bodies"]["force"][moving] = 0.0
    w["bodies"]["torque"][moving] = 0.0
    return status


def live_pairs(world):
    live = world["pairs"]["shapeA"] >= 0
    return np.stack([world["pairs"]["shapeA"][live], world["pairs"]["shapeB"][live]], axis=1).astype(np.int32)


def oracle_find_pairs(world):
    """Stage 1's pair discovery on the oracle side of a chain: the move buffer = the shapes whose `enlarged` flag is set;
    the query consumes it (src/broad_phase.c: s2UpdateBroadPhasePairs clears moveArray / moveSet at its end), so every flag
    is zero afterwards -- what s2amd_world_find_pairs does to the resident shapes."""
    moved = ((world["shapes"]["enlarged"] != 0) & (world["shapes"]["type"] != wire.SHAPE_FREE)).astype(np.uint8)
    new = oraclebind.find_pairs(world["bodies"], world["shapes"], moved, live_pairs(world), world["joints"])
    world["shapes"]["enlarged"] = 0
    return new


def moved_any(world):
    return bool(((world["shapes"]["enlarged"] != 0) & (world["shapes"]["type"] != wire.SHAPE_FREE)).any())


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


def rain_world(seed, count=120, spin=True):
    """A fuzz world for the whole loop: `count` bodies of mixed shape types (circles, capsules, boxes, regular
    polygons with and without rounding, a few two-shape bodies) falling into a static trough made of a box, two
    segments and a kinematic paddle; no contacts yet, every proxy in the move buffer, plenty of free contact slots.
    Geometry and mass are plausible rather than exact (both sides of a comparison get the same arrays)."""
    from solver2d_amd import synthetic
    rng = np.random.default_rng(seed)
    margin = np.float32(synthetic.AABB_MARGIN)
    nb = count + 3
    bodies = np.zeros(nb, dtype=wire.body_dtype)
    shape_list = []

    def add_shape(body, kind, verts, radius, normals=None):
        sh = np.zeros(1, dtype=wire.shape_dtype)[0]
        sh["body"], sh["type"] = body, kind
        sh["categoryBits"], sh["maskBits"], sh["groupIndex"] = 1, 0xFFFFFFFF, 0
        sh["proxyKey"] = (len(shape_list) << 4) | int(bodies[body]["type"])
        sh["count"] = len(verts)
        sh["radius"] = radius
        sh["vertices"][:len(verts)] = verts
        if normals is not None:
            sh["normals"][:len(normals)] = normals
        q = bodies[body]["rot"]
        p = bodies[body]["position"]
        world = np.array([(q[1] * v[0] - q[0] * v[1] + p[0], q[0] * v[0] + q[1] * v[1] + p[1]) for v in verts], dtype=np.float32)
        lo = world.min(axis=0) - np.float32(radius)
        hi = world.max(axis=0) + np.float32(radius)
        sh["aabb"] = (lo[0], lo[1], hi[0], hi[1])
        sh["fatAABB"] = (lo[0] - margin, lo[1] - margin, hi[0] + margin, hi[1] + margin)
        sh["enlarged"] = 1
        shape_list.append(sh)

    def ngon(n, r, phase=0.0):
        ang = phase + 2.0 * np.pi * np.arange(n) / n
        verts = np.stack([r * np.cos(ang), r * np.sin(ang)], axis=1).astype(np.float32)
        mid = ang + np.pi / n
        normals = np.stack([np.cos(mid), np.sin(mid)], axis=1).astype(np.float32)
        return verts, normals

    # the trough
    synthetic._static_body(bodies[0], 0.0, -1.0)
    v, n = ngon(4, 1.0)
    hx, hy = 14.0, 1.0
    add_shape(0, wire.SHAPE_POLYGON, [(-hx, -hy), (hx, -hy), (hx, hy), (-hx, hy)], 0.0, [(0, -1), (1, 0), (0, 1), (-1, 0)])
    synthetic._static_body(bodies[1], 0.0, 0.0)
    add_shape(1, wire.SHAPE_SEGMENT, [(-12.0, 0.0), (-13.5, 9.0)], 0.0)
    add_shape(1, wire.SHAPE_SEGMENT, [(12.0, 0.0), (13.5, 9.0)], 0.0)
    # a kinematic paddle sweeping the floor
    synthetic._static_body(bodies[2], -6.0, 1.2)
    bodies[2]["type"] = wire.BODY_KINEMATIC
    bodies[2]["linearVelocity"] = (1.5, 0.0)
    bodies[2]["angularVelocity"] = 1.0 if spin else 0.0
    add_shape(2, wire.SHAPE_CAPSULE, [(-1.2, 0.0), (1.2, 0.0)], 0.25)

    cols = 12
    for k in range(count):
        i = 3 + k
        x = -8.0 + 1.45 * (k % cols) + rng.uniform(-0.15, 0.15)
        y = 2.0 + 1.3 * (k // cols) + rng.uniform(-0.1, 0.1)
        r = np.float32(rng.uniform(0.3, 0.55))
        mass = np.float32(np.pi * r * r)
        synthetic._dynamic_body(bodies[i], x, y, mass, np.float32(0.5 * mass * r * r))
        a = rng.uniform(-np.pi, np.pi)
        bodies[i]["rot"] = (np.sin(a), np.cos(a))
        bodies[i]["linearVelocity"] = rng.uniform(-1.0, 1.0, 2)
        bodies[i]["angularVelocity"] = rng.uniform(-2.0, 2.0)
        kind = rng.integers(0, 6)
        if kind == 0:
            add_shape(i, wire.SHAPE_CIRCLE, [(0.0, 0.0)], r)
        elif kind == 1:
            add_shape(i, wire.SHAPE_CAPSULE, [(-0.6 * r, 0.0), (0.6 * r, 0.0)], 0.6 * r)
        elif kind == 2:
            add_shape(i, wire.SHAPE_POLYGON, [(-r, -0.7 * r), (r, -0.7 * r), (r, 0.7 * r), (-r, 0.7 * r)], 0.0,
                      [(0, -1), (1, 0), (0, 1), (-1, 0)])
        elif kind == 3:
            v, n = ngon(int(rng.integers(3, 9)), r, rng.uniform(0, 1))
            add_shape(i, wire.SHAPE_POLYGON, v, 0.0, n)
        elif kind == 4:
            v, n = ngon(int(rng.integers(3, 6)), 0.7 * r)
            add_shape(i, wire.SHAPE_POLYGON, v, 0.25 * r, n)
        else:  # two shapes on one body: a dumbbell
            add_shape(i, wire.SHAPE_CIRCLE, [(-0.5 * r, 0.0)], 0.5 * r)
            add_shape(i, wire.SHAPE_CIRCLE, [(0.5 * r, 0.0)], 0.5 * r)
    shapes = np.array(shape_list, dtype=wire.shape_dtype)
    slots = 8 * count
    contacts = np.zeros(slots, dtype=wire.contact_dtype)
    contacts["constraintIndex"] = -1
    pairs = np.zeros(slots, dtype=wire.pair_state_dtype)
    pairs["shapeA"] = -1
    pairs["shapeB"] = -1
    q, p, lc = bodies["rot"], bodies["position"], bodies["localCenter"]
    origins = np.ascontiguousarray(p, dtype=np.float32).copy()
    return {"bodies": bodies, "contacts": contacts, "joints": np.zeros(0, dtype=wire.joint_dtype), "shapes": shapes,
            "pairs": pairs, "origins": origins}


def wreck_world(seed, base):
    """A pyramid world with one to three heavy balls flying into it and spare contact slots: bursts of created and
    destroyed contacts between stretches where the pile is quiet enough for the strip paths."""
    from solver2d_amd import synthetic
    rng = np.random.default_rng(seed)
    w = synthetic.pyramid_world(base)
    nb = len(w["bodies"])
    balls = int(rng.integers(1, 4))
    bodies = np.concatenate([w["bodies"], np.zeros(balls, dtype=wire.body_dtype)])
    shapes = np.concatenate([w["shapes"], np.zeros(balls, dtype=wire.shape_dtype)])
    origins = np.concatenate([w["origins"], np.zeros((balls, 2), dtype=np.float32)])
    for k in range(balls):
        i = nb + k
        r = np.float32(rng.uniform(0.6, 1.5))
        side = -1.0 if rng.random() < 0.5 else 1.0
        x = side * (0.5 * base + 6.0 + 4.0 * k)
        y = rng.uniform(2.0, 0.6 * base)
        mass = np.float32(20.0 * np.pi * r * r)
        synthetic._dynamic_body(bodies[i], x, y, mass, np.float32(0.5 * mass * r * r))
        bodies[i]["linearVelocity"] = (-side * rng.uniform(15.0, 40.0), rng.uniform(0.0, 6.0))
        sh = shapes[i]
        sh["body"], sh["type"] = i, wire.SHAPE_CIRCLE
        sh["categoryBits"], sh["maskBits"] = 1, 0xFFFFFFFF
        sh["proxyKey"] = (i << 4) | wire.BODY_DYNAMIC
        sh["radius"] = r
        m = np.float32(synthetic.AABB_MARGIN)
        sh["aabb"] = (x - r, y - r, x + r, y + r)
        sh["fatAABB"] = (x - r - m, y - r - m, x + r + m, y + r + m)
        sh["enlarged"] = 1
        origins[i] = (x, y)
    extra = 4 * len(w["contacts"]) // 3 + 256
    contacts = np.concatenate([w["contacts"], np.zeros(extra, dtype=wire.contact_dtype)])
    contacts["constraintIndex"][len(w["contacts"]):] = -1
    pairs = np.concatenate([w["pairs"], np.zeros(extra, dtype=wire.pair_state_dtype)])
    pairs["shapeA"][len(w["pairs"]):] = -1
    pairs["shapeB"][len(w["pairs"]):] = -1
    return {"bodies": bodies, "contacts": contacts, "joints": w["joints"], "shapes": shapes, "pairs": pairs, "origins": origins}
