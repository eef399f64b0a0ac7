"""Synthetic solver inputs in wire format, generated without any collision code.

`pyramid(base, count)` is the state the reference hands to s2Solve_* on the FIRST s2World_Step of
the Pyramid / LargePyramid scene (recipe: reference samples/collection/sample_contact.cpp:511-552,
SURVEY.md 8d): unit boxes at rest, every touching pair already has its 2-point manifold with
zero separation and zero impulses.  tests/test_synthetic.py checks it against the reference's
captured step-0 state (same bodies, same set of manifolds; the contact ORDER is this module's own).
`joint_grid(n)` is the JointGrid scene (sample_joints.cpp:377-446): circles that never collide,
held together by revolute joints.
"""
import numpy as np

from . import wire

# s2ComputePolygonMass of s2MakeSquare(0.5), density 1 (reference geometry.c) as fp32
BOX_MASS = np.float32(1.0)
BOX_I = np.float32(0.16666669)
# circle r = 0.4, density 1: mass = pi r^2, I = m (0.5 r^2 + |center|^2) (reference geometry.c)
CIRCLE_R = np.float32(0.4)


def _dynamic_body(b, x, y, mass, inertia, gravity_scale=1.0):
    b["position"] = (x, y)
    b["rot"] = (0.0, 1.0)
    b["mass"] = mass
    b["invMass"] = np.float32(1.0) / np.float32(mass)
    b["I"] = inertia
    b["invI"] = np.float32(1.0) / np.float32(inertia)
    b["gravityScale"] = gravity_scale
    b["type"] = wire.BODY_DYNAMIC


def _static_body(b, x, y):
    b["position"] = (x, y)
    b["rot"] = (0.0, 1.0)
    b["gravityScale"] = 1.0
    b["type"] = wire.BODY_STATIC


def _manifold(c, a, b, normal, pts, friction=0.6):
    c["bodyA"], c["bodyB"] = a, b
    c["pointCount"] = len(pts)
    c["normal"] = normal
    c["friction"] = friction
    c["constraintIndex"] = -1
    for j, (la, lb) in enumerate(pts):
        c["points"][j]["localAnchorA"] = la
        c["points"][j]["localAnchorB"] = lb


def pyramid(base, count=1, pitch=None):
    """`count` disjoint box pyramids of `base` bricks at the bottom, each on its own static ground.
    Returns (bodies, contacts, joints)."""
    if count > 1:
        # one pyramid, then copies shifted onto a 32-wide lattice (fp32 additions, as a scene builder would do them)
        b0, c0, joints = pyramid(base, 1, pitch)
        if pitch is None:
            pitch = float(base + 20)
        k = np.arange(count)
        ox = ((k % 32) * pitch).astype(np.float32)
        oy = ((k // 32) * pitch).astype(np.float32)
        bodies = np.tile(b0, count)
        contacts = np.tile(c0, count)
        bodies["position"][:, 0] = (np.repeat(ox, len(b0)) + np.tile(b0["position"][:, 0], count)).astype(np.float32)
        bodies["position"][:, 1] = (np.repeat(oy, len(b0)) + np.tile(b0["position"][:, 1], count)).astype(np.float32)
        shift = np.repeat(k * len(b0), len(c0)).astype(np.int32)
        contacts["bodyA"] += shift
        contacts["bodyB"] += shift
        return bodies, contacts, joints
    return _pyramid_loops(base, count, pitch)


def _pyramid_loops(base, count=1, pitch=None):
    per = base * (base + 1) // 2
    nb = count * (per + 1)
    ncon = count * (base + (per - base) + 2 * (per - base))  # ground + side-by-side + stacking
    bodies = np.zeros(nb, dtype=wire.body_dtype)
    contacts = np.zeros(ncon, dtype=wire.contact_dtype)
    joints = np.zeros(0, dtype=wire.joint_dtype)
    if pitch is None:
        pitch = float(base + 20)
    h = np.float32(0.5)
    ci = 0
    for k in range(count):
        ox = np.float32((k % 32) * pitch)
        oy = np.float32((k // 32) * pitch)
        g = k * (per + 1)
        _static_body(bodies[g], ox, oy - np.float32(1.0))
        index = {}
        bi = g + 1
        for i in range(base):
            y = np.float32(np.float32(2.0 * i + 1.0) * h)
            for j in range(i, base):
                x = np.float32(np.float32(np.float32(i + 1.0) * h + np.float32(2.0 * (j - i)) * h) - h * np.float32(base))
                _dynamic_body(bodies[bi], ox + x, oy + y, BOX_MASS, BOX_I)
                index[(i, j)] = bi
                if i == 0:
                    # ground (A) - box (B); anchors relative to body origins, reference step-0 values
                    gx = float(x)
                    _manifold(contacts[ci], g, bi, (0.0, 1.0),
                              [((gx + 0.5, 1.0), (0.5, -0.5)), ((gx - 0.5, 1.0), (-0.5, -0.5))])
                    ci += 1
                else:
                    lo_right = index[(i - 1, j)]      # lower box half a unit to the right
                    lo_left = index[(i - 1, j - 1)]   # lower box half a unit to the left
                    _manifold(contacts[ci], lo_right, bi, (0.0, 1.0),
                              [((0.0, 0.5), (0.5, -0.5)), ((-0.5, 0.5), (0.0, -0.5))])
                    ci += 1
                    _manifold(contacts[ci], lo_left, bi, (0.0, 1.0),
                              [((0.5, 0.5), (0.0, -0.5)), ((0.0, 0.5), (-0.5, -0.5))])
                    ci += 1
                if j > i:
                    _manifold(contacts[ci], index[(i, j - 1)], bi, (1.0, 0.0),
                              [((0.5, -0.5), (-0.5, -0.5)), ((0.5, 0.5), (-0.5, 0.5))])
                    ci += 1
                bi += 1
    assert ci == ncon
    return bodies, contacts, joints


SPECULATIVE_DISTANCE = np.float32(0.02)  # 4 * s2_linearSlop (reference constants.h:10-12)
AABB_MARGIN = np.float32(0.1)            # s2_aabbMargin (reference constants.h:22)


def _box_shape(sh, body, body_type, hx, hy, px, py, index):
    """s2MakeBox(hx, hy) attached to an unrotated body at (px, py): vertices, normals (reference geometry.c: s2MakeBox),
    tight AABB + speculative margin and the fat AABB of s2CreateShape."""
    hx, hy = np.float32(hx), np.float32(hy)
    sh["body"] = body
    sh["type"] = wire.SHAPE_POLYGON
    sh["categoryBits"], sh["maskBits"], sh["groupIndex"] = 1, 0xFFFFFFFF, 0
    sh["proxyKey"] = (index << 4) | int(body_type)
    sh["count"] = 4
    sh["radius"] = 0.0
    sh["vertices"][:4] = [(-hx, -hy), (hx, -hy), (hx, hy), (-hx, hy)]
    sh["normals"][:4] = [(0.0, -1.0), (1.0, 0.0), (0.0, 1.0), (-1.0, 0.0)]
    lo = (np.float32(px) - hx, np.float32(py) - hy)
    hi = (np.float32(px) + hx, np.float32(py) + hy)
    sh["aabb"] = (lo[0] - SPECULATIVE_DISTANCE, lo[1] - SPECULATIVE_DISTANCE, hi[0] + SPECULATIVE_DISTANCE, hi[1] + SPECULATIVE_DISTANCE)
    sh["fatAABB"] = (lo[0] - AABB_MARGIN, lo[1] - AABB_MARGIN, hi[0] + AABB_MARGIN, hi[1] + AABB_MARGIN)


def pyramid_world(base, count=1, pitch=None):
    """pyramid() plus what stages 3 and 4 of s2World_Step read: one box shape per body (shape index == body index), the
    narrow-phase pair state of every contact slot (empty GJK cache, no feature ids yet) and the body origins.
    Returns a dict with the keys of tests/world_chain.py: bodies, contacts, joints, shapes, pairs, origins."""
    bodies, contacts, joints = pyramid(base, count, pitch)
    shapes = np.zeros(len(bodies), dtype=wire.shape_dtype)
    for i, b in enumerate(bodies):
        if b["type"] == wire.BODY_STATIC:
            # the scene's 100 m ground (widened for base 200); many pyramids: grounds that stay clear of each other
            ground_hx = max(100.0, float(base)) if count == 1 else 0.5 * base + 5.0
            _box_shape(shapes[i], i, b["type"], ground_hx, 1.0, b["position"][0], b["position"][1], i)
        else:
            _box_shape(shapes[i], i, b["type"], 0.5, 0.5, b["position"][0], b["position"][1], i)
    pairs = np.zeros(len(contacts), dtype=wire.pair_state_dtype)
    pairs["shapeA"] = contacts["bodyA"]
    pairs["shapeB"] = contacts["bodyB"]
    origins = np.ascontiguousarray(bodies["position"], dtype=np.float32).copy()  # localCenter is zero for a centred box
    return {"bodies": bodies, "contacts": contacts, "joints": joints, "shapes": shapes, "pairs": pairs, "origins": origins}


def joint_grid(numi, numk=None):
    """numi x numk circles on a unit lattice pinned by revolute joints (no contacts)."""
    numk = numi if numk is None else numk
    nb = numi * numk
    bodies = np.zeros(nb, dtype=wire.body_dtype)
    contacts = np.zeros(0, dtype=wire.contact_dtype)
    jl = []
    mass = np.float32(np.float32(np.pi) * CIRCLE_R * CIRCLE_R)
    inertia = np.float32(mass * np.float32(np.float32(0.5) * CIRCLE_R * CIRCLE_R))
    idx = 0
    for k in range(numk):
        for i in range(numi):
            b = bodies[idx]
            if numk // 2 - 3 <= k <= numk // 2 + 3 and i == 0:
                _static_body(b, float(k), float(-i))
                b["gravityScale"] = 2.0
            else:
                _dynamic_body(b, float(k), float(-i), mass, inertia, 2.0)
            if i > 0:
                jl.append((idx - 1, idx, (0.0, -0.5), (0.0, 0.5)))
            if k > 0:
                jl.append((idx - numi, idx, (0.5, 0.0), (-0.5, 0.0)))
            idx += 1
    joints = np.zeros(len(jl), dtype=wire.joint_dtype)
    for n, (a, b, la, lb) in enumerate(jl):
        j = joints[n]
        j["type"] = wire.JOINT_REVOLUTE
        j["bodyA"], j["bodyB"] = a, b
        j["localOriginAnchorA"] = la
        j["localOriginAnchorB"] = lb
    return bodies, contacts, joints


def platform(n, layers=1):
    """A wide DYNAMIC platform resting on static ground with n unit boxes side by side on top
    (and `layers` rows of them): the platform touches n boxes, so it forces n colours -- the
    high-degree-body case (tumbler drum, a crate full of parts) that exercises the sequential tail."""
    nb = 2 + n * layers
    bodies = np.zeros(nb, dtype=wire.body_dtype)
    contacts = np.zeros(1 + n + (n - 1) * layers + n * (layers - 1), dtype=wire.contact_dtype)
    joints = np.zeros(0, dtype=wire.joint_dtype)
    half = np.float32(0.5 * n)
    _static_body(bodies[0], 0.0, -1.0)
    pm = np.float32(4.0 * half)  # 2*half x 1 box of density 2
    pi = np.float32(pm * (4.0 * half * half + 1.0) / 12.0)
    _dynamic_body(bodies[1], 0.0, 0.5, pm, pi)
    ci = 0
    _manifold(contacts[ci], 0, 1, (0.0, 1.0), [((float(half), 1.0), (float(half), -0.5)), ((-float(half), 1.0), (-float(half), -0.5))])
    ci += 1
    idx = {}
    bi = 2
    for r in range(layers):
        for k in range(n):
            x = np.float32(k + 0.5) - half
            y = np.float32(1.5 + r)
            _dynamic_body(bodies[bi], x, y, BOX_MASS, BOX_I)
            idx[(r, k)] = bi
            if r == 0:
                _manifold(contacts[ci], 1, bi, (0.0, 1.0),
                          [((float(x) + 0.5, 0.5), (0.5, -0.5)), ((float(x) - 0.5, 0.5), (-0.5, -0.5))])
            else:
                _manifold(contacts[ci], idx[(r - 1, k)], bi, (0.0, 1.0), [((0.5, 0.5), (0.5, -0.5)), ((-0.5, 0.5), (-0.5, -0.5))])
            ci += 1
            if k > 0:
                _manifold(contacts[ci], idx[(r, k - 1)], bi, (1.0, 0.0), [((0.5, -0.5), (-0.5, -0.5)), ((0.5, 0.5), (-0.5, 0.5))])
                ci += 1
            bi += 1
    assert ci == len(contacts)
    return bodies, contacts, joints


def _offset_box(sh, body, body_type, hx, hy, cx, cy, index):
    """s2MakeOffsetBox(hx, hy, (cx, cy), 0) on an unrotated body at the world origin of its own frame: vertices and normals in
    the body frame; the AABBs are filled by the caller (they need the body's position)."""
    sh["body"], sh["type"] = body, wire.SHAPE_POLYGON
    sh["categoryBits"], sh["maskBits"], sh["groupIndex"] = 1, 0xFFFFFFFF, 0
    sh["proxyKey"] = (index << 4) | int(body_type)
    sh["count"], sh["radius"] = 4, 0.0
    sh["vertices"][:4] = [(cx - hx, cy - hy), (cx + hx, cy - hy), (cx + hx, cy + hy), (cx - hx, cy + hy)]
    sh["normals"][:4] = [(0.0, -1.0), (1.0, 0.0), (0.0, 1.0), (-1.0, 0.0)]


def tumbler_world(count, spare_slots_per_box=16):
    """BASELINE config 3's scene as a resident WORLD with no contacts yet: a hollow square drum (four wall shapes on one
    dynamic body) turned by a revolute motor against a static anchor, `count` small boxes on a grid inside (the recipe of
    solver2d_amd/scenes/scenes.c: sceneTumbler).  Every proxy starts in the move buffer, the contact pool is
    empty: the caller runs the whole loop -- pair query, contact creation, s2amd_world_step -- on it.
    Returns the dict of tests/world_chain.py's WORLD_KEYS."""
    side = 1
    while side * side < count:
        side += 1
    a = np.float32(0.125)
    inner = np.float32(side * (2.0 * a) * 1.5 + 1.0)
    half, wall = np.float32(0.5 * inner), np.float32(0.5)
    cy = np.float32(half + 2.0)
    nb = 2 + count
    bodies = np.zeros(nb, dtype=wire.body_dtype)
    shapes = np.zeros(4 + count, dtype=wire.shape_dtype)
    _static_body(bodies[0], 0.0, 0.0)
    # the drum: density 5, four walls (s2ComputePolygonMass of each offset box, summed about the body origin)
    walls = [(wall, half + wall, half + wall, 0.0), (wall, half + wall, -half - wall, 0.0), (half + wall, wall, 0.0, half + wall), (half + wall, wall, 0.0, -half - wall)]
    mass = inertia = np.float32(0.0)
    for i, (hx, hy, ox, oy) in enumerate(walls):
        m = np.float32(5.0 * 4.0 * hx * hy)
        mass += m
        inertia += np.float32(m * (4.0 * hx * hx + 4.0 * hy * hy) / 12.0 + m * (ox * ox + oy * oy))
        _offset_box(shapes[i], 1, wire.BODY_DYNAMIC, hx, hy, ox, oy, i)
    _dynamic_body(bodies[1], 0.0, cy, mass, inertia)
    box_mass = np.float32(4.0 * a * a)
    box_i = np.float32(box_mass * (8.0 * a * a) / 12.0)
    pitch = np.float32(2.0 * a * 1.25)
    x0 = np.float32(-0.5 * (side - 1) * pitch)
    y0 = np.float32(cy - half + a + 0.05)
    k = 0
    for r in range(side):
        for c in range(side):
            if k >= count:
                break
            _dynamic_body(bodies[2 + k], x0 + c * pitch, y0 + r * pitch, box_mass, box_i)
            _offset_box(shapes[4 + k], 2 + k, wire.BODY_DYNAMIC, a, a, 0.0, 0.0, 4 + k)
            k += 1
    # world-space boxes: tight AABB + speculative margin, fat AABB (s2CreateShape), everything in the move buffer
    pos = bodies["position"][shapes["body"]]
    lo = shapes["vertices"][:, :4, :].min(axis=1) + pos
    hi = shapes["vertices"][:, :4, :].max(axis=1) + pos
    shapes["aabb"][:, 0:2], shapes["aabb"][:, 2:4] = lo - SPECULATIVE_DISTANCE, hi + SPECULATIVE_DISTANCE
    shapes["fatAABB"][:, 0:2], shapes["fatAABB"][:, 2:4] = lo - AABB_MARGIN, hi + AABB_MARGIN
    shapes["enlarged"] = 1
    joints = np.zeros(1, dtype=wire.joint_dtype)
    j = joints[0]
    j["type"], j["bodyA"], j["bodyB"] = wire.JOINT_REVOLUTE, 0, 1
    j["localOriginAnchorA"], j["localOriginAnchorB"] = (0.0, float(cy)), (0.0, 0.0)
    j["enableMotor"], j["motorSpeed"], j["maxMotorTorque"] = 1, np.float32(0.05 * np.pi * 4.0), 1e8
    slots = spare_slots_per_box * count + 1024
    contacts = np.zeros(slots, dtype=wire.contact_dtype)
    contacts["constraintIndex"] = -1
    pairs = np.zeros(slots, dtype=wire.pair_state_dtype)
    pairs["shapeA"] = -1
    pairs["shapeB"] = -1
    return {"bodies": bodies, "contacts": contacts, "joints": joints, "shapes": shapes, "pairs": pairs,
            "origins": np.ascontiguousarray(bodies["position"], dtype=np.float32).copy()}
