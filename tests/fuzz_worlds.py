"""Random wire-format worlds for parity fuzzing: arbitrary (not physically tidy) solver inputs that
hit branches the scene corpus rarely reaches -- speculative and deep points mixed in one manifold,
ill-conditioned block K, limits at both ends, motors, mouse joints, kinematic and massless bodies,
rotated static bodies whose rot is NOT a fixed point of the normalisation, free slots everywhere."""
import numpy as np

from solver2d_amd import wire


def random_world(seed, n_bodies=40, n_contacts=80, n_joints=12):
    rng = np.random.default_rng(seed)
    f32 = np.float32
    b = np.zeros(n_bodies, dtype=wire.body_dtype)
    kinds = rng.choice([wire.BODY_STATIC, wire.BODY_KINEMATIC, wire.BODY_DYNAMIC, wire.BODY_DYNAMIC, wire.BODY_DYNAMIC,
                        wire.BODY_DYNAMIC, wire.BODY_FREE], size=n_bodies)
    kinds[0] = wire.BODY_STATIC
    kinds[1] = wire.BODY_DYNAMIC
    for i in range(n_bodies):
        t = int(kinds[i])
        b[i]["type"] = t
        if t == wire.BODY_FREE:
            continue
        b[i]["position"] = rng.uniform(-10, 10, 2)
        ang = f32(rng.uniform(-3, 3)) if rng.random() < 0.8 else f32(0.0)
        b[i]["rot"] = (np.sin(ang, dtype=f32), np.cos(ang, dtype=f32))
        b[i]["gravityScale"] = rng.choice([1.0, 1.0, 0.0, 2.0])
        if t != wire.BODY_STATIC:
            b[i]["linearVelocity"] = rng.uniform(-3, 3, 2)
            b[i]["angularVelocity"] = rng.uniform(-2, 2)
        if t == wire.BODY_DYNAMIC:
            massless = rng.random() < 0.05
            m = f32(0.0) if massless else f32(rng.uniform(0.1, 20.0))
            inertia = f32(0.0) if (massless or rng.random() < 0.05) else f32(rng.uniform(0.01, 5.0))
            b[i]["mass"] = m
            b[i]["invMass"] = f32(0.0) if m == 0 else f32(1.0) / m
            b[i]["I"] = inertia
            b[i]["invI"] = f32(0.0) if inertia == 0 else f32(1.0) / inertia
            b[i]["localCenter"] = rng.uniform(-0.2, 0.2, 2) if rng.random() < 0.5 else (0.0, 0.0)
            b[i]["force"] = rng.uniform(-5, 5, 2) if rng.random() < 0.3 else (0.0, 0.0)
            b[i]["torque"] = rng.uniform(-1, 1) if rng.random() < 0.3 else 0.0
            b[i]["linearDamping"] = rng.choice([0.0, 0.0, 0.1, 1.0])
            b[i]["angularDamping"] = rng.choice([0.0, 0.0, 0.05])
    live = np.flatnonzero(b["type"] >= 0)
    # the reference never pairs two immovable bodies (src/body.c s2ShouldBodiesCollide needs a dynamic
    # body) and divides by kA + kB in XPBD: every constraint gets at least one body with real mass
    massive = np.flatnonzero((b["type"] == wire.BODY_DYNAMIC) & (b["invMass"] > 0) & (b["invI"] > 0))

    c = np.zeros(n_contacts, dtype=wire.contact_dtype)
    c["constraintIndex"] = -1
    for i in range(n_contacts):
        if rng.random() < 0.15:
            c[i]["bodyA"] = c[i]["bodyB"] = -1  # free slot
            continue
        a = rng.choice(massive)
        bb = rng.choice(live[live != a])
        if rng.random() < 0.5:
            a, bb = bb, a
        c[i]["bodyA"], c[i]["bodyB"] = int(a), int(bb)
        pc = int(rng.choice([0, 1, 2, 2, 2]))
        c[i]["pointCount"] = pc
        ang = rng.uniform(-np.pi, np.pi)
        c[i]["normal"] = (np.cos(ang), np.sin(ang))
        c[i]["friction"] = rng.choice([0.0, 0.3, 0.6, 1.5])
        c[i]["frictionPersisted"] = int(rng.random() < 0.5)
        for j in range(2):
            p = c[i]["points"][j]
            p["localAnchorA"] = rng.uniform(-1, 1, 2)
            # redundant points (ill-conditioned block K) now and then
            if j == 1 and rng.random() < 0.2:
                p["localAnchorA"] = c[i]["points"][0]["localAnchorA"]
                p["localAnchorB"] = c[i]["points"][0]["localAnchorB"]
            else:
                p["localAnchorB"] = rng.uniform(-1, 1, 2)
            p["separation"] = rng.choice([-0.2, -0.02, -0.004, 0.0, 0.003, 0.015])
            p["normalImpulse"] = rng.choice([0.0, 0.0, 0.2, 1.5])
            p["tangentImpulse"] = rng.uniform(-0.3, 0.3) if p["normalImpulse"] > 0 else 0.0
            p["frictionAnchorA"] = p["localAnchorA"] + rng.uniform(-0.01, 0.01, 2)
            p["frictionAnchorB"] = p["localAnchorB"] + rng.uniform(-0.01, 0.01, 2)
            fa = ang + rng.uniform(-0.3, 0.3)
            p["frictionNormalA"] = (np.cos(fa), np.sin(fa))
            p["frictionNormalB"] = (np.cos(fa), np.sin(fa))

    jn = np.zeros(n_joints, dtype=wire.joint_dtype)
    for i in range(n_joints):
        r = rng.random()
        if r < 0.15:
            jn[i]["type"] = wire.JOINT_FREE
            jn[i]["bodyA"] = jn[i]["bodyB"] = -1
            continue
        bb = rng.choice(massive)
        a = rng.choice(live[live != bb])
        jn[i]["bodyA"], jn[i]["bodyB"] = int(a), int(bb)
        jn[i]["localOriginAnchorA"] = rng.uniform(-1, 1, 2)
        jn[i]["localOriginAnchorB"] = rng.uniform(-1, 1, 2)
        jn[i]["impulse"] = rng.uniform(-0.5, 0.5, 2)
        jn[i]["motorImpulse"] = rng.uniform(-0.2, 0.2)
        if r < 0.3:
            jn[i]["type"] = wire.JOINT_MOUSE
            jn[i]["hertz"] = rng.choice([1.0, 5.0, 15.0])
            jn[i]["dampingRatio"] = rng.choice([0.3, 0.7, 1.0])
            jn[i]["targetA"] = rng.uniform(-10, 10, 2)
        else:
            jn[i]["type"] = wire.JOINT_REVOLUTE
            jn[i]["enableMotor"] = int(rng.random() < 0.4)
            jn[i]["enableLimit"] = int(rng.random() < 0.6)
            jn[i]["maxMotorTorque"] = rng.choice([0.0, 5.0, 100.0])
            jn[i]["motorSpeed"] = rng.uniform(-3, 3)
            jn[i]["referenceAngle"] = rng.uniform(-1, 1)
            lo = rng.uniform(-1.5, 0.5)
            jn[i]["lowerAngle"] = lo
            jn[i]["upperAngle"] = lo + rng.choice([0.0, 0.01, 0.5, 2.0])
            jn[i]["lowerImpulse"] = rng.choice([0.0, 0.1])
            jn[i]["upperImpulse"] = rng.choice([0.0, 0.1])
    return b, c, jn
