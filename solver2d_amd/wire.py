"""numpy / ctypes mirrors of the C-ABI structs in include/solver2d_amd.h.

The dtypes are laid out exactly as the C compiler lays out the structs (all members are 4-byte
scalars, so there is no padding); `tests/test_abi.py` checks the sizes against the built
library.  Arrays of these dtypes are what the Python host side hands to the C entry points.
"""
import ctypes
import numpy as np

API_VERSION = 5

SOLVER_NAMES = [
    "Jacobi", "PGS", "PGS_NGS", "PGS_NGS_Block", "PGS_Soft",
    "SoftStep", "TGS_Sticky", "TGS_Soft", "TGS_NGS", "XPBD",
]
SOLVER_ID = {n: i for i, n in enumerate(SOLVER_NAMES)}

BODY_FREE, BODY_STATIC, BODY_KINEMATIC, BODY_DYNAMIC = -1, 0, 1, 2
JOINT_FREE, JOINT_REVOLUTE, JOINT_MOUSE = -1, 0, 1

f32, i32 = np.float32, np.int32

body_dtype = np.dtype([
    ("position", f32, 2), ("rot", f32, 2), ("linearVelocity", f32, 2), ("angularVelocity", f32),
    ("deltaPosition", f32, 2), ("localCenter", f32, 2), ("force", f32, 2), ("torque", f32),
    ("mass", f32), ("invMass", f32), ("I", f32), ("invI", f32),
    ("linearDamping", f32), ("angularDamping", f32), ("gravityScale", f32), ("type", i32),
])

manifold_point_dtype = np.dtype([
    ("localAnchorA", f32, 2), ("localAnchorB", f32, 2),
    ("frictionAnchorA", f32, 2), ("frictionAnchorB", f32, 2),
    ("frictionNormalA", f32, 2), ("frictionNormalB", f32, 2),
    ("separation", f32), ("normalImpulse", f32), ("tangentImpulse", f32),
])

contact_dtype = np.dtype([
    ("bodyA", i32), ("bodyB", i32), ("pointCount", i32), ("frictionPersisted", i32),
    ("normal", f32, 2), ("friction", f32), ("constraintIndex", i32),
    ("points", manifold_point_dtype, 2),
])

joint_dtype = np.dtype([
    ("type", i32), ("bodyA", i32), ("bodyB", i32), ("enableMotor", i32), ("enableLimit", i32),
    ("localOriginAnchorA", f32, 2), ("localOriginAnchorB", f32, 2),
    ("impulse", f32, 2), ("motorImpulse", f32), ("lowerImpulse", f32), ("upperImpulse", f32),
    ("maxMotorTorque", f32), ("motorSpeed", f32), ("referenceAngle", f32),
    ("lowerAngle", f32), ("upperAngle", f32),
    ("hertz", f32), ("dampingRatio", f32), ("targetA", f32, 2),
])

u32 = np.uint32
shape_dtype = np.dtype([
    ("body", i32), ("type", i32), ("categoryBits", u32), ("maskBits", u32), ("groupIndex", i32),
    ("proxyKey", i32), ("enlarged", i32), ("count", i32), ("radius", f32),
    ("aabb", f32, 4), ("fatAABB", f32, 4), ("vertices", f32, (8, 2)), ("normals", f32, (8, 2)),
])
SHAPE_FREE, SHAPE_CAPSULE, SHAPE_CIRCLE, SHAPE_POLYGON, SHAPE_SEGMENT = -1, 0, 1, 2, 3
SHAPE_SIZE = shape_dtype.itemsize
assert SHAPE_SIZE == 196

u16, u8 = np.uint16, np.uint8
pair_state_dtype = np.dtype([
    ("shapeA", i32), ("shapeB", i32), ("cacheMetric", f32), ("cacheCount", u16), ("id", u16, 2),
    ("cacheIndexA", u8, 3), ("cacheIndexB", u8, 3), ("persisted", u8, 2), ("pad", u8, 2),
])
PAIR_STATE_SIZE = pair_state_dtype.itemsize
assert PAIR_STATE_SIZE == 28
PAIR_UPDATED, PAIR_SEPARATED, PAIR_FREE = 0, 1, -1

BODY_SIZE, CONTACT_SIZE, JOINT_SIZE = body_dtype.itemsize, contact_dtype.itemsize, joint_dtype.itemsize
assert BODY_SIZE == 88 and manifold_point_dtype.itemsize == 60 and CONTACT_SIZE == 152 and JOINT_SIZE == 92


class StepParams(ctypes.Structure):
    _fields_ = [
        ("solverType", ctypes.c_int32), ("dt", ctypes.c_float), ("velIters", ctypes.c_int32),
        ("posIters", ctypes.c_int32), ("warmStart", ctypes.c_int32), ("gravity", ctypes.c_float * 2),
    ]

    @classmethod
    def make(cls, solver, dt=1.0 / 60.0, vel_iters=4, pos_iters=2, warm_start=True, gravity=(0.0, -10.0)):
        sid = SOLVER_ID[solver] if isinstance(solver, str) else int(solver)
        return cls(sid, np.float32(dt), int(vel_iters), int(pos_iters), 1 if warm_start else 0,
                   (ctypes.c_float * 2)(*gravity))

    def as_dict(self):
        return dict(solverType=int(self.solverType), dt=float(self.dt), velIters=int(self.velIters),
                    posIters=int(self.posIters), warmStart=int(self.warmStart),
                    gravity=[float(self.gravity[0]), float(self.gravity[1])])


class StepStats(ctypes.Structure):
    _fields_ = [
        ("constraintCount", ctypes.c_int32), ("jointCount", ctypes.c_int32),
        ("contactColors", ctypes.c_int32), ("jointColors", ctypes.c_int32),
        ("solveSweeps", ctypes.c_int32), ("kernelLaunches", ctypes.c_int32),
        ("deviceMs", ctypes.c_float), ("solveKernelMs", ctypes.c_float), ("hostPrepMs", ctypes.c_float),
        ("graphReplayed", ctypes.c_int32), ("solveLaunches", ctypes.c_int32),
        ("eventPairOverheadMs", ctypes.c_float), ("groupCount", ctypes.c_int32), ("messagePassing", ctypes.c_int32),
        ("stripCount", ctypes.c_int32), ("seamCount", ctypes.c_int32), ("persistent", ctypes.c_int32), ("persistFallbacks", ctypes.c_int32),
        ("structureBuilds", ctypes.c_int32), ("placedContacts", ctypes.c_int32), ("potentialConstraints", ctypes.c_int32), ("pairLanes", ctypes.c_int32),
        ("asyncBuildsRequested", ctypes.c_int32), ("asyncBuildsAdopted", ctypes.c_int32), ("asyncWaitMs", ctypes.c_float),
        ("bodiesAdopted", ctypes.c_int32), ("seamBodiesAdded", ctypes.c_int32), ("roundsOpened", ctypes.c_int32),
        ("overflowContacts", ctypes.c_int32), ("slicedStep", ctypes.c_int32), ("slicedSteps", ctypes.c_int32), ("nearHandoffTimeouts", ctypes.c_int32),
    ]


class WorldStepInfo(ctypes.Structure):
    """s2amdWorldStepInfo"""
    _fields_ = [
        ("separatedCount", ctypes.c_int32), ("activeContacts", ctypes.c_int32), ("graphChanged", ctypes.c_int32),
        ("movedCount", ctypes.c_int32), ("contactsMs", ctypes.c_float), ("solveMs", ctypes.c_float), ("stepMs", ctypes.c_float),
    ]


def solve_sweeps_per_step(solver, vel_iters, pos_iters):
    """Full passes of a s2SolveContacts_* function per s2World_Step, per reference driver.

    TGS_Soft / SoftStep: velIters * (1 + [posIters > 0])   (solve_tgs_soft.c:211-269)
    Jacobi / PGS_Soft:   velIters + posIters                 (solve_jacobi.c:218,255)
    PGS:                 velIters                            (solve_pgs.c:186-199)
    PGS_NGS / Block:     velIters velocity + posIters position sweeps
    TGS_NGS:             velIters * 2 (velocity + NGS position per sub-step)
    TGS_Sticky:          velIters + posIters
    XPBD:                velIters * 2 (position + velocity relax per sub-step)
    """
    name = solver if isinstance(solver, str) else SOLVER_NAMES[int(solver)]
    if name in ("TGS_Soft", "SoftStep"):
        return vel_iters * (2 if pos_iters > 0 else 1)
    if name in ("Jacobi", "PGS_Soft", "TGS_Sticky", "PGS_NGS", "PGS_NGS_Block"):
        return vel_iters + pos_iters
    if name == "PGS":
        return vel_iters
    if name in ("TGS_NGS", "XPBD"):
        return 2 * vel_iters
    raise ValueError(name)


def as_ptr(arr, ctype=ctypes.c_void_p):
    if arr is None or len(arr) == 0:
        return ctypes.c_void_p(0)
    assert arr.flags["C_CONTIGUOUS"]
    return ctypes.c_void_p(arr.ctypes.data)


# s2amdShapeBox (include/solver2d_amd.h): what stage 4 changed in a shape
# s2TreeNode byte for byte (include/solver2d/dynamic_tree.h:14-41; s2amdTreeNode)
tree_node_dtype = np.dtype([("aabb", np.float32, 4), ("categoryBits", np.uint32), ("parent", np.int32), ("child1", np.int32), ("child2", np.int32),
                            ("userData", np.int32), ("height", np.int16), ("enlarged", np.uint8), ("pad", np.uint8, 9)])
assert tree_node_dtype.itemsize == 48
shape_box_dtype = np.dtype([("aabb", np.float32, 4), ("fatAABB", np.float32, 4), ("enlarged", np.int32)])
assert shape_box_dtype.itemsize == 36
