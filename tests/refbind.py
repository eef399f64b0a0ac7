"""ctypes binding of oracle/_ref/libs2ref.so -- the unmodified reference + capture hooks.

TEST INFRASTRUCTURE.  Only tests/ and tools under oracle/ import this.  The library is built by
`make -C oracle ref` in the build container (needs /root/reference) and travels to the GPU box as
a prebuilt .so; when it is absent the tests that need it skip.
"""
import ctypes
import os
import numpy as np

from solver2d_amd import wire

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libs2ref.so")


class WorldId(ctypes.Structure):
    _fields_ = [("index", ctypes.c_int16), ("revision", ctypes.c_uint16)]


REPLACE_FN = ctypes.CFUNCTYPE(
    ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(wire.StepParams),
    ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_int32)

_lib = None


def available():
    return os.path.exists(REF_SO)


def lib():
    global _lib
    if _lib is None:
        L = ctypes.CDLL(REF_SO)
        L.s2scene_create.restype = WorldId
        L.s2scene_create.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        L.s2scene_count.restype = ctypes.c_int
        L.s2scene_name.restype = ctypes.c_char_p
        L.s2scene_name.argtypes = [ctypes.c_int]
        L.s2ref_step.argtypes = [WorldId, ctypes.c_float, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        L.s2ref_step.restype = None
        L.s2scene_pre_step.argtypes = [WorldId, ctypes.c_int, ctypes.c_float]
        L.s2scene_pre_step.restype = None
        L.s2scene_post_step.argtypes = [WorldId, ctypes.c_float]
        L.s2scene_post_step.restype = None
        L.s2ref_destroy_world.argtypes = [WorldId]
        L.s2ref_destroy_world.restype = None
        L.s2ref_set_mode.argtypes = [ctypes.c_int]
        L.s2ref_set_mode.restype = None
        L.s2ref_set_replace.argtypes = [REPLACE_FN, ctypes.c_void_p]
        L.s2ref_set_replace.restype = None
        L.s2ref_params.restype = ctypes.POINTER(wire.StepParams)
        L.s2ref_snapshot.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 6
        L.s2ref_world_sizes.argtypes = [WorldId] + [ctypes.POINTER(ctypes.c_int32)] * 3
        L.s2ref_pack_world.argtypes = [WorldId, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        L.s2ref_contact_pairs.argtypes = [WorldId, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32]
        L.s2ref_sizeof.restype = ctypes.c_size_t
        L.s2ref_sizeof.argtypes = [ctypes.c_int]
        L.s2ref_shape_capacity.argtypes = [WorldId]
        L.s2ref_pack_shapes.argtypes = [WorldId, ctypes.c_void_p, ctypes.c_void_p]
        L.s2ref_broadphase_capture.argtypes = [ctypes.c_void_p] * 7
        L.s2ref_broadphase_seconds.restype = ctypes.c_double
        L.s2ref_broadphase_seconds.argtypes = [ctypes.c_int]
        L.s2ref_solve_seconds.restype = ctypes.c_double
        L.s2ref_solve_seconds.argtypes = [ctypes.c_int]
        L.s2ref_solve_calls.restype = ctypes.c_long
        _lib = L
    return _lib


def _copy_array(ptr, count, dtype):
    if count <= 0 or not ptr:
        return np.zeros(0, dtype=dtype)
    buf = (ctypes.c_char * (count * dtype.itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype, count=count).copy()


class RefWorld:
    """A world inside the reference library, stepped through the public s2World_Step."""

    def __init__(self, scene, solver, p0=0, p1=0):
        L = lib()
        sid = wire.SOLVER_ID[solver] if isinstance(solver, str) else int(solver)
        self.solver = wire.SOLVER_NAMES[sid]
        self.id = L.s2scene_create(scene.encode(), sid, int(p0), int(p1))
        if self.id.index < 0:
            raise RuntimeError("s2scene_create(%s) failed" % scene)
        self.steps = 0

    def close(self):
        if self.id is not None:
            lib().s2ref_destroy_world(self.id)
            self.id = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def step(self, dt=1.0 / 60.0, vel_iters=4, pos_iters=2, warm_start=True):
        """One frame of the sample as the reference's GUI runs it: what the sample's Step override does before stepping the world
        (Warm Start Energy destroys its top body, Rush applies its forces), s2World_Step, what it does after (Ragdoll Stress)."""
        L = lib()
        L.s2scene_pre_step(self.id, self.steps, ctypes.c_float(dt))
        L.s2ref_step(self.id, ctypes.c_float(dt), vel_iters, pos_iters, 1 if warm_start else 0)
        self.steps += 1
        L.s2scene_post_step(self.id, ctypes.c_float(1.0 / dt if dt > 0 else 0.0))

    def sizes(self):
        nb, nc, nj = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
        lib().s2ref_world_sizes(self.id, ctypes.byref(nb), ctypes.byref(nc), ctypes.byref(nj))
        return nb.value, nc.value, nj.value

    def pack(self):
        nb, nc, nj = self.sizes()
        bodies = np.zeros(nb, dtype=wire.body_dtype)
        contacts = np.zeros(nc, dtype=wire.contact_dtype)
        joints = np.zeros(nj, dtype=wire.joint_dtype)
        lib().s2ref_pack_world(self.id, wire.as_ptr(bodies), wire.as_ptr(contacts), wire.as_ptr(joints))
        return bodies, contacts, joints

    def contact_pairs(self):
        _, nc, _ = self.sizes()
        a = np.zeros(nc, dtype=np.int32)
        b = np.zeros(nc, dtype=np.int32)
        lib().s2ref_contact_pairs(self.id, wire.as_ptr(a), wire.as_ptr(b), nc)
        return a, b

    def pack_shapes(self):
        """(shapes, origins[bodyCapacity, 2]) of the world as it stands."""
        ns = lib().s2ref_shape_capacity(self.id)
        nb, _, _ = self.sizes()
        shapes = np.zeros(ns, dtype=wire.shape_dtype)
        origins = np.zeros((nb, 2), dtype=np.float32)
        lib().s2ref_pack_shapes(self.id, wire.as_ptr(shapes), wire.as_ptr(origins))
        return shapes, origins

    def step_captured(self, dt=1.0 / 60.0, vel_iters=4, pos_iters=2, warm_start=True):
        """Step once with the capture hook armed; returns (params, pre, post) where pre/post are
        (bodies, contacts, joints) wire arrays at solver entry / exit."""
        L = lib()
        L.s2ref_set_mode(1)
        try:
            self.step(dt, vel_iters, pos_iters, warm_start)
        finally:
            L.s2ref_set_mode(0)
        return last_capture()


def last_capture():
    L = lib()
    out = []
    for which in (0, 1):
        pb, pc, pj = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
        nb, nc, nj = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
        L.s2ref_snapshot(which, ctypes.byref(pb), ctypes.byref(nb), ctypes.byref(pc), ctypes.byref(nc),
                         ctypes.byref(pj), ctypes.byref(nj))
        out.append((_copy_array(pb.value, nb.value, wire.body_dtype),
                    _copy_array(pc.value, nc.value, wire.contact_dtype),
                    _copy_array(pj.value, nj.value, wire.joint_dtype)))
    p = L.s2ref_params().contents
    params = wire.StepParams(p.solverType, p.dt, p.velIters, p.posIters, p.warmStart,
                             (ctypes.c_float * 2)(p.gravity[0], p.gravity[1]))
    return params, out[0], out[1]


class Replace:
    """Context manager: route the reference's s2Solve_* through `fn(params, bodies, contacts,
    joints)` which must mutate the numpy wire arrays in place and return 0."""

    def __init__(self, fn):
        def tramp(user, params, pb, nb, pc, nc, pj, nj):
            bodies = np.frombuffer((ctypes.c_char * (nb * wire.BODY_SIZE)).from_address(pb), dtype=wire.body_dtype) if nb else np.zeros(0, wire.body_dtype)
            contacts = np.frombuffer((ctypes.c_char * (nc * wire.CONTACT_SIZE)).from_address(pc), dtype=wire.contact_dtype) if nc else np.zeros(0, wire.contact_dtype)
            joints = np.frombuffer((ctypes.c_char * (nj * wire.JOINT_SIZE)).from_address(pj), dtype=wire.joint_dtype) if nj else np.zeros(0, wire.joint_dtype)
            try:
                return int(fn(params.contents, bodies, contacts, joints))
            except Exception as e:  # pragma: no cover - surfaced through replace_error
                print("replace callback raised:", repr(e))
                return -99
        self._cb = REPLACE_FN(tramp)

    def __enter__(self):
        L = lib()
        L.s2ref_set_replace(self._cb, None)
        L.s2ref_set_mode(2)
        return self

    def __exit__(self, *a):
        L = lib()
        L.s2ref_set_mode(0)
        L.s2ref_set_replace(ctypes.cast(None, REPLACE_FN), None)
        if L.s2ref_replace_error() != 0:
            raise RuntimeError("replace callback failed with %d" % L.s2ref_replace_error())


def broadphase_capture():
    """State at the last captured s2UpdateBroadPhasePairs entry and the pairs it created:
    (shapes, moved[ns], existing[ne, 2], created[nn, 2])."""
    L = lib()
    ps, pm, pe, pn = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
    ns, ne, nn = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
    L.s2ref_broadphase_capture(ctypes.byref(ps), ctypes.byref(ns), ctypes.byref(pm), ctypes.byref(pe), ctypes.byref(ne),
                               ctypes.byref(pn), ctypes.byref(nn))
    shapes = _copy_array(ps.value, ns.value, wire.shape_dtype)
    moved = _copy_array(pm.value, ns.value, np.dtype(np.uint8))
    existing = _copy_array(pe.value, 2 * ne.value, np.dtype(np.int32)).reshape(-1, 2)
    created = _copy_array(pn.value, 2 * nn.value, np.dtype(np.int32)).reshape(-1, 2)
    return shapes, moved, existing, created


def narrowphase_capture():
    """State at the end of Stage 2 of the last captured step (input of the "update contacts" loop) and at solver
    entry (its output): dict of shapes, bodies, origins, pairs_pre, contacts_pre, pairs_post, contacts_post."""
    L = lib()
    p = [ctypes.c_void_p() for _ in range(7)]
    ns, nb, nc = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
    L.s2ref_narrowphase_capture(ctypes.byref(p[0]), ctypes.byref(ns), ctypes.byref(p[1]), ctypes.byref(nb), ctypes.byref(p[2]),
                                ctypes.byref(p[3]), ctypes.byref(p[4]), ctypes.byref(p[5]), ctypes.byref(p[6]), ctypes.byref(nc))
    return {
        "shapes": _copy_array(p[0].value, ns.value, wire.shape_dtype),
        "bodies": _copy_array(p[1].value, nb.value, wire.body_dtype),
        "origins": _copy_array(p[2].value, 2 * nb.value, np.dtype(np.float32)).reshape(-1, 2),
        "pairs_pre": _copy_array(p[3].value, nc.value, wire.pair_state_dtype),
        "contacts_pre": _copy_array(p[4].value, nc.value, wire.contact_dtype),
        "pairs_post": _copy_array(p[5].value, nc.value, wire.pair_state_dtype),
        "contacts_post": _copy_array(p[6].value, nc.value, wire.contact_dtype),
    }


def narrowphase_seconds(reset=False):
    L = lib()
    L.s2ref_narrowphase_seconds.restype = ctypes.c_double
    return L.s2ref_narrowphase_seconds(1 if reset else 0)
