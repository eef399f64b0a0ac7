"""Python host side of the C-ABI in include/solver2d_amd.h (ctypes over solver2d_amd/libs2amd.so).

The mirror of the reference's plug point: `Solver.solve(params, bodies, contacts, joints)` is
`s2Solve_<Variant>(world, context)` (reference src/solvers.h:70-79) on wire arrays.  There is no
CPU path here: if the HIP library is missing or no GPU is visible the constructor raises.
"""
import ctypes
import os

import numpy as np

from . import wire

_HERE = os.path.dirname(os.path.abspath(__file__))
# S2AMD_LIB: another build of the same C-ABI (experiments: `make -C solver2d_amd/csrc variant NAME=...`)
LIB_PATH = os.environ.get("S2AMD_LIB") or os.path.join(_HERE, "libs2amd.so")
# The tolerance-mode build: the same sources with -ffp-contract=fast (FMA contraction).  Not bit-equal to the reference;
# within the tolerances tests/test_gpu_fast.py states and checks.  A library of its own so that one process can hold both.
FAST_LIB_PATH = os.environ.get("S2AMD_FAST_LIB") or os.path.join(_HERE, "libs2amd_fast.so")

EXPORTS = [
    "s2amd_api_version", "s2amd_build_flags", "s2amd_device_count", "s2amd_device_bus_id", "s2amd_last_error", "s2amd_create", "s2amd_destroy",
    "s2amd_solve", "s2amd_upload", "s2amd_step_resident", "s2amd_download", "s2amd_save_bodies",
    "s2amd_restore_bodies", "s2amd_get_contact_order", "s2amd_get_joint_order", "s2amd_get_writable_bodies", "s2amd_get_stats",
    "s2amd_set_option", "s2amd_export_poses", "s2amd_export_poses_async", "s2amd_export_bodies_async", "s2amd_export_wait", "s2amd_measure_dominant", "s2amd_refit_shapes", "s2amd_find_pairs", "s2amd_synchronize", "s2amd_update_contacts", "s2amd_find_islands", "s2amd_color_constraints",
    "s2amd_world_upload", "s2amd_world_step", "s2amd_world_download", "s2amd_world_find_pairs", "s2amd_world_set_contacts",
    "s2amd_device_alloc", "s2amd_device_free", "s2amd_device_read", "s2amd_world_separated", "s2amd_world_download_boxes", "s2amd_world_set_refit_order", "s2amd_world_download_step", "s2amd_world_set_tree", "s2amd_world_get_tree",
    "s2amd_get_strip_owners",
    "s2amd_sharded_create", "s2amd_sharded_destroy", "s2amd_sharded_shard_count", "s2amd_sharded_solver", "s2amd_sharded_upload", "s2amd_sharded_step",
    "s2amd_sharded_download", "s2amd_sharded_read_bodies", "s2amd_sharded_reshard", "s2amd_sharded_get_partition",
    "s2amd_sharded_step_async", "s2amd_sharded_wait", "s2amd_sharded_get_step_ops", "s2amd_sharded_count_ops",
]

_libs = {}


class S2AmdError(RuntimeError):
    pass


def load(fast=False):
    """Load libs2amd.so (fast=True: libs2amd_fast.so).  Raises if it has not been built (python __graft_entry__.py /
    make -C solver2d_amd/csrc)."""
    path = FAST_LIB_PATH if fast else LIB_PATH
    if path in _libs:
        return _libs[path]
    if not os.path.exists(path):
        raise S2AmdError("HIP extension %s is missing: build it with `make -C solver2d_amd/csrc` "
                         "(there is no CPU fallback)" % path)
    L = ctypes.CDLL(path)
    vp, i32 = ctypes.c_void_p, ctypes.c_int32
    L.s2amd_api_version.restype = ctypes.c_int
    L.s2amd_device_count.restype = ctypes.c_int
    L.s2amd_last_error.restype = ctypes.c_char_p
    L.s2amd_build_flags.restype = ctypes.c_char_p
    L.s2amd_create.argtypes = [ctypes.c_int, ctypes.POINTER(vp)]
    L.s2amd_destroy.argtypes = [vp]
    L.s2amd_destroy.restype = None
    L.s2amd_solve.argtypes = [vp, ctypes.POINTER(wire.StepParams), vp, i32, vp, i32, vp, i32]
    L.s2amd_upload.argtypes = [vp, vp, i32, vp, i32, vp, i32]
    L.s2amd_step_resident.argtypes = [vp, ctypes.POINTER(wire.StepParams)]
    L.s2amd_download.argtypes = [vp, vp, i32, vp, i32, vp, i32]
    L.s2amd_save_bodies.argtypes = [vp]
    L.s2amd_restore_bodies.argtypes = [vp]
    L.s2amd_synchronize.argtypes = [vp]
    L.s2amd_get_contact_order.argtypes = [vp, vp, i32, vp, i32, ctypes.POINTER(i32), ctypes.POINTER(i32)]
    L.s2amd_get_joint_order.argtypes = [vp, vp, i32, vp, i32, ctypes.POINTER(i32), ctypes.POINTER(i32)]
    L.s2amd_get_writable_bodies.argtypes = [vp, vp, i32, ctypes.POINTER(i32)]
    L.s2amd_get_strip_owners.argtypes = [vp, vp, vp, i32, ctypes.POINTER(i32)]
    L.s2amd_get_stats.argtypes = [vp, ctypes.POINTER(wire.StepStats)]
    L.s2amd_set_option.argtypes = [vp, ctypes.c_char_p, i32]
    L.s2amd_export_poses.argtypes = [vp, vp, i32]
    L.s2amd_export_poses_async.argtypes = [vp, vp, i32, i32]
    L.s2amd_export_bodies_async.argtypes = [vp, vp, i32, i32]
    L.s2amd_export_wait.argtypes = [vp, i32]
    L.s2amd_measure_dominant.argtypes = [vp, ctypes.POINTER(wire.StepParams), i32, ctypes.POINTER(ctypes.c_float),
                                         ctypes.POINTER(i32), ctypes.POINTER(i32)]
    L.s2amd_refit_shapes.argtypes = [vp, vp, i32, vp, i32, vp]
    L.s2amd_find_pairs.argtypes = [vp, vp, i32, vp, i32, vp, vp, i32, vp, i32, vp, i32, ctypes.POINTER(i32)]
    L.s2amd_update_contacts.argtypes = [vp, vp, i32, vp, vp, i32, vp, vp, i32, vp]
    L.s2amd_world_upload.argtypes = [vp, vp, i32, vp, i32, vp, i32, vp, i32, vp, vp]
    L.s2amd_world_step.argtypes = [vp, ctypes.POINTER(wire.StepParams), ctypes.POINTER(wire.WorldStepInfo)]
    L.s2amd_world_find_pairs.argtypes = [vp, vp, i32, ctypes.POINTER(i32)]
    L.s2amd_world_set_contacts.argtypes = [vp, vp, i32, vp, vp]
    L.s2amd_world_separated.argtypes = [vp, vp, i32, ctypes.POINTER(i32)]
    L.s2amd_world_download_boxes.argtypes = [vp, vp, i32]
    L.s2amd_world_set_refit_order.argtypes = [vp, vp, i32]
    L.s2amd_world_set_tree.argtypes = [vp, i32, vp, i32, i32]
    L.s2amd_world_get_tree.argtypes = [vp, i32, vp, i32, ctypes.POINTER(i32)]
    L.s2amd_world_download.argtypes = [vp, vp, i32, vp, i32, vp, i32, vp, i32, vp, vp, vp]
    L.s2amd_device_alloc.argtypes = [vp, ctypes.c_uint64, ctypes.POINTER(vp)]
    L.s2amd_device_free.argtypes = [vp, vp]
    L.s2amd_device_read.argtypes = [vp, vp, vp, ctypes.c_uint64]
    L.s2amd_find_islands.argtypes = [vp, vp, i32, vp, i32, vp, i32, vp, ctypes.POINTER(i32)]
    L.s2amd_color_constraints.argtypes = [vp, vp, i32, vp, i32, vp, ctypes.POINTER(i32), ctypes.POINTER(i32)]
    L.s2amd_sharded_create.argtypes = [vp, i32, ctypes.POINTER(vp)]
    L.s2amd_sharded_destroy.argtypes = [vp]
    L.s2amd_sharded_destroy.restype = None
    L.s2amd_sharded_shard_count.argtypes = [vp]
    L.s2amd_sharded_solver.argtypes = [vp, i32]
    L.s2amd_sharded_solver.restype = vp
    L.s2amd_sharded_upload.argtypes = [vp, vp, i32, vp, i32, vp, i32]
    L.s2amd_sharded_step.argtypes = [vp, ctypes.POINTER(wire.StepParams)]
    L.s2amd_sharded_step_async.argtypes = [vp, ctypes.POINTER(wire.StepParams)]
    L.s2amd_sharded_wait.argtypes = [vp]
    L.s2amd_sharded_get_step_ops.argtypes = [vp, ctypes.POINTER(i32), ctypes.POINTER(i32), ctypes.POINTER(i32)]
    L.s2amd_sharded_count_ops.argtypes = [i32, i32, ctypes.POINTER(i32)]
    L.s2amd_sharded_download.argtypes = [vp, vp, i32, vp, i32, vp, i32]
    L.s2amd_sharded_read_bodies.argtypes = [vp, i32, vp, i32]
    L.s2amd_sharded_reshard.argtypes = [vp, vp, i32, vp, i32]
    L.s2amd_sharded_get_partition.argtypes = [vp, vp, i32, ctypes.POINTER(i32), ctypes.POINTER(i32)]
    if L.s2amd_api_version() != wire.API_VERSION:
        raise S2AmdError("libs2amd.so API version %d != %d" % (L.s2amd_api_version(), wire.API_VERSION))
    if fast and not L.s2amd_build_flags().decode().count("contract=fast"):
        raise S2AmdError("%s is not a tolerance-mode build (%s)" % (path, L.s2amd_build_flags().decode()))
    _libs[path] = L
    return L


def device_count():
    return load().s2amd_device_count()


def device_bus_id(device):
    buf = ctypes.create_string_buffer(64)
    _check(load().s2amd_device_bus_id(int(device), buf, 64))
    return buf.value.decode()


def _check(rc, L=None):
    if rc != 0:
        raise S2AmdError("s2amd error %d: %s" % (rc, (L or load()).s2amd_last_error().decode(errors="replace")))


_env_options_logged = False


def env_options():
    """S2AMD_OPTIONS="key=value,key=value": options for every solver this process creates (bisecting a difference
    without editing the caller).  Validated here -- a malformed entry raises naming it -- and logged once to stderr,
    because it silently changes every solver of the process, the parity tests' included."""
    global _env_options_logged
    text = os.environ.get("S2AMD_OPTIONS", "")
    out = []
    for kv in filter(None, (e.strip() for e in text.split(","))):
        k, eq, v = kv.partition("=")
        try:
            if not eq or not k.strip():
                raise ValueError
            out.append((k.strip(), int(v)))
        except ValueError:
            raise S2AmdError("S2AMD_OPTIONS: entry %r is not key=integer" % kv) from None
    if out and not _env_options_logged:
        import sys
        print("[s2amd] S2AMD_OPTIONS applied to every solver: %s" % ", ".join("%s=%d" % e for e in out), file=sys.stderr)
        _env_options_logged = True
    return out


class Solver:
    """One device-resident world (one HIP stream).  Arrays are numpy structured arrays of the
    dtypes in solver2d_amd.wire and are mutated in place, like the reference mutates its pools."""

    def __init__(self, device=0, graph=True, profile=False, fast=False):
        """fast=True: the tolerance-mode build (libs2amd_fast.so, FMA contraction on; results within the stated
        tolerances of the oracle) instead of the bit-exact one."""
        L = load(fast=fast)
        h = ctypes.c_void_p()
        _check(L.s2amd_create(int(device), ctypes.byref(h)), L)
        self._L = L
        self._h = h
        self.fast = bool(fast)
        try:
            self.set_option("graph", 1 if graph else 0)
            self.set_option("profile", 1 if profile else 0)
            for k, v in env_options():
                self.set_option(k, v)
        except Exception:
            self.close()
            raise

    def _ck(self, rc):
        _check(rc, self._L)

    def close(self):
        if getattr(self, "_h", None):
            self._L.s2amd_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def writable_bodies(self, body_capacity):
        """(writable uint8[nb], solver class): the bodies the coloured sweeps write -- what one colour must not share."""
        out = np.zeros(int(body_capacity), dtype=np.uint8)
        cls = ctypes.c_int32()
        self._ck(self._L.s2amd_get_writable_bodies(self._h, wire.as_ptr(out), len(out), ctypes.byref(cls)))
        return out, cls.value

    def set_option(self, key, value):
        self._ck(self._L.s2amd_set_option(self._h, key.encode(), int(value)))

    @staticmethod
    def _args(bodies, contacts, joints):
        for arr, dt in ((bodies, wire.body_dtype), (contacts, wire.contact_dtype), (joints, wire.joint_dtype)):
            if arr.dtype != dt or not arr.flags["C_CONTIGUOUS"]:
                raise ValueError("array must be a contiguous %s array" % (dt.names,))
        return (wire.as_ptr(bodies), len(bodies), wire.as_ptr(contacts), len(contacts), wire.as_ptr(joints), len(joints))

    def solve(self, params, bodies, contacts, joints):
        """== s2Solve_<params.solverType>(world, context): upload, solve, download; in place."""
        self._ck(self._L.s2amd_solve(self._h, ctypes.byref(params), *self._args(bodies, contacts, joints)))
        return bodies, contacts, joints

    def upload(self, bodies, contacts, joints):
        self._ck(self._L.s2amd_upload(self._h, *self._args(bodies, contacts, joints)))

    def step_resident(self, params):
        self._ck(self._L.s2amd_step_resident(self._h, ctypes.byref(params)))

    def download(self, bodies, contacts, joints):
        self._ck(self._L.s2amd_download(self._h, *self._args(bodies, contacts, joints)))
        return bodies, contacts, joints

    def save_bodies(self):
        self._ck(self._L.s2amd_save_bodies(self._h))

    def restore_bodies(self):
        self._ck(self._L.s2amd_restore_bodies(self._h))

    def export_poses(self, device_ptr, capacity):
        """{position, rot} per body into a caller-owned device buffer (float32[capacity, 4])."""
        self._ck(self._L.s2amd_export_poses(self._h, ctypes.c_void_p(int(device_ptr)), int(capacity)))

    def export_bodies_async(self, device_ptr, capacity, slot):
        """Two float4 per body -- {position, rot}, {linearVelocity, angularVelocity, 0} -- enqueued like export_poses_async."""
        self._ck(self._L.s2amd_export_bodies_async(self._h, ctypes.c_void_p(int(device_ptr)), int(capacity), int(slot)))

    def export_poses_async(self, device_ptr, capacity, slot):
        """export_poses enqueued behind the steps already on the solver's stream; pair with export_wait(slot)."""
        self._ck(self._L.s2amd_export_poses_async(self._h, ctypes.c_void_p(int(device_ptr)), int(capacity), int(slot)))

    def export_wait(self, slot):
        self._ck(self._L.s2amd_export_wait(self._h, int(slot)))

    def device_alloc(self, nbytes):
        """A zeroed device buffer owned by the caller (an int address); free it with device_free."""
        p = ctypes.c_void_p()
        self._ck(self._L.s2amd_device_alloc(self._h, int(nbytes), ctypes.byref(p)))
        return int(p.value)

    def device_free(self, ptr):
        self._ck(self._L.s2amd_device_free(self._h, ctypes.c_void_p(int(ptr))))

    def device_read(self, ptr, shape, dtype=np.float32):
        out = np.zeros(shape, dtype=dtype)
        self._ck(self._L.s2amd_device_read(self._h, wire.as_ptr(out.reshape(-1)), ctypes.c_void_p(int(ptr)), out.nbytes))
        return out

    def measure_dominant(self, params, repeats=20):
        """(us per launch, launches per sweep, constraints per launch) of the dominant kernel; see
        s2amd_measure_dominant."""
        us, n, c = ctypes.c_float(), ctypes.c_int32(), ctypes.c_int32()
        self._ck(self._L.s2amd_measure_dominant(self._h, ctypes.byref(params), int(repeats), ctypes.byref(us), ctypes.byref(n), ctypes.byref(c)))
        return us.value, n.value, c.value

    def refit_shapes(self, bodies, shapes, origins):
        """Stage 4 of s2World_Step on wire arrays (in place): origins, tight and fat AABBs, `enlarged`."""
        assert shapes.dtype == wire.shape_dtype and origins.dtype == np.float32 and origins.shape == (len(bodies), 2)
        self._ck(self._L.s2amd_refit_shapes(self._h, wire.as_ptr(bodies), len(bodies), wire.as_ptr(shapes), len(shapes), wire.as_ptr(origins)))
        return shapes, origins

    def update_contacts(self, bodies, origins, shapes, pairs, contacts):
        """Stage 3 of s2World_Step (s2UpdateContact per live contact) on wire arrays, in place; returns status int32[nc]."""
        assert pairs.dtype == wire.pair_state_dtype and contacts.dtype == wire.contact_dtype and len(pairs) == len(contacts)
        assert shapes.dtype == wire.shape_dtype
        origins = np.ascontiguousarray(origins, dtype=np.float32)
        assert origins.shape == (len(bodies), 2)
        status = np.zeros(len(contacts), dtype=np.int32)
        self._ck(self._L.s2amd_update_contacts(self._h, wire.as_ptr(bodies), len(bodies), wire.as_ptr(origins), wire.as_ptr(shapes), len(shapes),
                                            wire.as_ptr(pairs), wire.as_ptr(contacts), len(contacts), wire.as_ptr(status)))
        return status

    # ---- resident world: stage 3 -> solve -> stage 4 chained in HBM ----
    def world_upload(self, bodies, contacts, joints, shapes, pairs, origins):
        assert shapes.dtype == wire.shape_dtype and pairs.dtype == wire.pair_state_dtype and len(pairs) == len(contacts)
        origins = np.ascontiguousarray(origins, dtype=np.float32)
        assert origins.shape == (len(bodies), 2)
        self._ck(self._L.s2amd_world_upload(self._h, *self._args(bodies, contacts, joints), wire.as_ptr(shapes), len(shapes), wire.as_ptr(pairs),
                                         wire.as_ptr(origins)))

    def world_step(self, params):
        """One s2World_Step minus pair creation on the resident world; returns the step's counters as a dict."""
        info = wire.WorldStepInfo()
        self._ck(self._L.s2amd_world_step(self._h, ctypes.byref(params), ctypes.byref(info)))
        return {k: getattr(info, k) for k, _ in wire.WorldStepInfo._fields_}

    def world_find_pairs(self):
        """Stage 1's new pairs for the shapes the last refit moved: int32[n, 2] sorted by (A, B)."""
        cap = 1024
        while True:
            out = np.zeros((cap, 2), dtype=np.int32)
            n = ctypes.c_int32()
            rc = self._L.s2amd_world_find_pairs(self._h, wire.as_ptr(out), cap, ctypes.byref(n))
            if rc == -5 and n.value > cap:  # S2AMD_E_CAPACITY
                cap = n.value
                continue
            self._ck(rc)
            return out[: n.value].copy()

    def world_set_refit_order(self, order):
        """The order the caller's refit visits the movable shapes in (= the move buffer's order)."""
        order = np.ascontiguousarray(order, dtype=np.int32)
        self._ck(self._L.s2amd_world_set_refit_order(self._h, wire.as_ptr(order), len(order)))

    def world_set_tree(self, body_type, nodes, root):
        """One of the reference's three broad-phase trees (s2TreeNode records, wire.tree_node_dtype) onto the device."""
        nodes = np.ascontiguousarray(nodes)
        assert nodes.dtype == wire.tree_node_dtype
        self._ck(self._L.s2amd_world_set_tree(self._h, int(body_type), wire.as_ptr(nodes), len(nodes), int(root)))

    def world_get_tree(self, body_type, node_capacity):
        """(nodes, root) of a device tree as the reference would hold it now."""
        nodes = np.zeros(int(node_capacity), dtype=wire.tree_node_dtype)
        root = ctypes.c_int32()
        self._ck(self._L.s2amd_world_get_tree(self._h, int(body_type), wire.as_ptr(nodes), len(nodes), ctypes.byref(root)))
        return nodes, root.value

    def world_download_boxes(self, shape_capacity):
        """s2amdShapeBox of every resident shape slot after the last world_step."""
        out = np.zeros(int(shape_capacity), dtype=wire.shape_box_dtype)
        self._ck(self._L.s2amd_world_download_boxes(self._h, wire.as_ptr(out), len(out)))
        return out

    def world_separated(self, expected=64):
        """Contact slots the last world_step destroyed (their pairs separated), ascending."""
        out = np.zeros(max(int(expected), 1), dtype=np.int32)
        n = ctypes.c_int32()
        rc = self._L.s2amd_world_separated(self._h, wire.as_ptr(out), len(out), ctypes.byref(n))
        if rc == -5:
            out = np.zeros(n.value, dtype=np.int32)
            rc = self._L.s2amd_world_separated(self._h, wire.as_ptr(out), len(out), ctypes.byref(n))
        self._ck(rc)
        return out[: n.value].copy()

    def world_set_contacts(self, slots, contacts, pairs):
        slots = np.ascontiguousarray(slots, dtype=np.int32)
        contacts = np.ascontiguousarray(contacts)
        pairs = np.ascontiguousarray(pairs)
        assert contacts.dtype == wire.contact_dtype and pairs.dtype == wire.pair_state_dtype and len(slots) == len(contacts) == len(pairs)
        self._ck(self._L.s2amd_world_set_contacts(self._h, wire.as_ptr(slots), len(slots), wire.as_ptr(contacts), wire.as_ptr(pairs)))

    def world_download(self, bodies, contacts, joints, shapes, pairs, origins):
        """Fills the given arrays (same sizes as uploaded) and returns them with the last stage-3 status."""
        origins = np.ascontiguousarray(origins, dtype=np.float32)
        status = np.zeros(len(contacts), dtype=np.int32)
        self._ck(self._L.s2amd_world_download(self._h, *self._args(bodies, contacts, joints), wire.as_ptr(shapes), len(shapes), wire.as_ptr(pairs),
                                           wire.as_ptr(origins), wire.as_ptr(status)))
        return bodies, contacts, joints, shapes, pairs, origins, status

    def find_islands(self, bodies, contacts, joints):
        """(island_of_body int32[nb], island_count): connected components over the movable bodies, on the device."""
        island = np.full(len(bodies), -2, dtype=np.int32)
        n = ctypes.c_int32()
        self._ck(self._L.s2amd_find_islands(self._h, wire.as_ptr(bodies), len(bodies), wire.as_ptr(contacts), len(contacts), wire.as_ptr(joints),
                                         len(joints), wire.as_ptr(island), ctypes.byref(n)))
        return island, n.value

    def color_constraints(self, bodies, contacts):
        """(color_of_contact int32[nc], color_count, rounds): deterministic Jones-Plassmann colouring on the device."""
        colour = np.full(len(contacts), -2, dtype=np.int32)
        n, r = ctypes.c_int32(), ctypes.c_int32()
        self._ck(self._L.s2amd_color_constraints(self._h, wire.as_ptr(bodies), len(bodies), wire.as_ptr(contacts), len(contacts), wire.as_ptr(colour),
                                              ctypes.byref(n), ctypes.byref(r)))
        return colour, n.value, r.value

    def find_pairs(self, bodies, shapes, moved, existing, joints):
        """New broad-phase pairs as int32[n, 2] sorted by (A, B); see s2amd_find_pairs."""
        existing = np.ascontiguousarray(existing, dtype=np.int32).reshape(-1, 2)
        moved = np.ascontiguousarray(moved, dtype=np.uint8)
        cap = 4096
        while True:
            out = np.zeros((cap, 2), dtype=np.int32)
            n = ctypes.c_int32()
            rc = self._L.s2amd_find_pairs(self._h, wire.as_ptr(bodies), len(bodies), wire.as_ptr(shapes), len(shapes), wire.as_ptr(moved),
                                         wire.as_ptr(existing), len(existing), wire.as_ptr(joints), len(joints), wire.as_ptr(out), cap,
                                         ctypes.byref(n))
            if rc == -5 and n.value > cap:
                cap = n.value
                continue
            self._ck(rc)
            return out[: n.value].copy()

    def _order(self, fn):
        n, nc = ctypes.c_int32(), ctypes.c_int32()
        self._ck(fn(self._h, None, 0, None, 0, ctypes.byref(n), ctypes.byref(nc)))
        order = np.zeros(max(n.value, 1), dtype=np.int32)
        offsets = np.zeros(nc.value + 1, dtype=np.int32)
        self._ck(fn(self._h, order.ctypes.data, len(order), offsets.ctypes.data, len(offsets), ctypes.byref(n), ctypes.byref(nc)))
        return order[: n.value].copy(), offsets

    def contact_order(self):
        """(order, colorOffsets) of the last step; see s2amd_get_contact_order."""
        return self._order(self._L.s2amd_get_contact_order)

    def strip_owners(self, body_capacity):
        """(ownerStrip, onSeam, stripCount): see s2amd_get_strip_owners."""
        owner = np.full(max(body_capacity, 1), -1, dtype=np.int32)
        seam = np.full(max(body_capacity, 1), -1, dtype=np.int32)
        n = ctypes.c_int32()
        self._ck(self._L.s2amd_get_strip_owners(self._h, owner.ctypes.data, seam.ctypes.data, len(owner), ctypes.byref(n)))
        return owner[:body_capacity], seam[:body_capacity], n.value

    def joint_order(self):
        return self._order(self._L.s2amd_get_joint_order)

    def synchronize(self):
        """Waits for the steps enqueued under option "async" and raises a deferred device error."""
        self._ck(self._L.s2amd_synchronize(self._h))

    def stats(self):
        st = wire.StepStats()
        self._ck(self._L.s2amd_get_stats(self._h, ctypes.byref(st)))
        return {k: getattr(st, k) for k, _ in wire.StepStats._fields_}


class _BorrowedSolver(Solver):
    """A shard's s2amdSolver as a Solver object (orders, stats, options): owned by the ShardedSolver, never destroyed from here."""

    def __init__(self, L, handle):
        self._L, self._h, self.fast = L, ctypes.c_void_p(handle), False

    def close(self):
        self._h = None


class ShardedSolver:
    """include/solver2d_amd.h: s2amd_sharded_* -- one process, one shard per entry of `devices` (HIP ordinals; the same ordinal more
    than once = logical shards on one GPU).  The world's islands are found on the device and bin-packed onto the shards; a step is every
    shard's s2Solve_* plus ONE exchange of the owned body records between the devices.  Arrays as for Solver."""

    def __init__(self, devices, fast=False):
        L = load(fast=fast)
        devs = (ctypes.c_int32 * len(devices))(*[int(d) for d in devices])
        h = ctypes.c_void_p()
        _check(L.s2amd_sharded_create(devs, len(devices), ctypes.byref(h)), L)
        self._L, self._h = L, h
        self.shards = len(devices)
        self.body_capacity = 0

    def close(self):
        if getattr(self, "_h", None):
            self._L.s2amd_sharded_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _ck(self, rc):
        _check(rc, self._L)

    def shard(self, i):
        """The Solver of shard i (borrowed: queries and options only)."""
        h = self._L.s2amd_sharded_solver(self._h, int(i))
        if not h:
            raise S2AmdError("no shard %d" % i)
        return _BorrowedSolver(self._L, h)

    def upload(self, bodies, contacts, joints):
        self._ck(self._L.s2amd_sharded_upload(self._h, *Solver._args(bodies, contacts, joints)))
        self.body_capacity = len(bodies)

    def step(self, params):
        self._ck(self._L.s2amd_sharded_step(self._h, ctypes.byref(params)))

    def step_async(self, params):
        """The step enqueued on the shards' streams, nothing waited for (wait() collects)."""
        self._ck(self._L.s2amd_sharded_step_async(self._h, ctypes.byref(params)))

    def wait(self):
        self._ck(self._L.s2amd_sharded_wait(self._h))

    def step_ops(self):
        """(stream operations the last step enqueued, host waits since, exchange form: 0 stores / 1 rccl / 2 peer copies)"""
        ops, waits, ex = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
        self._ck(self._L.s2amd_sharded_get_step_ops(self._h, ctypes.byref(ops), ctypes.byref(waits), ctypes.byref(ex)))
        return ops.value, waits.value, ex.value

    def download(self, bodies, contacts, joints):
        self._ck(self._L.s2amd_sharded_download(self._h, *Solver._args(bodies, contacts, joints)))
        return bodies, contacts, joints

    def read_bodies(self, shard=0):
        """float32[bodies, 8] {position, rot, linearVelocity, angularVelocity, 0} of the WHOLE world as shard `shard`'s device holds it."""
        out = np.zeros((self.body_capacity, 8), dtype=np.float32)
        self._ck(self._L.s2amd_sharded_read_bodies(self._h, int(shard), wire.as_ptr(out.reshape(-1)), self.body_capacity))
        return out

    def reshard(self, contacts=None, joints=None):
        for arr, dt in ((contacts, wire.contact_dtype), (joints, wire.joint_dtype)):
            if arr is not None and (arr.dtype != dt or not arr.flags["C_CONTIGUOUS"]):
                raise ValueError("array must be a contiguous %s array" % (dt.names,))
        self._ck(self._L.s2amd_sharded_reshard(self._h, wire.as_ptr(contacts) if contacts is not None else None, len(contacts) if contacts is not None else 0,
                                               wire.as_ptr(joints) if joints is not None else None, len(joints) if joints is not None else 0))

    def partition(self):
        """(shard_of_body int32[bodies] (-1: owned by nobody), island_count, reshards so far)"""
        out = np.full(max(self.body_capacity, 1), -1, dtype=np.int32)
        n, r = ctypes.c_int32(), ctypes.c_int32()
        self._ck(self._L.s2amd_sharded_get_partition(self._h, out.ctypes.data, len(out), ctypes.byref(n), ctypes.byref(r)))
        return out[: self.body_capacity], n.value, r.value
