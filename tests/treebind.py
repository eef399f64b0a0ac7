"""ctypes view of the reference's OWN s2DynamicTree (include/solver2d/dynamic_tree.h, src/dynamic_tree.c) inside
oracle/_ref/libs2ref.so -- TEST INFRASTRUCTURE: the checker of tests/tree_parallel.py and of the device trees."""
import ctypes
import numpy as np

from tests import refbind
from tests.tree_parallel import NODE, NULL


class Vec2(ctypes.Structure):
    _fields_ = [("x", ctypes.c_float), ("y", ctypes.c_float)]


class Box(ctypes.Structure):
    _fields_ = [("lower", Vec2), ("upper", Vec2)]


class DynamicTree(ctypes.Structure):  # s2DynamicTree, dynamic_tree.h:43-58
    _fields_ = [("nodes", ctypes.c_void_p), ("root", ctypes.c_int32), ("nodeCount", ctypes.c_int32), ("nodeCapacity", ctypes.c_int32),
                ("freeList", ctypes.c_int32), ("proxyCount", ctypes.c_int32), ("leafIndices", ctypes.c_void_p), ("leafBoxes", ctypes.c_void_p),
                ("leafCenters", ctypes.c_void_p), ("binIndices", ctypes.c_void_p), ("rebuildCapacity", ctypes.c_int32)]


def box(b):
    return Box(Vec2(float(b[0]), float(b[1])), Vec2(float(b[2]), float(b[3])))


_ready = False


def _lib():
    global _ready
    L = refbind.lib()
    if not _ready:
        P = ctypes.POINTER(DynamicTree)
        L.s2DynamicTree_Create.restype = DynamicTree
        L.s2DynamicTree_Destroy.argtypes = [P]
        L.s2DynamicTree_CreateProxy.restype = ctypes.c_int32
        L.s2DynamicTree_CreateProxy.argtypes = [P, Box, ctypes.c_uint32, ctypes.c_int32]
        L.s2DynamicTree_DestroyProxy.argtypes = [P, ctypes.c_int32]
        L.s2DynamicTree_EnlargeProxy.argtypes = [P, ctypes.c_int32, Box]
        L.s2DynamicTree_Rebuild.restype = ctypes.c_int32
        L.s2DynamicTree_Rebuild.argtypes = [P, ctypes.c_bool]
        _ready = True
    return L


class RefTree:
    def __init__(self):
        self.t = _lib().s2DynamicTree_Create()

    def close(self):
        _lib().s2DynamicTree_Destroy(ctypes.byref(self.t))

    def create_proxy(self, b, category=1, user=0):
        return _lib().s2DynamicTree_CreateProxy(ctypes.byref(self.t), box(b), category, user)

    def destroy_proxy(self, proxy):
        _lib().s2DynamicTree_DestroyProxy(ctypes.byref(self.t), proxy)

    def enlarge(self, proxy, b):
        _lib().s2DynamicTree_EnlargeProxy(ctypes.byref(self.t), proxy, box(b))

    def rebuild(self):
        return _lib().s2DynamicTree_Rebuild(ctypes.byref(self.t), False)

    def nodes(self):
        n = self.t.nodeCapacity
        buf = (ctypes.c_char * (n * NODE.itemsize)).from_address(self.t.nodes)
        return np.frombuffer(buf, dtype=NODE, count=n).copy()

    @property
    def root(self):
        return self.t.root

    @property
    def free_list(self):
        return self.t.freeList


def same_nodes(a, b, what=""):
    """every live node equal field for field (zeros compare equal whatever their sign); free nodes: `next` and height"""
    assert len(a) == len(b), what
    for f in ("parent", "height"):
        assert np.array_equal(a[f], b[f]), "%s: %s differs at %s" % (what, f, np.nonzero(a[f] != b[f])[0][:8])
    live = a["height"] >= 0
    for f in ("child1", "child2", "userData", "categoryBits", "enlarged"):
        assert np.array_equal(a[f][live], b[f][live]), "%s: %s differs at %s" % (what, f, np.nonzero(live & (a[f] != b[f]))[0][:8])
    assert np.all(a["aabb"][live] == b["aabb"][live]), "%s: boxes differ at %s" % (what, np.nonzero(live & np.any(a["aabb"] != b["aabb"], axis=1))[0][:8])
