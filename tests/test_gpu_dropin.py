"""Drop-in test: the UNMODIFIED reference (broad phase, narrow phase, contact bookkeeping, world
step) drives the HIP solver through the s2Solve_* plug point (oracle/ref_hook.c replace mode).

L2, per step: the HIP result must equal the oracle run in the device's order, bit for bit.
L3, per trajectory: Gauss-Seidel is order dependent, so the GPU (colour order) and the reference
(pool order) trajectories differ by an amount that is a property of the SOLVER, not of the port.
The stated tolerance is therefore relative: the GPU-vs-reference deviation after N steps must not
exceed 3x the deviation between two CPU runs of the reference algorithm itself that differ only in
sweep order (pool order vs reversed pool order, both through the bit-pinned oracle), floor 2 cm.
Needs oracle/_ref/libs2ref.so (shipped prebuilt to the GPU box).
"""
import numpy as np
import pytest

from solver2d_amd import hip, wire
from tests import common, oraclebind, refbind

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not refbind.available(), reason="oracle/_ref/libs2ref.so not built")]

CASES = [
    ("pyramid", 20, 0, 90),
    ("mixed", 24, 0, 90),
    ("joint_grid", 10, 10, 60),
    ("tumbler", 150, 0, 60),
    ("circle_pile", 16, 0, 80),
]


@pytest.mark.parametrize("solver_name", wire.SOLVER_NAMES)
@pytest.mark.parametrize("scene,p0,p1,steps", CASES)
def test_reference_world_with_hip_solver(scene, p0, p1, steps, solver_name):
    vel, pos = common.DEFAULT_ITERS[solver_name]
    with hip.Solver(0) as gpu:
        mismatches = []

        def replace(params, bodies, contacts, joints):
            pre = (bodies.copy(), contacts.copy(), joints.copy())
            gpu.solve(params, bodies, contacts, joints)
            order, _ = gpu.contact_order()
            jorder, _ = gpu.joint_order()
            oraclebind.solve(params, *pre, contact_order=order, joint_order=jorder)
            try:
                common.compare_exact((bodies, contacts, joints), pre, "step")
            except AssertionError as e:
                mismatches.append(str(e))
            return 0

        def reversed_order(params, bodies, contacts, joints):
            co = np.flatnonzero(contacts["pointCount"] > 0)[::-1].astype(np.int32)
            jo = np.flatnonzero(joints["type"] >= 0)[::-1].astype(np.int32)
            oraclebind.solve(params, bodies, contacts, joints, contact_order=co, joint_order=jo)
            return 0

        with refbind.RefWorld(scene, solver_name, p0, p1) as wg, refbind.RefWorld(scene, solver_name, p0, p1) as wr, \
                refbind.RefWorld(scene, solver_name, p0, p1) as wo:
            with refbind.Replace(replace):
                for _ in range(steps):
                    wg.step(1.0 / 60.0, vel, pos, True)
            with refbind.Replace(reversed_order):
                for _ in range(steps):
                    wo.step(1.0 / 60.0, vel, pos, True)
            for _ in range(steps):
                wr.step(1.0 / 60.0, vel, pos, True)
            assert not mismatches, "%d steps differ from the oracle; first: %s" % (len(mismatches), mismatches[0])

            bg, _, _ = wg.pack()
            br, _, _ = wr.pack()
            bo, _, _ = wo.pack()
            live = br["type"] >= 0
            assert np.isfinite(bg["position"][live]).all() or not np.isfinite(br["position"][live]).all()
            if scene in ("pyramid", "joint_grid", "vertical_stack"):
                dev_gpu = float(np.abs(bg["position"][live] - br["position"][live]).max())
                dev_cpu = float(np.abs(bo["position"][live] - br["position"][live]).max())
                assert dev_gpu <= max(3.0 * dev_cpu, 0.02), "GPU order deviates %.4g m, CPU reversed order %.4g m" % (dev_gpu, dev_cpu)
                if dev_gpu < 0.02:
                    pa_g, pb_g = wg.contact_pairs()
                    pa_r, pb_r = wr.contact_pairs()
                    assert sorted(zip(pa_g.tolist(), pb_g.tolist())) == sorted(zip(pa_r.tolist(), pb_r.tolist()))


@pytest.mark.parametrize("scene,p0,solver_name,steps", [("pyramid", 20, "TGS_Soft", 60), ("mixed", 24, "PGS_NGS", 60), ("tumbler", 150, "SoftStep", 40)])
def test_native_shim_public_api_runs_on_the_gpu(scene, p0, solver_name, steps):
    """The binding of INTEGRATION.md in C, no Python between the two libraries: oracle/ref_hook.c: s2ref_use_amd dlopens
    libs2amd.so and routes the reference's s2Solve_* switch into s2amd_solve, so a program that only calls the PUBLIC
    s2World_Step runs its solver on the GPU.  The trajectory must be the one the Python-callback route produces (same
    library, same inputs): bit for bit."""
    import ctypes
    vel, pos = common.DEFAULT_ITERS[solver_name]
    L = refbind.lib()
    L.s2ref_use_amd.argtypes = [ctypes.c_char_p, ctypes.c_int]
    L.s2ref_use_amd.restype = ctypes.c_int
    with refbind.RefWorld(scene, solver_name, p0, 0) as native:
        assert L.s2ref_use_amd(hip.LIB_PATH.encode(), 0) == 0
        try:
            for _ in range(steps):
                native.step(1.0 / 60.0, vel, pos, True)
            assert L.s2ref_replace_error() == 0
        finally:
            assert L.s2ref_use_amd(None, 0) == 0
        bn, cn, jn = native.pack()
    with hip.Solver(0) as gpu, refbind.RefWorld(scene, solver_name, p0, 0) as routed:
        def replace(params, bodies, contacts, joints):
            gpu.solve(params, bodies, contacts, joints)
            return 0
        with refbind.Replace(replace):
            for _ in range(steps):
                routed.step(1.0 / 60.0, vel, pos, True)
        br, cr, jr = routed.pack()
    assert bn.tobytes() == br.tobytes() and cn.tobytes() == cr.tobytes() and jn.tobytes() == jr.tobytes()
    assert np.isfinite(bn["position"][bn["type"] >= 0]).all()


WHOLE_STEP_CASES = [("pyramid", 20, "TGS_Soft", 60), ("joint_grid", 10, "TGS_NGS", 40), ("circle_pile", 16, "XPBD", 60), ("pyramid", 45, "PGS_Soft", 30)]
WHOLE_STEP_CASES += [("mixed", 24, name, 60) for name in wire.SOLVER_NAMES] + [("tumbler", 150, name, 50) for name in wire.SOLVER_NAMES]


@pytest.mark.parametrize("scene,p0,solver_name,steps", WHOLE_STEP_CASES)
def test_native_shim_whole_step_on_the_gpu(scene, p0, solver_name, steps, monkeypatch):
    """oracle/ref_hook.c: s2ref_use_amd_world: the library's exported s2World_Step keeps the reference's stage 1 and 2 (dynamic
    trees, contact pool) and runs stage 3, the solve and stage 4 on the resident world chain (s2amd_world_step), bringing
    back bodies, separations and re-inflated boxes every step and the manifolds on demand.  The trajectory must be the one
    the solver-only shim produces -- the device's narrow phase and refit are bit-exact restatements of the host's, the
    constraint graph and hence the sweep order are the same --, bit for bit: bodies, manifolds, impulses, pair table."""
    import ctypes
    # (the same structure POLICY on both routes: "group_patience" -- LDS groups stand back where created contacts keep hitting them -- is driven
    # by what the placement of the whole-step route could not take, events the solve-only route, which builds for every new contact, never
    # sees; "flip_colours" -- a hub's manifold that gains its points takes a colour position -- places where that route builds)
    monkeypatch.setenv("S2AMD_OPTIONS", "group_patience=0,flip_colours=0")
    vel, pos = common.DEFAULT_ITERS[solver_name]
    L = refbind.lib()
    for f in (L.s2ref_use_amd, L.s2ref_use_amd_world):
        f.argtypes = [ctypes.c_char_p, ctypes.c_int]
        f.restype = ctypes.c_int
    L.s2ref_world_uploads.restype = ctypes.c_long
    p1 = 10 if scene == "joint_grid" else 0
    results = []
    for use in (L.s2ref_use_amd_world, L.s2ref_use_amd):
        with refbind.RefWorld(scene, solver_name, p0, p1) as w:
            assert use(hip.LIB_PATH.encode(), 0) == 0
            uploads0 = L.s2ref_world_uploads()
            try:
                for _ in range(steps):
                    w.step(1.0 / 60.0, vel, pos, True)
                assert L.s2ref_replace_error() == 0
                b, c, j = w.pack()
                pairs = w.contact_pairs()
                uploads = L.s2ref_world_uploads() - uploads0
            finally:
                assert use(None, 0) == 0
            results.append((b, c, j, pairs, uploads))
    (bw, cw, jw, pw, uw), (bs, cs, js, ps, us) = results
    assert us == 0 and 1 <= uw <= 4, (uw, us)  # the world is uploaded once, and again only when the contact pool grew
    assert np.array_equal(pw[0], ps[0]) and np.array_equal(pw[1], ps[1]), "pair tables differ"
    live = bs["type"] >= 0
    assert np.isfinite(bw["position"][live]).all()
    common.compare_exact((bw, cw, jw), (bs, cs, js), "whole-step shim vs solver-only shim")


@pytest.mark.parametrize("scene,p0,solver_name,steps", [("pyramid", 20, "TGS_Soft", 90), ("circle_pile", 16, "PGS_Soft", 80), ("tumbler", 150, "SoftStep", 120),
                                                        ("mixed", 24, "PGS_NGS_Block", 120), ("shapes_zoo", 40, "TGS_Sticky", 150),
                                                        ("far_ragdoll_pile", 0, "PGS", 100), ("card_house", 0, "XPBD", 60)])
def test_native_shim_whole_step_with_device_pairs(scene, p0, solver_name, steps):
    """As above with stage 1's pair discovery on the device too (s2ref_world_device_pairs): the reference's trees are
    kept up to date but no longer queried.  The device returns the new pairs as a set; the binding creates the contacts in
    the order the reference's own stage 1 would have found them in (s2amdBinding_OrderPairs, checked against the reference
    in tests/test_creation_order.py), so every contact gets the same pool slot as with the host's stage 1 and the two
    routes are the same computation: bodies, manifolds and the pool's pairs slot for slot, bit for bit, at the end of a
    trajectory that creates and destroys contacts all the way."""
    import ctypes
    vel, pos = common.DEFAULT_ITERS[solver_name]
    L = refbind.lib()
    L.s2ref_use_amd_world.argtypes = [ctypes.c_char_p, ctypes.c_int]
    L.s2ref_use_amd_world.restype = ctypes.c_int
    results = []
    for device_pairs in (1, 0):
        with refbind.RefWorld(scene, solver_name, p0, 0) as w:
            assert L.s2ref_use_amd_world(hip.LIB_PATH.encode(), 0) == 0
            L.s2ref_world_device_pairs(device_pairs)
            try:
                for _ in range(steps):
                    w.step(1.0 / 60.0, vel, pos, True)
                assert L.s2ref_replace_error() == 0
                b, c, j = w.pack()
                pa, pb = w.contact_pairs()
            finally:
                L.s2ref_world_device_pairs(0)
                assert L.s2ref_use_amd_world(None, 0) == 0
            live = pa >= 0
            pairs = list(zip(pa[live].tolist(), pb[live].tolist()))
            assert len(set(pairs)) == len(pairs), "duplicate contact"
            results.append((b, c, j, pa, pb))
    (bd, cd, jd, pad, pbd), (bh, ch, jh, pah, pbh) = results
    assert int((pah >= 0).sum()) > 0
    assert np.array_equal(pad, pah) and np.array_equal(pbd, pbh), "contacts in other pool slots: %d of %d differ" % (int((pad != pah).sum() + (pbd != pbh).sum()), len(pah))
    common.compare_exact((bd, cd, jd), (bh, ch, jh), "%s/%s device pairs vs host pairs" % (scene, solver_name))


def test_native_shim_whole_step_notices_a_replaced_world():
    """Worlds live in a static array in the reference: a world destroyed and created again sits at the same address with
    pools of the same size.  The shim must not mistake it for the world it holds on the device (s2World.stepId says so), and
    a step taken by the reference itself in between (mode off and on again) must be noticed the same way."""
    import ctypes
    L = refbind.lib()
    L.s2ref_use_amd_world.argtypes = [ctypes.c_char_p, ctypes.c_int]
    L.s2ref_use_amd_world.restype = ctypes.c_int
    vel, pos = common.DEFAULT_ITERS["TGS_Soft"]
    assert L.s2ref_use_amd_world(hip.LIB_PATH.encode(), 0) == 0
    try:
        with refbind.RefWorld("mixed", "TGS_Soft", 24, 0) as first:
            for _ in range(10):
                first.step(1.0 / 60.0, vel, pos, True)
        with refbind.RefWorld("mixed", "TGS_Soft", 24, 0) as second:
            for _ in range(20):
                second.step(1.0 / 60.0, vel, pos, True)
            got = second.pack()
        assert L.s2ref_replace_error() == 0
    finally:
        assert L.s2ref_use_amd_world(None, 0) == 0
    assert L.s2ref_use_amd_world(hip.LIB_PATH.encode(), 0) == 0
    try:
        with refbind.RefWorld("mixed", "TGS_Soft", 24, 0) as fresh:
            for _ in range(20):
                fresh.step(1.0 / 60.0, vel, pos, True)
            want = fresh.pack()
    finally:
        assert L.s2ref_use_amd_world(None, 0) == 0
    common.compare_exact(got, want, "second world in the same slot vs a fresh run")


def test_ten_worlds_interleaved_keep_their_own_device_state():
    """The samples' GUI keeps up to ten worlds alive and steps them one after the other every frame (samples/main.cpp:36,
    :805-813; the pool has 32 slots, include/solver2d/constants.h:12).  The binding (shim/s2_amd_binding.c) holds ONE device
    state per world, indexed by s2World.index: ten worlds -- one per solver -- stepped interleaved for 60 frames are each
    uploaded exactly once, and every one ends bit-identical to the same world stepped alone."""
    import ctypes
    L = refbind.lib()
    L.s2ref_use_amd_world.argtypes = [ctypes.c_char_p, ctypes.c_int]
    L.s2ref_use_amd_world.restype = ctypes.c_int
    L.s2ref_world_uploads.restype = ctypes.c_long
    frames = 60
    scenes = ["mixed", "pyramid", "circle_pile", "tumbler", "ragdoll", "arch", "joint_grid", "card_house", "shapes_zoo", "vertical_stack"]
    sizes = {"mixed": 24, "pyramid": 14, "circle_pile": 16, "tumbler": 80, "joint_grid": 8, "shapes_zoo": 30, "vertical_stack": 8}
    cases = [(scenes[i], sizes.get(scenes[i], 0), name) + common.DEFAULT_ITERS[name] for i, name in enumerate(wire.SOLVER_NAMES)]

    def final_state(w):
        b, c, j = w.pack()
        return b.tobytes(), c.tobytes(), j.tobytes(), w.contact_pairs()[0].tobytes()

    alone = []
    for scene, p0, name, vel, pos in cases:
        with refbind.RefWorld(scene, name, p0, 8 if scene == "joint_grid" else 0) as w:
            assert L.s2ref_use_amd_world(hip.LIB_PATH.encode(), 0) == 0
            try:
                for _ in range(frames):
                    w.step(1.0 / 60.0, vel, pos, True)
                assert L.s2ref_replace_error() == 0
                alone.append(final_state(w))
            finally:
                assert L.s2ref_use_amd_world(None, 0) == 0
    worlds = [refbind.RefWorld(scene, name, p0, 8 if scene == "joint_grid" else 0) for scene, p0, name, _v, _p in cases]
    try:
        assert L.s2ref_use_amd_world(hip.LIB_PATH.encode(), 0) == 0
        uploads0 = L.s2ref_world_uploads()
        grew = [w.sizes()[1] for w in worlds]
        regrowths = 0
        for _ in range(frames):
            for i, (w, (_scene, _p0, _name, vel, pos)) in enumerate(zip(worlds, cases)):
                w.step(1.0 / 60.0, vel, pos, True)
                if w.sizes()[1] != grew[i]:  # the contact pool grew: the one legitimate reason for another upload
                    grew[i] = w.sizes()[1]
                    regrowths += 1
        assert L.s2ref_replace_error() == 0
        uploads = L.s2ref_world_uploads() - uploads0
        together = [final_state(w) for w in worlds]
    finally:
        assert L.s2ref_use_amd_world(None, 0) == 0
        for w in worlds:
            w.close()
    assert len(worlds) <= uploads <= len(worlds) + regrowths, (uploads, regrowths)
    for (scene, _p0, name, _v, _p), a, t in zip(cases, alone, together):
        assert a == t, "%s/%s differs between the interleaved and the stand-alone run" % (scene, name)


def test_destroyed_world_releases_its_device_state_and_the_slot_starts_afresh():
    """s2DestroyWorld frees the world's device state (the binding is called first); a new world created in the same slot of
    the reference's world table is uploaded from scratch and is not confused with its predecessor."""
    import ctypes
    L = refbind.lib()
    L.s2ref_use_amd_world.argtypes = [ctypes.c_char_p, ctypes.c_int]
    L.s2ref_use_amd_world.restype = ctypes.c_int
    L.s2ref_world_uploads.restype = ctypes.c_long
    assert L.s2ref_use_amd_world(hip.LIB_PATH.encode(), 0) == 0
    try:
        u0 = L.s2ref_world_uploads()
        with refbind.RefWorld("pyramid", "TGS_Soft", 10, 0) as w:
            first_index = w.id.index
            for _ in range(5):
                w.step(1.0 / 60.0, 8, 4, True)
        with refbind.RefWorld("mixed", "PGS", 24, 0) as w:
            assert w.id.index == first_index  # the same slot of s2_worlds[]
            for _ in range(30):
                w.step(1.0 / 60.0, 4, 2, True)
            got = w.pack()
            assert L.s2ref_replace_error() == 0
        assert L.s2ref_world_uploads() - u0 >= 2
    finally:
        assert L.s2ref_use_amd_world(None, 0) == 0
    # the same second world with the solver-only binding (host stage 3 / 4): bit-identical
    assert L.s2ref_use_amd(hip.LIB_PATH.encode(), 0) == 0
    try:
        with refbind.RefWorld("mixed", "PGS", 24, 0) as w:
            for _ in range(30):
                w.step(1.0 / 60.0, 4, 2, True)
            want = w.pack()
    finally:
        assert L.s2ref_use_amd(None, 0) == 0
    common.compare_exact(got, want, "second world in a re-used slot")
