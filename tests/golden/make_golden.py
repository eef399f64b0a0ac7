"""Generates tests/golden/*.npz from the UNMODIFIED reference (oracle/_ref/libs2ref.so).

Run in the build container (needs `make -C oracle ref`, i.e. /root/reference present):
    python tests/golden/make_golden.py
Each fixture is data only: the wire-format state at s2Solve_* entry ("pre_*") and exit ("post_*")
of one s2World_Step of the reference, plus the step parameters.  No reference source text is
stored.  tests/test_golden.py replays pre -> oracle -> compares bitwise with post, and the GPU
tests replay pre -> HIP -> compare with the oracle run in the device's colour order.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from solver2d_amd import wire  # noqa: E402
from tests import common, refbind  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))

# (scene, p0, p1, capture-at-steps)
CASES = [
    ("pyramid", 10, 0, (0, 1, 45)),
    ("mixed", 24, 0, (30, 75)),
    ("joint_grid", 6, 6, (0, 20)),
    ("circle_pile", 12, 0, (40,)),
    # the reference's own edge-case samples (solver2d_amd/scenes/scenes.c cites each): capture steps chosen after the
    # interesting event -- the heavy box has landed, the overlap is at its deepest, the ragdoll lies on its joint limits
    ("arch", 0, 0, (30,)),
    ("high_mass_ratio", 1, 0, (50,)),
    ("high_mass_ratio", 2, 0, (115,)),
    ("high_mass_ratio", 3, 0, (115,)),
    ("overlap_recovery", 0, 0, (3,)),
    ("card_house", 0, 0, (20,)),
    ("far_pyramid", 0, 0, (30,)),
    ("far_stack", 0, 0, (30,)),
    ("far_recovery", 0, 0, (3,)),
    ("far_ragdoll_pile", 0, 0, (60,)),
    ("far_chain", 0, 0, (40,)),
    ("ragdoll", 0, 0, (75,)),
    ("ball_and_chain", 40, 0, (50,)),
    ("bridge", 40, 0, (30,)),
    # (round 6) the rest of the reference's samples: at rest, right after the top circle is destroyed, sliding on mixed frictions, under
    # applied forces, mid-topple, packed and overlapping, stacking up, with three ragdolls in the funnel, with joints still open
    ("single_box", 0, 0, (30,)),
    ("warm_start_energy", 0, 0, (121,)),
    ("friction_ramp", 0, 0, (90,)),
    ("rush", 0, 0, (40,)),
    ("double_domino", 0, 0, (60,)),
    ("confined", 0, 0, (10,)),
    ("circle_stack", 0, 0, (100,)),
    ("ragdoll_stress", 0, 0, (95,)),
    ("stretched_chain", 0, 0, (20,)),
]
ONLY_MISSING = bool(os.environ.get("S2_GOLDEN_ONLY_MISSING"))


def big_inputs():
    """Full-size solver INPUTS (no post state: the GPU tests check them against the oracle): BASELINE.json configs[2], the Tumbler with
    10,000 boxes stepped by the reference until the boxes have settled in the turning drum, captured at s2Solve_* entry.  Settled
    under TGS_Soft: the reference's own Jacobi solver diverges on piles."""
    path = os.path.join(OUT, "big_tumbler10000_input.npz")
    if ONLY_MISSING and os.path.exists(path):
        return 0
    with refbind.RefWorld("tumbler", "TGS_Soft", 10000, 0) as w:
        for _ in range(150):
            w.step(1.0 / 60.0, 8, 4, True)
        _params, pre, _post = w.step_captured(1.0 / 60.0, 8, 4, True)
    np.savez_compressed(path, pre_bodies=pre[0], pre_contacts=pre[1], pre_joints=pre[2])
    return os.path.getsize(path)


def main():
    total = big_inputs()
    for scene, p0, p1, at in CASES:
        for solver in wire.SOLVER_NAMES:
            vel, pos = common.DEFAULT_ITERS[solver]
            if ONLY_MISSING and all(os.path.exists(os.path.join(OUT, "%s%d_%s_step%03d.npz" % (scene, p0, solver, a))) for a in at):
                continue
            with refbind.RefWorld(scene, solver, p0, p1) as world:
                for step in range(max(at) + 1):
                    if step in at:
                        params, pre, post = world.step_captured(1.0 / 60.0, vel, pos, True)
                        name = "%s%d_%s_step%03d.npz" % (scene, p0, solver, step)
                        path = os.path.join(OUT, name)
                        np.savez_compressed(
                            path,
                            params=np.array([params.solverType, params.velIters, params.posIters, params.warmStart], dtype=np.int32),
                            params_f=np.array([params.dt, params.gravity[0], params.gravity[1]], dtype=np.float32),
                            pre_bodies=pre[0], pre_contacts=pre[1], pre_joints=pre[2],
                            post_bodies=post[0], post_contacts=post[1], post_joints=post[2])
                        total += os.path.getsize(path)
                    else:
                        world.step(1.0 / 60.0, vel, pos, True)
    # broad phase + refit captures: state at s2UpdateBroadPhasePairs entry, the
    # contacts it created, and the shapes / origins before and after Stage 4
    for scene, p0, p1, at in (("mixed", 24, 0, (0, 30, 90)), ("pyramid", 8, 0, (0, 1)), ("tumbler", 60, 0, (0, 40)),
                              ("far_ragdoll_pile", 0, 0, (0, 25)), ("far_pyramid", 0, 0, (0, 12)), ("card_house", 0, 0, (0,))):
        if ONLY_MISSING and all(os.path.exists(os.path.join(OUT, "bp_%s%d_step%03d.npz" % (scene, p0, a))) for a in at):
            continue
        with refbind.RefWorld(scene, "TGS_Soft", p0, p1) as world:
            for step in range(max(at) + 1):
                shapes_before, origins_before = world.pack_shapes()
                _params, pre, post = world.step_captured(1.0 / 60.0, 8, 4, True)
                if step in at:
                    shapes_after, origins_after = world.pack_shapes()
                    bp_shapes, moved, existing, created = refbind.broadphase_capture()
                    path = os.path.join(OUT, "bp_%s%d_step%03d.npz" % (scene, p0, step))
                    np.savez_compressed(path, bodies_entry=pre[0], joints=pre[2], bp_shapes=bp_shapes, moved=moved, existing=existing,
                                        created=created, bodies_solved=post[0], shapes_before=shapes_before,
                                        origins_before=origins_before, shapes_after=shapes_after, origins_after=origins_after)
                    total += os.path.getsize(path)
    # narrow phase captures: input of Stage 3 (end of Stage 2) and its output (solver entry)
    for scene, p0, at in (("shapes_zoo", 40, (30, 90, 150)), ("mixed", 24, (40, 100)), ("pyramid", 8, (0, 2, 30)), ("circle_pile", 20, (60,)),
                          ("arch", 0, (1, 40)), ("card_house", 0, (1, 25)), ("far_pyramid", 0, (20,)), ("far_stack", 0, (25,)),
                          ("far_ragdoll_pile", 0, (50,)), ("overlap_recovery", 0, (0, 4)), ("high_mass_ratio", 2, (110,))):
        if ONLY_MISSING and all(os.path.exists(os.path.join(OUT, "np_%s%d_step%03d.npz" % (scene, p0, a))) for a in at):
            continue
        with refbind.RefWorld(scene, "TGS_Soft", p0, 0) as world:
            for step in range(max(at) + 1):
                world.step_captured(1.0 / 60.0, 4, 2, True)
                if step in at:
                    cap = refbind.narrowphase_capture()
                    path = os.path.join(OUT, "np_%s%d_step%03d.npz" % (scene, p0, step))
                    np.savez_compressed(path, **cap)
                    total += os.path.getsize(path)
    # world chains (stage 3 -> solve -> stage 4 resident): the input of stage 3 at step `start` and at
    # step `start + k`, in a window where the reference's stage 1 created no contact (pair creation stays with the caller
    # of s2amd_world_step).  Separations inside the window are part of the chain.
    for scene, p0, solver, start, k in (("pyramid", 8, "TGS_Soft", 30, 3), ("mixed", 24, "PGS", 101, 3), ("shapes_zoo", 40, "TGS_Sticky", 154, 3),
                                        ("circle_pile", 20, "XPBD", 81, 3), ("mixed", 24, "Jacobi", 104, 3),
                                        ("shapes_zoo", 40, "PGS_NGS_Block", 202, 3), ("joint_grid", 6, "TGS_NGS", 5, 3),
                                        # k = 0: an input only (the tumbler never goes three steps without creating a contact); the GPU
                                        # test runs the whole loop on it, pair creation included, against the oracle chain
                                        ("tumbler", 60, "TGS_Soft", 100, 0),
                                        # the reference's edge-case samples as whole-step chains
                                        ("arch", 0, "TGS_Soft", 40, 3), ("card_house", 0, "SoftStep", 30, 3), ("far_pyramid", 0, "TGS_Soft", 40, 3),
                                        ("ragdoll", 0, "PGS_NGS_Block", 80, 3), ("far_ragdoll_pile", 0, "PGS_Soft", 87, 3),
                                        ("high_mass_ratio", 1, "PGS_NGS", 60, 3), ("overlap_recovery", 0, "TGS_Sticky", 30, 3),
                                        ("far_stack", 0, "PGS", 40, 3)):
        vel, pos = common.DEFAULT_ITERS[solver]
        path = os.path.join(OUT, "world_%s%d_%s_step%03d_k%d.npz" % (scene, p0, solver, start, k))
        if os.environ.get("S2_GOLDEN_ONLY_MISSING") and os.path.exists(path):
            continue
        with refbind.RefWorld(scene, solver, p0, 6 if scene == "joint_grid" else 0) as world:
            for _ in range(start):
                world.step(1.0 / 60.0, vel, pos, True)
            caps = []
            for i in range(k + 1):
                params, pre, _post = world.step_captured(1.0 / 60.0, vel, pos, True)
                cap = refbind.narrowphase_capture()
                created = len(refbind.broadphase_capture()[3])
                assert i == 0 or created == 0, "%s/%s: the reference created a contact inside the window" % (scene, solver)
                caps.append({"bodies": cap["bodies"], "contacts": cap["contacts_pre"], "joints": pre[2], "shapes": cap["shapes"],
                             "pairs": cap["pairs_pre"], "origins": cap["origins"]})
            arrays = {key: caps[0][key] for key in caps[0]}
            arrays.update({"out_" + key: caps[k][key] for key in caps[k]})
            np.savez_compressed(path, steps=np.array([k], dtype=np.int32),
                                params=np.array([params.solverType, params.velIters, params.posIters, params.warmStart], dtype=np.int32),
                                params_f=np.array([params.dt, params.gravity[0], params.gravity[1]], dtype=np.float32), **arrays)
            total += os.path.getsize(path)
    print("wrote fixtures, %.1f KiB total" % (total / 1024.0))


if __name__ == "__main__":
    main()
