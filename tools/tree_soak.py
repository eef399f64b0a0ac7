"""Soak of the device trees through the product binding in check mode (shim/s2_amd_binding.c: S2AMD_CHECK_TREES): every scene of the
corpus under a solver in turn, the binding replaying every step into the reference's own trees beside the device and comparing the
order of every pair query and the trees node for node.  Prints one line per run; exit code 1 on any difference.

    python tools/tree_soak.py [steps] [big]     (needs oracle/_ref/libs2ref.so and a GPU; big: full-size worlds -- segments longer than
                                                 a workgroup, the whole tree rebuilt every step)"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from solver2d_amd import hip, wire  # noqa: E402
from tests import common, refbind  # noqa: E402

RUNS = [("pyramid", 40), ("mixed", 24), ("mixed", 60), ("tumbler", 150), ("tumbler", 1200), ("shapes_zoo", 40), ("shapes_zoo", 160), ("circle_pile", 20),
        ("circle_pile", 60), ("card_house", 0), ("arch", 0), ("far_ragdoll_pile", 0), ("far_pyramid", 0), ("ragdoll_stress", 0), ("rush", 0),
        ("confined", 25), ("friction_ramp", 0), ("double_domino", 0), ("warm_start_energy", 0), ("circle_stack", 0), ("high_mass_ratio", 1),
        ("overlap_recovery", 0), ("vertical_stack", 30), ("multi_pyramid", 6), ("bridge", 40)]


BIG = [("pyramid", 200), ("tumbler", 10000), ("circle_pile", 140), ("shapes_zoo", 3000), ("pyramid", 120), ("mixed", 400)]
BIG_SOLVERS = ["PGS_NGS_Block", "Jacobi", "TGS_Soft", "PGS", "XPBD", "SoftStep"]


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    big = len(sys.argv) > 2 and sys.argv[2] == "big"
    L = refbind.lib()
    L.s2ref_use_amd_world.argtypes = [ctypes.c_char_p, ctypes.c_int]
    L.s2ref_use_amd_world.restype = ctypes.c_int
    out = (ctypes.c_long * 4)()
    bad = 0
    for k, (scene, p0) in enumerate(BIG if big else RUNS):
        solver = BIG_SOLVERS[k] if big else wire.SOLVER_NAMES[k % len(wire.SOLVER_NAMES)]
        vel, pos = common.DEFAULT_ITERS[solver]
        n = steps if scene != "ragdoll_stress" else max(steps, 500)
        with refbind.RefWorld(scene, solver, p0, 10 if scene == "multi_pyramid" else 0) as w:
            assert L.s2ref_use_amd_world(hip.LIB_PATH.encode(), 0) == 0
            L.s2ref_world_device_pairs(1)
            L.s2amdBinding_DeviceTrees(1, 1)
            L.s2amdBinding_TreeCheck(out)
            try:
                for _ in range(n):
                    w.step(1.0 / 60.0, vel, pos, True)
                err = L.s2ref_replace_error()
                L.s2amdBinding_TreeCheck(out)
            finally:
                L.s2amdBinding_DeviceTrees(1, 0)
                L.s2ref_world_device_pairs(0)
                L.s2ref_use_amd_world(None, 0)
        q, qd, t, td = list(out)
        print("%-18s %5d %-14s %4d steps: %5d queries compared, %d ordered differently; %5d trees compared, %d differ; error %d" % (scene, p0, solver, n, q, qd, t, td, err),
              flush=True)
        bad += qd + td + (1 if err else 0)
    print("tree soak: %s" % ("no difference" if bad == 0 else "%d DIFFERENCES" % bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
