"""One-off: the device-pairs route of the whole-step binding against the host-pairs route (bit identity, pool slots included:
tests/test_gpu_dropin.py) on every scene of scenes.c x several solvers.    python tools/dropin_soak.py [steps]"""
import os
import sys
import time
import traceback

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from solver2d_amd import wire  # noqa: E402
from tests import test_gpu_dropin as T  # noqa: E402

SCENES = [("pyramid", 14), ("multi_pyramid", 6), ("tumbler", 120), ("mixed", 24), ("vertical_stack", 12), ("circle_pile", 16), ("shapes_zoo", 40), ("arch", 0),
          ("high_mass_ratio", 1), ("high_mass_ratio", 2), ("overlap_recovery", 0), ("card_house", 0), ("far_pyramid", 0), ("far_stack", 0),
          ("far_recovery", 0), ("far_ragdoll_pile", 0), ("far_chain", 0), ("ragdoll", 0), ("ball_and_chain", 20), ("bridge", 20)]


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 80
    runs = failed = 0
    t0 = time.time()
    for i, (scene, p0) in enumerate(SCENES):
        for k in range(3):
            solver = wire.SOLVER_NAMES[(3 * i + k * 4 + 1) % 10]
            runs += 1
            try:
                T.test_native_shim_whole_step_with_device_pairs(scene, p0, solver, steps)
            except AssertionError as e:
                if "int((pah >= 0).sum()) > 0" in str(e) or str(e).strip() == "":
                    print("note: %s/%s has no contacts" % (scene, solver))
                    continue
                failed += 1
                print("FAILED %s %d %s\n%s" % (scene, p0, solver, str(e)[:1200]))
            except Exception:
                failed += 1
                print("ERROR %s %d %s\n%s" % (scene, p0, solver, traceback.format_exc()[-1200:]))
    print("%d runs, %d failed, %.0f s" % (runs, failed, time.time() - t0))
    return 1 if failed else 0


if __name__ == "__main__":
    sys.exit(main())
