"""One-off: tests/test_fuzz.py's random worlds with seeds the suite does not hold (all ten solvers per world, through the LDS
groups, the global path and the strip paths, with and without joints).    python tools/fuzz_soak.py [first_seed] [count]"""
import os
import sys
import time
import traceback

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from tests import test_fuzz as T  # noqa: E402
from tests import test_gpu_generic as G  # noqa: E402


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    runs = failed = 0
    t0 = time.time()
    for seed in range(first, first + count):
        for name, fn, args in (("groups", T.test_gpu_equals_oracle_on_random_worlds, (seed, 1)), ("global", T.test_gpu_equals_oracle_on_random_worlds, (seed, 0)),
                               ("strips", T.test_gpu_equals_oracle_on_random_worlds_through_strips, (seed, 0)),
                               ("strips+joints", T.test_gpu_equals_oracle_on_random_worlds_through_strips, (seed, 6)),
                               ("interpreter", G.test_perturbed_piles_with_joints_through_the_op_interpreter, (seed, 0)),
                               ("interpreter+joints", G.test_perturbed_piles_with_joints_through_the_op_interpreter, (seed, 30))):
            runs += 1
            try:
                fn(*args)
            except Exception:
                failed += 1
                print("FAILED %s seed %d\n%s" % (name, seed, traceback.format_exc()[-1500:]))
    print("%d runs, %d failed, %.0f s" % (runs, failed, time.time() - t0))
    return 1 if failed else 0


if __name__ == "__main__":
    sys.exit(main())
