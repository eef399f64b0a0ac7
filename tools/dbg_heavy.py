import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from solver2d_amd import hip, synthetic, wire
from tests import common, oraclebind
from tests.test_gpu_incremental import _with_spare_slots, _artificial_contact
from tests.test_gpu_parity import gpu_vs_oracle_loose
for inc in (1, 0):
    rng = np.random.default_rng(21)
    params = wire.StepParams.make("Jacobi", 1.0 / 60.0, 4, 2, True)
    pre = _with_spare_slots(synthetic.platform(10, layers=2), 64)
    n0 = len(pre[1]) - 64
    with hip.Solver(0) as s:
        s.set_option("groups", 0); s.set_option("strips", 0); s.set_option("incremental", inc)
        state = gpu_vs_oracle_loose(s, params, pre, "step 0")
        top = [i for i in range(12, 22)]
        for n in range(8):
            c = _artificial_contact(rng, state[0], pre[1][1], set())
            c["bodyA"], c["bodyB"] = 1, top[n]
            state[1][n0 + n] = c
            try:
                state = gpu_vs_oracle_loose(s, params, state, "step %d" % (n + 1))
                print("inc", inc, "step", n + 1, "ok", {k: s.stats()[k] for k in ("structureBuilds", "placedContacts", "constraintCount", "kernelLaunches")}, "finite", bool(np.isfinite(state[0]["position"]).all()))
            except AssertionError as e:
                print("inc", inc, "step", n + 1, "FAIL", str(e)[:300])
                break
