import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from solver2d_amd import hip, synthetic, wire
from tests import common
from tests.test_gpu_strips import _pyramid_with_free_bodies, _touch, _box, gpu_vs_oracle_loose
base = 100
pre = _pyramid_with_free_bodies(base, 0, 2)
params = wire.StepParams.make("TGS_Soft", 1.0 / 60.0, 8, 4, True)
template = int(np.flatnonzero(pre[1]["pointCount"] == 2)[len(pre[1]) // 3])
for col in (49, 50):
    with hip.Solver(0) as s:
        s.set_option("strip_patience", 0)
        state = common.copy3(pre)
        for step in range(2):
            state = gpu_vs_oracle_loose(s, params, state, "w")
        print("before", {k: s.stats()[k] for k in ("structureBuilds", "persistent", "pairLanes", "placedContacts", "persistFallbacks", "kernelLaunches")}, flush=True)
        _touch(state[1], len(pre[1]) - 2, template, _box(base, 20, col), _box(base, 20, col + 3))
        for step in range(3):
            try:
                state = gpu_vs_oracle_loose(s, params, state, "far %d %d" % (col, step))
            except AssertionError as e:
                print("PARITY FAIL", str(e)[:300])
            print(col, step, {k: s.stats()[k] for k in ("structureBuilds", "persistent", "pairLanes", "placedContacts", "persistFallbacks", "kernelLaunches")}, flush=True)
