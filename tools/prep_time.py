import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
from solver2d_amd import hip, synthetic, wire
pre = synthetic.pyramid(200)
p = wire.StepParams.make("TGS_Soft", 1/60, 8, 4, True)
with hip.Solver(0) as g:
    for strips in (1, 0):
        g.set_option("strips", strips)
        w = [a.copy() for a in pre]
        t0 = time.perf_counter(); g.solve(p, *w); t1 = time.perf_counter()
        st = g.stats()
        print("strips", strips, "first solve ms", 1e3*(t1-t0), "hostPrepMs", st["hostPrepMs"], "deviceMs", st["deviceMs"])
        ts = []
        for i in range(5):
            t0 = time.perf_counter(); g.solve(p, *w); ts.append(1e3*(time.perf_counter()-t0))
        print("   repeat solve ms (graph unchanged)", min(ts), g.stats()["hostPrepMs"])
        # change the graph slightly: deactivate one contact -> structure rebuild
        w[1]["pointCount"][5] = 0
        t0 = time.perf_counter(); g.solve(p, *w); t1 = time.perf_counter()
        print("   solve after a graph change ms", 1e3*(t1-t0), "hostPrepMs", g.stats()["hostPrepMs"])
