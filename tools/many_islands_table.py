"""Many equal islands under every solver: ms per resident step, launches, groups, strips (a survey for cliffs: who sweeps an island depends
on how many of its size there are).   python tools/many_islands_table.py [count 512] [base 40]"""
import sys, os, time, json
sys.path.insert(0, os.getcwd())
import numpy as np
from solver2d_amd import hip, synthetic, wire
count = int(sys.argv[1]) if len(sys.argv) > 1 else 512
base = int(sys.argv[2]) if len(sys.argv) > 2 else 40
bodies, contacts, joints = synthetic.pyramid(base, count=count)
for name in wire.SOLVER_NAMES:
    vel, pos = (8, 4) if name in ("TGS_Soft", "SoftStep") else (4, 2)
    params = wire.StepParams.make(name, 1.0 / 60.0, vel, pos, True)
    with hip.Solver(0) as gpu:
        gpu.set_option("strip_patience", 0)
        for kv in sys.argv[3:]:
            k, v = kv.split("=")
            gpu.set_option(k, int(v))
        gpu.upload(bodies, contacts, joints)
        gpu.save_bodies()
        gpu.set_option("async", 1)
        for _ in range(6):
            gpu.restore_bodies(); gpu.step_resident(params)
        gpu.synchronize()
        t0 = time.perf_counter()
        n = 40
        for _ in range(n):
            gpu.restore_bodies(); gpu.step_resident(params)
        gpu.synchronize()
        ms = (time.perf_counter() - t0) * 1e3 / n
        gpu.set_option("async", 0)
        gpu.restore_bodies(); gpu.step_resident(params)
        st = gpu.stats()
        print("%d x base-%d %-14s %.3f ms/step  launches %d groups %d strips %d persistent %d" % (count, base, name, ms, st["kernelLaunches"], st["groupCount"], st["stripCount"], st["persistent"]), flush=True)
