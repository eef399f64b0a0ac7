"""bench.py's whole-step leg timed from step 60 and from step 200 of a new world (round 4: the strip-width search of the worker
thread runs between steps 32 and 128 and shares the device with the steps beside it).    python tools/whole_step_settle.py"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
for settle in (60, 200):
    w = bench.whole_step_leg(0, 200, 8, 4, settle, 240)
    print(os.environ.get("S2AMD_OPTIONS", "(default)"), "settle", settle, "whole_step_ms %.4f with_pair_query %.4f solver_device %.4f" % (w["whole_step_ms"], w["whole_step_with_pair_query_ms"], w["solver_device_ms"]), flush=True)
