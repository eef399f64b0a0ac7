"""One-off soak of IncrementalStrips::takeOnly: tests/test_gpu_world.py::test_rain_world_loop_on_the_op_interpreters_strips over N more seeds
(every step bit-exact against the oracle chain; r6: 160 runs, 0 mismatches, 56 placed into running strips).  tools/interpreter_strips_soak.py N"""
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
os.chdir(ROOT)
from tests import test_gpu_world as t
bad = 0; n = 0; hits = 0
import io, contextlib
for seed in range(100, 100 + int(sys.argv[1])):
    for name in ("PGS_NGS_Block", "XPBD", "TGS_Sticky", "TGS_NGS"):
        try:
            _persistent, exercised = t._rain_loop_on_the_interpreters_strips(seed, name)
            hits += 1 if exercised else 0
        except AssertionError as e:
            bad += 1
            print("MISMATCH", seed, name, str(e)[:200], flush=True)
        n += 1
print("runs", n, "mismatches", bad, "runs that placed into running strips", hits)
