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
        buf = io.StringIO()
        try:
            with contextlib.redirect_stdout(buf):
                t.test_rain_world_loop_on_the_op_interpreters_strips(seed, name)
        except AssertionError as e:
            msg = str(e)
            if msg.startswith("(") and "rain-interpreter" not in msg and "new pairs" not in msg and "step" not in msg:
                pass  # only the "was it exercised" assertion
            else:
                bad += 1
                print("MISMATCH", seed, name, msg[:200], flush=True)
        out = buf.getvalue()
        n += 1
        if "structure" in out:
            hits += int(out.strip().split()[-1] != "0")
print("runs", n, "mismatches", bad, "runs that placed into running strips", hits)
