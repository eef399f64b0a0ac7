#!/usr/bin/env python3
"""Runs bench.py once per option set given on the command line ("a=1,b=2" per argument, "-" = defaults) and
prints one compact line each: ms/step, launches/step, dominant-kernel time."""
import json
import subprocess
import sys

for spec in sys.argv[1:]:
    cmd = [sys.executable, "bench.py", "--no-cpu", "--steps", "100", "--warmup", "20"]
    if spec != "-":
        for kv in spec.split(","):
            cmd += ["--opt", kv]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    try:
        d = json.loads(out.stdout.strip().splitlines()[-1])
        print("%-40s %.4f ms/step  launches %3d  dominant %.2f us  frac %.3f" % (
            spec, d["ms_per_step"], d["config"]["kernel_launches_per_step"], d["roofline"]["avg_launch_us"], d["roofline"]["frac"]), flush=True)
    except Exception as e:
        print(spec, "FAILED", e, out.stdout[-300:], out.stderr[-600:], flush=True)
