#!/usr/bin/env python3
"""The few numbers of a bench.py line one looks at first.  Usage: python tools/bench_summary.py line.json"""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print("value %.4g %s  %.4f ms/step  launches %s  dominant %.2f us  frac %.3f  traffic %s" % (
    d["value"], d["unit"], d["ms_per_step"], d["config"].get("kernel_launches_per_step"), r["avg_launch_us"], r["frac"], r.get("traffic")))
if "value_fast" in d:
    f = d["value_fast"]
    if "error" in f:
        print("value_fast: ERROR", f["error"])
    else:
        print("value_fast %.4g  %.4f ms/step  dominant %.2f us  frac %.3f  one step vs bit-exact: %s" % (
            f["value"], f["ms_per_step"], f["roofline"]["avg_launch_us"], f["roofline"]["frac"], {k: v for k, v in f["one_step_vs_bit_exact_build"].items() if k != "from"}))
for k in ("whole_step", "churn"):
    if k in d:
        print(k, json.dumps({a: b for a, b in d[k].items() if not isinstance(b, (dict, list))})[:600])
if "configs" in d:
    for k, v in d["configs"].items():
        print(k, "%.4f ms/step" % v["ms_per_step"], "launches", v.get("config", {}).get("kernel_launches_per_step"), "frac", v.get("roofline", {}).get("frac"))
if "cpu_baseline" in d:
    print("cpu_baseline", d["cpu_baseline"])
