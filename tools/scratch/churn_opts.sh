#!/bin/bash
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
mkdir -p gpurun_out/dbg
for opts in "--opt strip_patience=0 --opt strip_retry=0" "--opt strip_retry=0"; do
  S2AMD_DEBUG_PLACE=1 S2AMD_DEBUG_PREP=1 timeout 600 python tools/churn_bench.py --trace $opts > gpurun_out/dbg/c.json 2> gpurun_out/dbg/c.err
  python - gpurun_out/dbg/c.json "$opts" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    print("churn %-46s rebuild_steps %d persistent %d/%d median %.3f mean %.3f over1ms %d over2ms %d slowest %s worker %s" % (sys.argv[2] or "(default)", d["steps_that_rebuilt_the_structure"], d["steps_on_persistent_kernel"], d["steps"], d["churn_steps_median"]["step_ms"], d["all_steps"]["step_ms"], d["steps_over_1ms"], d["steps_over_2ms"], d["slowest_steps_ms"][:8], d["structure_builds_by_the_worker_thread"]))
except Exception as e:
    print("churn", sys.argv[2], "FAILED", e)
PY
  grep "no strip home\|no free round\|reason: [a-z]" gpurun_out/dbg/c.err | cut -c1-160
  grep "^step" gpurun_out/dbg/c.err | awk '{ if ($3+0 > 1.0) print }' | cut -c1-200 | head -20
done
