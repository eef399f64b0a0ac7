#!/bin/bash
# round 4, GPU call J: the whole GPU suite, then the wreck-200 churn with and without joining-body placement
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r04j
mkdir -p $OUT
timeout 2400 python -m pytest tests -q -m gpu -k "world or dropin or churn or async" > $OUT/gpu_suite.log 2>&1; echo "gpu subset rc=$?" | tee $OUT/summary.txt
tail -8 $OUT/gpu_suite.log | cut -c1-200 | tee -a $OUT/summary.txt
for opts in ""; do
  name=churn$(echo "$opts" | tr -dc 'a-z0-9_=' )
  S2AMD_DEBUG_PLACE=1 S2AMD_DEBUG_PREP=1 timeout 600 python tools/churn_bench.py --trace $opts > $OUT/$name.json 2> $OUT/$name.trace
  python - $OUT/$name.json "$opts" <<'PY' | tee -a gpurun_out/r04j/summary.txt
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    print("churn %-22s rebuild_steps %d persistent %d/%d median %.3f mean %.3f over1ms %d over2ms %d slowest %s joined %s" % (sys.argv[2] or "(default)", d["steps_that_rebuilt_the_structure"], d["steps_on_persistent_kernel"], d["steps"], d["churn_steps_median"]["step_ms"], d["all_steps"]["step_ms"], d["steps_over_1ms"], d["steps_over_2ms"], d["slowest_steps_ms"][:6], {k: v for k, v in d["joined_without_rebuild"].items() if k != "note"}))
except Exception as e:
    print("churn", sys.argv[2], "FAILED", e)
PY
  grep "no strip home\|no free round\|reason: [a-z]" $OUT/$name.trace | cut -c1-160 | tee -a $OUT/summary.txt
done
