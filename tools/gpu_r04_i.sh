#!/bin/bash
# round 4, GPU call I: bodies that join an island move to the strip they touch (no rebuild) -- tests, then the wreck-200 churn
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r04i
mkdir -p $OUT
S2AMD_DEBUG_PLACE=1 timeout 900 python -m pytest tests/test_gpu_strips.py -q -x -m gpu -k "join or adopt or seam or never_faced" > $OUT/join_tests.log 2>&1; echo "tests rc=$?" | tee $OUT/summary.txt
tail -25 $OUT/join_tests.log | cut -c1-300 | tee -a $OUT/summary.txt
for opts in "" "--opt strip_adopt=0"; do
  name=churn$(echo "$opts" | tr -dc 'a-z0-9_=' )
  S2AMD_DEBUG_PLACE=1 S2AMD_DEBUG_PREP=1 timeout 600 python tools/churn_bench.py --trace $opts > $OUT/$name.json 2> $OUT/$name.trace
  python - $OUT/$name.json "$opts" <<'PY' | tee -a gpurun_out/r04i/summary.txt
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    print("churn %-22s rebuild_steps %d persistent %d/%d median %.3f mean %.3f over1ms %d over2ms %d slowest %s" % (sys.argv[2] or "(default)", d["steps_that_rebuilt_the_structure"], d["steps_on_persistent_kernel"], d["steps"], d["churn_steps_median"]["step_ms"], d["all_steps"]["step_ms"], d["steps_over_1ms"], d["steps_over_2ms"], d["slowest_steps_ms"][:6]))
except Exception as e:
    print("churn", sys.argv[2], "FAILED", e)
PY
  grep "no strip home\|no free round\|reason: [a-z]" $OUT/$name.trace | cut -c1-160 | tee -a $OUT/summary.txt
done
timeout 600 python bench.py --no-extras --no-cpu > $OUT/bench.json 2> $OUT/bench.err; python -c "
import json; d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1]); print('bench ms/step %.4f value %.3e launches %d kernel_us %.1f' % (d['ms_per_step'], d['value'], d['config']['kernel_launches_per_step'], d['roofline']['avg_launch_us']))" | tee -a $OUT/summary.txt
timeout 600 python tools/solver_table.py --solvers PGS_Soft,SoftStep,TGS_Soft --steps 200 2> $OUT/table.err | cut -c1-150 | tee -a $OUT/summary.txt
