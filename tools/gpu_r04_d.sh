#!/bin/bash
# round 4, GPU call D: structure builds in a worker thread (wreck-200 churn with and without), the product drop-in's interposed API, the whole GPU suite
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r04d
mkdir -p $OUT
rm -f $OUT/summary.txt
for opts in "" "--opt async_build=0" "--opt strip_patience=1"; do
  name=churn$(echo "$opts" | tr -dc 'a-z0-9_=' )
  timeout 600 python tools/churn_bench.py $opts > $OUT/$name.json 2> $OUT/$name.err
  python - $OUT/$name.json "$opts" <<'PY' | tee -a gpurun_out/r04d/summary.txt
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    print("churn %-28s rebuild_steps %d persistent %d/%d median %.3f mean %.3f over1ms %d over2ms %d slowest %s worker %s" % (sys.argv[2] or "(default)", d["steps_that_rebuilt_the_structure"], d["steps_on_persistent_kernel"], d["steps"], d["churn_steps_median"]["step_ms"], d["all_steps"]["step_ms"], d["steps_over_1ms"], d["steps_over_2ms"], d["slowest_steps_ms"][:6], d["structure_builds_by_the_worker_thread"]))
except Exception as e:
    print("churn", sys.argv[2], "FAILED", e)
PY
done
timeout 600 python bench.py --no-extras --no-cpu > $OUT/bench.json 2> $OUT/bench.err; python -c "
import json; d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1]); print('bench ms/step %.4f value %.3e launches %d kernel_us %.1f' % (d['ms_per_step'], d['value'], d['config']['kernel_launches_per_step'], d['roofline']['avg_launch_us']))" | tee -a $OUT/summary.txt
timeout 2400 python -m pytest tests -q -m gpu > $OUT/gpu_suite.log 2>&1; echo "gpu suite rc=$?" | tee -a $OUT/summary.txt
tail -12 $OUT/gpu_suite.log | tee -a $OUT/summary.txt
