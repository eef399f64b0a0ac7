#!/bin/bash
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
mkdir -p gpurun_out/dbg
for opts in ""; do
  S2AMD_DEBUG_PREP=1 timeout 600 python tools/churn_bench.py --trace $opts > gpurun_out/dbg/c.json 2> gpurun_out/dbg/c.trace
  grep "^step" gpurun_out/dbg/c.trace | awk '{ if ($3+0 > 1.0 && NR > 2) print }' | cut -c1-170
  python -c "
import json; d=json.load(open('gpurun_out/dbg/c.json')); print('persistent', d['steps_on_persistent_kernel'], 'median', round(d['churn_steps_median']['step_ms'],3), 'mean', round(d['all_steps']['step_ms'],3), 'over1', d['steps_over_1ms'], d['structure_builds_by_the_worker_thread'])"
done
timeout 900 python -m pytest tests -q -m gpu -x -k "world or async or dropin" 2>&1 | tail -2
