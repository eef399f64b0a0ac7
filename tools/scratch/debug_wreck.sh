#!/bin/bash
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
mkdir -p gpurun_out/dbg
S2AMD_DEBUG_PLACE=1 S2AMD_DEBUG_PREP=1 timeout 600 python -m pytest tests/test_gpu_world.py -x -q -m gpu -k "wrecking_ball_world_loop and SoftStep" > gpurun_out/dbg/wreck.log 2>&1
grep -n "no strip home\|no free round\|reason: [a-z]\|strips;\|Error\|passed\|failed" gpurun_out/dbg/wreck.log | cut -c1-220 | tail -40
for opt in "strip_adopt=0" "wide=0" "strip_slack=0"; do
  echo "== $opt"; S2AMD_OPTIONS=$opt timeout 600 python -m pytest tests/test_gpu_world.py -x -q -m gpu -k "wrecking_ball_world_loop and SoftStep" 2>&1 | tail -1
done
