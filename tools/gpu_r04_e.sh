#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r04e
mkdir -p $OUT
S2AMD_DEBUG_PREP=1 timeout 600 python tools/churn_bench.py --trace > $OUT/churn.json 2> $OUT/churn_trace.txt
grep -n "^step" $OUT/churn_trace.txt | awk '{ if ($3+0 > 2.0) print }' | head -40
