#!/bin/bash
# round 4, GPU call H: why placement fails in the wreck-200 churn (S2AMD_DEBUG_PLACE), with the per-step trace
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r04h
mkdir -p $OUT
S2AMD_DEBUG_PLACE=1 S2AMD_DEBUG_PREP=1 timeout 600 python tools/churn_bench.py --trace > $OUT/churn.json 2> $OUT/churn_trace.txt
grep -c "" $OUT/churn_trace.txt
grep "no strip home\|no free round\|reason:" $OUT/churn_trace.txt | sed 's/([0-9]*, [0-9]*)/(a, b)/; s/strips [0-9-]*, [0-9-]*/strips x, y/; s/group [0-9]*/group g/; s/mask [0-9a-f]*/mask m/; s/#[0-9]*: [0-9]* potential.*reason/reason/' | sort | uniq -c | sort -rn | head -30
