#!/bin/bash
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
mkdir -p gpurun_out/dbg
S2AMD_DEBUG_PREP=1 S2AMD_DEBUG_PLACE=1 timeout 600 python tools/churn_bench.py --world tumbler --steps 120 --trace > gpurun_out/dbg/t.json 2> gpurun_out/dbg/t.trace
grep "reason:" gpurun_out/dbg/t.trace | sed 's/#[0-9]*: [0-9]* potential.*reason/reason/' | sort | uniq -c | sort -rn | head
grep "no strip home\|no free round\|no free colour" gpurun_out/dbg/t.trace | sed 's/([0-9]*, [0-9]*)/(a, b)/' | sort | uniq -c | sort -rn | head -5
grep "^step" gpurun_out/dbg/t.trace | awk 'NR%10==0' | cut -c1-210
grep "prep " gpurun_out/dbg/t.trace | awk '{k=$3" "$4" "$5; s[k]+=$(NF-1); n[k]++} END{for (k in s) printf "%-30s total %.1f ms over %d\n", k, s[k], n[k]}' | sort -k3 -rn | head -14
