#!/bin/bash
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
mkdir -p gpurun_out/dbg
echo skip suite
S2AMD_DEBUG_PREP=1 timeout 600 python tools/churn_bench.py --world tumbler --steps 120 --trace > gpurun_out/dbg/t.json 2> gpurun_out/dbg/t.trace
grep "reason:" gpurun_out/dbg/t.trace | sed 's/#[0-9]*: [0-9]* potential.*reason/reason/' | sort | uniq -c | sort -rn | head
grep "^step" gpurun_out/dbg/t.trace | awk 'NR%10==0' | cut -c1-210
python -c "
import json; d=json.load(open('gpurun_out/dbg/t.json'))
print({a: round(b,3) for a,b in d['all_steps'].items()}); print('rebuild steps', d['steps_that_rebuilt_the_structure'], 'placed', d['contacts_placed_without_rebuild'])"
