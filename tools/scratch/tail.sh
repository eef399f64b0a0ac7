#!/bin/bash
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
mkdir -p gpurun_out/dbg
timeout 600 python tools/config3_bench.py TGS_Soft 2>/dev/null | head -c 400; echo
timeout 600 python tools/config3_bench.py SoftStep 2>/dev/null | head -c 400; echo
timeout 900 python tools/churn_bench.py --world tumbler --steps 200 > gpurun_out/dbg/t.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/dbg/t.json')); print('tumbler loop: mean step %.3f median churn %.3f rebuild steps %d of %d solve_device %.3f' % (d['all_steps']['step_ms'], d['churn_steps_median']['step_ms'], d['steps_that_rebuilt_the_structure'], d['steps'], d['all_steps']['solve_device_ms']))"
timeout 2400 python -m pytest tests -q -m gpu -x -k "tumbler or hub or tail or world or dropin or incremental" 2>&1 | tail -3
