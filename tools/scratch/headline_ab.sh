#!/bin/bash
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
mkdir -p gpurun_out/dbg
for i in 1 2; do
timeout 600 python bench.py --no-extras --no-cpu > gpurun_out/dbg/bench.json 2> gpurun_out/dbg/bench.err; python -c "
import json; d=json.loads(open('gpurun_out/dbg/bench.json').read().strip().splitlines()[-1]); print('bench ms/step %.4f value %.3e launches %d kernel_us %.1f frac %.3f' % (d['ms_per_step'], d['value'], d['config']['kernel_launches_per_step'], d['roofline']['avg_launch_us'], d['roofline']['frac']))"
done
timeout 600 python tools/solver_table.py --solvers PGS_Soft,SoftStep,TGS_Soft --steps 200 2> gpurun_out/dbg/table.err | cut -c1-150
timeout 900 python -m pytest tests/test_gpu_strips.py tests/test_gpu_selfstrips.py -q -x -m gpu 2>&1 | tail -3
