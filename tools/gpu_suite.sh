#!/bin/bash
# the whole -m gpu suite, log under gpurun_out/suite/
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/suite
mkdir -p $OUT
timeout 2400 python -m pytest tests -q -m gpu > $OUT/gpu_suite.log 2>&1; echo "gpu suite rc=$?" | tee $OUT/summary.txt
tail -15 $OUT/gpu_suite.log | tee -a $OUT/summary.txt
