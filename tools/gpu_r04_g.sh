#!/bin/bash
# round 4, GPU call G: s2Solve_PGS_Soft and s2Solve_SoftStep on the 512-thread strip kernel -- parity, then their lines of the solver table against wide=0
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r04g
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_strips.py tests/test_gpu_world.py -q -x -m gpu -k "pgs_soft or softstep or kernels_agree or PGS_Soft or SoftStep" > $OUT/soft_tests.log 2>&1; echo "tests rc=$?" | tee $OUT/summary.txt
tail -5 $OUT/soft_tests.log | tee -a $OUT/summary.txt
for w in 1 0; do
  timeout 600 python tools/solver_table.py --solvers PGS_Soft,SoftStep,TGS_Soft --steps 200 --opt wide=$w > $OUT/table_wide$w.json 2> $OUT/table_wide$w.err
  echo "wide=$w" | tee -a $OUT/summary.txt; tail -4 $OUT/table_wide$w.err | tee -a $OUT/summary.txt; cut -c1-160 $OUT/table_wide$w.json | tee -a $OUT/summary.txt
done
