#!/bin/bash
# round 4, GPU call F: the rocprofv3 passes (headline, config 5, FETCH_SIZE calibration), config-5 scaling rows, solver table, drop-in demo
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r04f
mkdir -p $OUT
bash tools/profile_r04.sh all > $OUT/profile.log 2>&1
tail -12 $OUT/profile.log
timeout 600 python tools/config5_scaling.py > $OUT/r04_config5_scaling.json 2> $OUT/config5_scaling.err; tail -3 $OUT/config5_scaling.err | cut -c1-300
timeout 600 python tools/solver_table.py > $OUT/r04_solver_table.jsonl 2> $OUT/solver_table.err; tail -15 $OUT/r04_solver_table.jsonl | cut -c1-200
timeout 600 python tools/solver_table.py --world joint_grid --base 100 --solvers PGS_NGS PGS_NGS_Block TGS_Soft >> $OUT/r04_solver_table.jsonl 2>> $OUT/solver_table.err
timeout 600 tools/dropin_product_demo.sh > $OUT/r04_dropin_demo.txt 2>&1; cat $OUT/r04_dropin_demo.txt | cut -c1-220
timeout 600 tools/dropin_product_demo.sh 200 40 pyramid 3 4 2 45 >> $OUT/r04_dropin_demo.txt 2>&1
timeout 600 python tools/churn_bench.py > $OUT/r04_churn_wreck200.json 2> $OUT/churn.err
