#!/bin/bash
# round 4, GPU call L: the numbers that changed after call K (whole-step leg behind the search; tail slack; near-path option): churn runs, solver table, the full bench line
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r04l
mkdir -p $OUT
for opts in "" "--opt strip_adopt=0"; do
  name=r04_churn_wreck200$(echo "$opts" | tr -dc 'a-z0-9_=' | sed 's/optstrip_adopt=0/_no_adoption/')
  timeout 600 python tools/churn_bench.py $opts > $OUT/$name.json 2> $OUT/$name.err
  python - $OUT/$name.json "$opts" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    print("churn %-22s rebuild_steps %d persistent %d/%d median %.3f mean %.3f over1ms %d over2ms %d slowest %s" % (sys.argv[2] or "(default)", d["steps_that_rebuilt_the_structure"], d["steps_on_persistent_kernel"], d["steps"], d["churn_steps_median"]["step_ms"], d["all_steps"]["step_ms"], d["steps_over_1ms"], d["steps_over_2ms"], d["slowest_steps_ms"][:6]))
except Exception as e:
    print("churn", sys.argv[2], "FAILED", e)
PY
done
timeout 900 python tools/churn_bench.py --world tumbler --steps 200 > $OUT/r04_churn_tumbler.json 2> $OUT/churn_tumbler.err; python -c "
import json; d=json.load(open('$OUT/r04_churn_tumbler.json')); print('tumbler loop: mean step %.3f median churn %.3f rebuild steps %d of %d' % (d['all_steps']['step_ms'], d['churn_steps_median']['step_ms'], d['steps_that_rebuilt_the_structure'], d['steps']))"
timeout 600 python tools/solver_table.py > $OUT/r04_solver_table.jsonl 2> $OUT/solver_table.err; cut -c1-120 $OUT/r04_solver_table.jsonl
timeout 600 python tools/solver_table.py --world joint_grid --base 100 --solvers PGS_NGS,PGS_NGS_Block,TGS_Soft >> $OUT/r04_solver_table.jsonl 2>> $OUT/solver_table.err
timeout 600 python tools/config3_bench.py TGS_Soft > $OUT/r04_config3b_tumbler_tgs_soft.json 2> $OUT/config3b.err; head -c 700 $OUT/r04_config3b_tumbler_tgs_soft.json; echo
timeout 1500 python bench.py > $OUT/r04_bench_final.json 2> $OUT/bench_final.err; python -c "
import json; d=json.loads(open('$OUT/r04_bench_final.json').read().strip().splitlines()[-1]); print('bench ms/step %.4f value %.3e launches %d kernel_us %.1f frac %.3f cpu %.3e whole_step %.4f / %.4f c3 %.4f c4 %.4f c5 %.4f' % (d['ms_per_step'], d['value'], d['config']['kernel_launches_per_step'], d['roofline']['avg_launch_us'], d['roofline']['frac'], d['cpu_baseline']['value'], d['whole_step']['whole_step_ms'], d['whole_step']['whole_step_with_pair_query_ms'], d['configs']['3_tumbler']['ms_per_step'], d['configs']['4_joint_grid']['ms_per_step'], d['configs']['5_one_gpu']['ms_per_step']))"
timeout 2400 python -m pytest tests -q -m gpu > $OUT/gpu_suite.log 2>&1; echo "gpu suite rc=$?"; tail -3 $OUT/gpu_suite.log | cut -c1-200
