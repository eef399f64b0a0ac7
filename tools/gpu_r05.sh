#!/bin/bash
# The round's GPU calls, one stage per call: tools/gpu_r05.sh <stage> (run by gpurun from the repo root; everything lands under
# gpurun_out/r5/<stage>/).  Stages: fast (tolerance-mode build: error measurement, its tests, bench line), suite (the whole -m gpu
# suite), churn (wreck-200 and Tumbler loops with the structure builds' phase times), bench (the driver's command), profile
# (tools/profile_r05.sh [part]), tables (solver table, configs 3 / 3b / 4).
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
stage=${1:-suite}
O=gpurun_out/r5/$stage
mkdir -p $O
case $stage in
fast)
  timeout 600 python tools/fast_mode_error.py --out $O/fast_mode_error.json > $O/fast_mode_error.log 2>&1; echo "error tool rc=$?"
  timeout 900 python -m pytest tests/test_gpu_fast.py -q --tb=line -p no:cacheprovider > $O/test_gpu_fast.log 2>&1; echo "fast tests rc=$?"; tail -15 $O/test_gpu_fast.log
  timeout 600 python bench.py --steps 200 --warmup 60 --no-cpu --no-extras > $O/bench_headline.json 2> $O/bench_headline.err; echo "bench rc=$?"
  python tools/bench_summary.py $O/bench_headline.json
  ;;
suite)
  timeout 2400 python -m pytest tests -q -m gpu -x -p no:cacheprovider > $O/gpu_suite.log 2>&1; echo "gpu suite rc=$?"; tail -8 $O/gpu_suite.log
  ;;
churn)
  S2AMD_DEBUG_PREP=1 timeout 300 python tools/churn_bench.py --world wreck --steps 240 --trace > $O/churn_wreck200.json 2> $O/churn_wreck200.trace; echo "wreck rc=$?"
  S2AMD_DEBUG_PREP=1 timeout 300 python tools/churn_bench.py --world tumbler --steps 200 --trace > $O/churn_tumbler.json 2> $O/churn_tumbler.trace; echo "tumbler rc=$?"
  ;;
scan)
  # which wrecking-ball worlds meet a contact that fits nowhere in their strips (overflow positions, sliced steps)
  for base in 100 60; do for seed in 0 1 2 3 4 5 6 7; do
    timeout 120 python tools/churn_bench.py --world wreck --base $base --seed $seed --steps 160 > $O/wreck_${base}_$seed.json 2> $O/wreck_${base}_$seed.err
    python -c "import json,sys; d=json.load(open('$O/wreck_${base}_$seed.json')); print('base $base seed $seed', d['overflow']['steps_run_sliced'], d['overflow']['most_contacts_waiting'], 'rebuilds', d['steps_that_rebuilt_the_structure'], 'async', d['structure_builds_by_the_worker_thread'], 'over1ms', d['steps_over_1ms'], d['slowest_steps_ms'][:3])"
  done; done
  ;;
bench)
  timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; python tools/bench_summary.py $O/bench.json
  ;;
profile)
  bash tools/profile_r05.sh ${2:-all}
  ;;
tables)
  # the per-solver / per-config numbers committed as profiles/r05_*: all ten solvers at base 200, configs 3 / 3b / 4
  timeout 600 python tools/solver_table.py --steps 100 > $O/r05_solver_table.jsonl 2> $O/solver_table.err; echo "solver table rc=$?"
  timeout 300 python tools/solver_table.py --world joint_grid --base 100 --solvers PGS_NGS,TGS_Soft,PGS_NGS_Block --steps 100 >> $O/r05_solver_table.jsonl 2>> $O/solver_table.err
  timeout 300 python tools/config3_bench.py Jacobi 2> $O/config3.err | tail -1 > $O/r05_config3_tumbler_jacobi.json
  timeout 300 python tools/config3_bench.py TGS_Soft 2>> $O/config3.err | tail -1 > $O/r05_config3b_tumbler_tgs_soft.json
  cat $O/r05_solver_table.jsonl $O/r05_config3_tumbler_jacobi.json $O/r05_config3b_tumbler_tgs_soft.json | cut -c1-200
  ;;
*)
  echo "unknown stage $stage"; exit 2;;
esac
