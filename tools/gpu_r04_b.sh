#!/bin/bash
# round 4, GPU call B: where the one-launch step's time goes (stamped build), A/B with the pipelined prologue, direct launches against graph replay, the whole GPU suite
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r04b
mkdir -p $OUT
rm -f $OUT/summary.txt
bench_line() {
  python - "$1" "$2" <<'PY' | tee -a gpurun_out/r04b/summary.txt
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-62s ms/step %.4f value %.3e launches %d kernel_us %.1f frac %.3f dev_ms %.4f" % (sys.argv[2] or "(default)", d["ms_per_step"], d["value"], d["config"]["kernel_launches_per_step"], d["roofline"]["avg_launch_us"], d["roofline"]["frac"], d["config"]["device_ms_per_step"]))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
i=0
for opts in "" "--opt strip_body_warm=0" "--opt self_contained=0" "--opt self_contained=0 --opt strip_body_warm=0" "--no-graph" "--no-graph --opt self_contained=0"; do
  i=$((i+1))
  timeout 300 python bench.py --no-extras --no-cpu $opts > $OUT/bench$i.json 2> $OUT/bench$i.err
  bench_line $OUT/bench$i.json "$opts"
done
for opts in "" "--opt strip_body_warm=0" "--opt self_contained=0" "--opt self_contained=0 --opt strip_body_warm=0"; do
  echo "== stamps: $opts" | tee -a $OUT/summary.txt
  S2AMD_DEBUG_TIMES=1 S2AMD_LIB=$PWD/solver2d_amd/libs2amd_stamps.so timeout 300 python bench.py --steps 50 --no-extras --no-cpu $opts 2>&1 | grep "us per phase" | tee -a $OUT/summary.txt
done
timeout 300 python -m pytest tests/test_dropin_product.py -q -m gpu > $OUT/dropin_product.log 2>&1; echo "dropin_product rc=$?" | tee -a $OUT/summary.txt
grep -n "AssertionError: (" $OUT/dropin_product.log | head -5 | tee -a $OUT/summary.txt
timeout 1500 python -m pytest tests -q -m gpu -x --deselect tests/test_dropin_product.py::test_bodies_created_and_replaced_between_resident_steps > $OUT/gpu_suite.log 2>&1; echo "gpu suite rc=$?" | tee -a $OUT/summary.txt
tail -6 $OUT/gpu_suite.log | tee -a $OUT/summary.txt
