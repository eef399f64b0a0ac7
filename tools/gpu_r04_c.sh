#!/bin/bash
# round 4, GPU call C: the self-contained step after the prologue changes; where its extra time goes (commit wait on / off, another stamped workgroup); the whole GPU suite
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r04c
mkdir -p $OUT
rm -f $OUT/summary.txt
bench_line() {
  python - "$1" "$2" <<'PY' | tee -a gpurun_out/r04c/summary.txt
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-72s ms/step %.4f value %.3e launches %d kernel_us %.1f frac %.3f dev_ms %.4f graph %s" % (sys.argv[2] or "(default)", d["ms_per_step"], d["value"], d["config"]["kernel_launches_per_step"], d["roofline"]["avg_launch_us"], d["roofline"]["frac"], d["config"]["device_ms_per_step"], d["config"]["graph_replay"]))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
i=0
for opts in "" "--opt self_contained=0" "--opt strip_body_warm=0" "--opt self_contained=0 --opt strip_body_warm=0" "--opt graph_min_launches=0" "--opt graph_min_launches=0 --opt self_contained=0"; do
  i=$((i+1))
  timeout 300 python bench.py --no-extras --no-cpu $opts > $OUT/bench$i.json 2> $OUT/bench$i.err
  bench_line $OUT/bench$i.json "$opts"
done
export S2AMD_LIB=$PWD/solver2d_amd/libs2amd_stamps.so
for opts in "" "--opt persist_debug=32" "--opt persist_debug=256" "--opt persist_debug=512 --opt self_contained=0" "--opt self_contained=0"; do
  i=$((i+1))
  echo "== instrumented build: $opts" | tee -a $OUT/summary.txt
  S2AMD_DEBUG_TIMES=1 timeout 300 python bench.py --steps 50 --no-extras --no-cpu $opts 2> $OUT/stamps$i.err > $OUT/bench$i.json
  grep "us per phase" $OUT/stamps$i.err | tee -a $OUT/summary.txt
  bench_line $OUT/bench$i.json "instrumented $opts"
done
unset S2AMD_LIB
timeout 2400 python -m pytest tests -q -m gpu > $OUT/gpu_suite.log 2>&1; echo "gpu suite rc=$?" | tee -a $OUT/summary.txt
tail -8 $OUT/gpu_suite.log | tee -a $OUT/summary.txt
