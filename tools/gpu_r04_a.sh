#!/bin/bash
# round 4, GPU call A: the one-launch headline step -- parity first, then the A/B of its two features on the headline bench
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r04a
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_selfstrips.py -x -q > $OUT/selfstrips.log 2>&1; echo "selfstrips rc=$?" | tee -a $OUT/summary.txt
tail -5 $OUT/selfstrips.log
for opts in "" "--opt strip_body_warm=0" "--opt self_contained=0" "--opt self_contained=0 --opt strip_body_warm=0" "--restore" "--restore --opt self_contained=0 --opt strip_body_warm=0"; do
  name=$(echo "bench$opts" | tr -d ' =-' | tr -c 'a-zA-Z0-9_\n' '_')
  timeout 300 python bench.py --no-extras --no-cpu $opts > $OUT/$name.json 2> $OUT/$name.err
  python - "$OUT/$name.json" "$opts" <<'PY' | tee -a $OUT/summary.txt
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-60s ms/step %.4f value %.3e launches %d kernel_us %.1f frac %.3f" % (sys.argv[2] or "(default)", d["ms_per_step"], d["value"], d["config"]["kernel_launches_per_step"], d["roofline"]["avg_launch_us"], d["roofline"]["frac"]))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
done
timeout 1500 python -m pytest tests/test_gpu_strips.py tests/test_gpu_world.py tests/test_gpu_dropin.py tests/test_dropin_product.py tests/test_gpu_incremental.py -x -q -m gpu > $OUT/regress.log 2>&1; echo "regress rc=$?" | tee -a $OUT/summary.txt
tail -5 $OUT/regress.log
cat $OUT/summary.txt
