#!/bin/bash
# A/B of builds of the persistent kernels on the headline (bench.py --no-extras): tools/kernel_ab.sh <out dir> [variant suffixes...]
# Prints ms/step, the dominant kernel's us per launch and device ms per step per variant ("" = the shipped library).
out=$1; shift
mkdir -p $out
for v in "" "$@"; do
  L=$PWD/solver2d_amd/libs2amd$v.so
  S2AMD_LIB=$L python bench.py --steps 100 --no-extras --no-cpu > $out/bench$v.json 2>/dev/null
  python3 -c "
import json,sys
d=json.loads(open('$out/bench$v.json').read().strip().split(chr(10))[-1])
print('variant[$v]', round(d['ms_per_step'],4), round(d['roofline']['avg_launch_us'],1), round(d['config']['device_ms_per_step'],4))
"
done
if [ -f solver2d_amd/libs2amd_stamps.so ]; then
  S2AMD_DEBUG_TIMES=1 S2AMD_LIB=$PWD/solver2d_amd/libs2amd_stamps.so python bench.py --steps 50 --no-extras --no-cpu 2>&1 | grep "us per phase" | tee $out/stamps.txt
fi
