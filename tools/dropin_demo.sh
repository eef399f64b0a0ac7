#!/bin/sh
# Builds tools/dropin_demo.c against oracle/_ref/libs2ref.so (the unmodified reference + the shim of INTEGRATION.md,
# `make -C oracle ref` in the build container) and runs it on this box's GPU.
#   tools/dropin_demo.sh                                   # LargePyramid base 200, TGS_Soft 8/4, 40 timed steps
#   tools/dropin_demo.sh 10000 40 solver2d_amd/libs2amd.so tumbler 0 4 2 120
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
gcc -O2 tools/dropin_demo.c -o gpurun_out/dropin_demo -Loracle/_ref -ls2ref -Wl,-rpath,"$PWD/oracle/_ref" -lm
if [ $# -eq 0 ]; then
	set -- 200 40 solver2d_amd/libs2amd.so
fi
exec gpurun_out/dropin_demo "$@"
