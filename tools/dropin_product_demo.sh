#!/bin/sh
# The product drop-in at work: tools/dropin_product_demo.bin (built in the container that has the reference's headers:
# `tools/dropin_product_demo.sh build`) linked against shim/_build/libsolver2d_amd.so, run once per route.
#   tools/dropin_product_demo.sh build            # make -C shim + compile the demo (needs /root/reference)
#   tools/dropin_product_demo.sh [200 40 pyramid 7 8 4 45]
set -e
cd "$(dirname "$0")/.."
if [ "$1" = "build" ]; then
	make -s -C shim REF="${REF:-/root/reference}"
	gcc -O2 -I"${REF:-/root/reference}/include" tools/dropin_product_demo.c solver2d_amd/scenes/scenes.c -o tools/dropin_product_demo.bin \
		-Lshim/_build -lsolver2d_amd -Wl,-rpath,'$ORIGIN/../shim/_build' -lm
	exit 0
fi
if [ $# -eq 0 ]; then
	set -- 200 40 pyramid 7 8 4 45
fi
export S2AMD_LIBRARY="$PWD/solver2d_amd/libs2amd.so"
S2AMD_DROPIN=off tools/dropin_product_demo.bin "$@"
S2AMD_DROPIN=solver tools/dropin_product_demo.bin "$@"
S2AMD_DROPIN=step S2AMD_DEVICE_PAIRS=0 tools/dropin_product_demo.bin "$@"
S2AMD_DROPIN=step S2AMD_DEVICE_PAIRS=1 tools/dropin_product_demo.bin "$@"
