#!/bin/sh
# The pile that comes down: the base-200 pyramid under the reference's default solver (PGS_NGS_Block, 4 / 2 iterations cannot hold it),
# through the product drop-in, 100 timed steps after 500 -- the regime in which contacts are created and destroyed every step all over
# one big graph.  Prints the demo's two lines, the number of structure builds and what forced them.   tools/collapse_demo.sh [options]
cd "$(dirname "$0")/.."
export S2AMD_LIBRARY=$PWD/solver2d_amd/libs2amd.so S2AMD_DROPIN=step S2AMD_DEVICE_PAIRS=1
S2AMD_OPTIONS=$1 S2AMD_DEBUG_PREP=1 tools/dropin_product_demo.bin 200 100 pyramid 3 4 2 500 2> /tmp/collapse.err | tail -2
echo "structure builds in the 600 steps: $(grep -c 'rebuild #' /tmp/collapse.err)"
grep -o "reason: .*" /tmp/collapse.err | sort | uniq -c | sort -rn | head -6
