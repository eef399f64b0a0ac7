#!/bin/sh
# Every sample scene at a size worth a GPU through the product drop-in's whole-step route, under the reference's default solver and
# under TGS_Soft: ms per public s2World_Step (and the host parts) after a short and after a long run-in, beside the reference alone.
# A survey for cliffs (a regime that builds its structure every step, a scene that falls off the persistent kernels), not a benchmark.
#   tools/scene_survey.sh [steps] [settle]
cd "$(dirname "$0")/.."
export S2AMD_LIBRARY=$PWD/solver2d_amd/libs2amd.so S2AMD_DEVICE_PAIRS=1
steps=${1:-60}; settle=${2:-300}
for spec in "pyramid 100" "multi_pyramid 40" "joint_grid 100" "tumbler 4000" "mixed 60" "circle_pile 60" "shapes_zoo 40" "far_ragdoll_pile 20" "rush 60" "confined 40" "circle_stack 60" "ragdoll_stress 30" "double_domino 60" "bridge 160" "card_house 6"; do
  set -- $spec
  for solver in "3 4 2" "7 8 4"; do
    ref=$(S2AMD_DROPIN=off tools/dropin_product_demo.bin $2 $steps $1 $solver $settle 2>&1 | head -1 | sed 's/.*: \([0-9.]*\) ms per.*/\1/')
    out=$(S2AMD_DROPIN=step tools/dropin_product_demo.bin $2 $steps $1 $solver $settle 2>&1)
    dev=$(echo "$out" | head -1 | sed 's/.*: \([0-9.]*\) ms per.*/\1/')
    parts=$(echo "$out" | sed -n 2p | sed 's/ *per step: //')
    echo "$1 $2 solver [$solver]: reference $ref ms, device $dev ms | $parts"
  done
done
