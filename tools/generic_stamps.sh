#!/bin/bash
# Where a launch of genericStepKernel spends its time: a build of generic_kernel.hip with -DS2_GENERIC_INSTRUMENTED=1 (one workgroup in the
# middle writes wall_clock64 at kernel start, after the loads, and per constraint op after the interiors / the forward hand-off / the seam /
# the return hand-off; body ops one stamp each; one after the store) beside the shipped objects, run on the base-200 pyramid and the
# 100 x 100 joint grid.  tools/generic_stamps.sh <out file>
set -e
cd "$(dirname "$0")/.."
C=solver2d_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fvisibility=hidden -Iinclude -I$C -Wall -Wno-unused-function"
if [ ! -f solver2d_amd/libs2amd_gstamps.so ] || [ $C/generic_kernel.hip -nt solver2d_amd/libs2amd_gstamps.so ]; then
  mkdir -p $C/build_var
  /opt/rocm/bin/hipcc $FLAGS -DS2_GENERIC_INSTRUMENTED=1 -x hip -c $C/generic_kernel.hip -o $C/build_var/generic_kernel.gstamps.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o solver2d_amd/libs2amd_gstamps.so $(ls $C/build/*.o | grep -v generic_kernel) $C/build_var/generic_kernel.gstamps.o
fi
out=${1:-/dev/stdout}
{
  for s in PGS_NGS_Block PGS PGS_NGS TGS_NGS XPBD; do
    S2AMD_DEBUG_TIMES=1 S2AMD_LIB=$PWD/solver2d_amd/libs2amd_gstamps.so python tools/solver_table.py --solvers $s --steps 20 2>&1 | grep "^\[s2amd\] persistent step\|^{"
  done
  for s in PGS_NGS TGS_Soft; do
    S2AMD_DEBUG_TIMES=1 S2AMD_LIB=$PWD/solver2d_amd/libs2amd_gstamps.so python tools/solver_table.py --world joint_grid --base 100 --solvers $s --steps 20 2>&1 | grep "^\[s2amd\] persistent step\|^{"
  done
} > $out
