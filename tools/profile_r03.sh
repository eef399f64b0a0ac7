#!/bin/bash
# rocprofv3 passes behind profiles/r03_*: run on the GPU box from the repo root (gpurun).  Counter passes are separate runs
# (--kernel-trace --pmc X only), never combined with the sys / hip / hsa trace domains; each under `timeout` (S2AMD_PROFILE_PASS_SECONDS, 180).  Usage: tools/profile_r03.sh [headline|config5|generic|all]
R=$PWD
export TMPDIR=/tmp
cd /tmp
O=$R/gpurun_out/prof3
mkdir -p $O
what=${1:-all}
B="python $R/bench.py --steps 200 --warmup 40 --no-cpu --no-extras"
C5="python $R/bench.py --config 5 --steps 60 --warmup 10"
# every pass under its own limit: a pass that does not come back must not take the box's budget with it
T=${S2AMD_PROFILE_PASS_SECONDS:-180}
if [ $what = headline -o $what = all ]; then
  timeout $T rocprofv3 --kernel-trace --stats -d $O/h_stats -o trace -- $B > $O/h_stats.log 2>&1
  timeout $T rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/h_fetch -o pmc -- $B > $O/h_fetch.log 2>&1
  timeout $T rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/h_write -o pmc -- $B > $O/h_write.log 2>&1
  timeout $T rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES -d $O/h_sq_a -o pmc -- $B > $O/h_sq_a.log 2>&1
  timeout $T rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS -d $O/h_sq_b -o pmc -- $B > $O/h_sq_b.log 2>&1
fi
if [ $what = config5 -o $what = all ]; then
  timeout $T rocprofv3 --kernel-trace --stats -d $O/c5_stats -o trace -- $C5 > $O/c5_stats.log 2>&1
  timeout $T rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/c5_fetch -o pmc -- $C5 > $O/c5_fetch.log 2>&1
  timeout $T rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/c5_write -o pmc -- $C5 > $O/c5_write.log 2>&1
  timeout $T rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES -d $O/c5_sq -o pmc -- $C5 > $O/c5_sq.log 2>&1
fi
if [ $what = generic -o $what = all ]; then
  # the op interpreter (generic_kernel.hip) under the reference's default solver at base 200 and on JointGrid 100x100
  timeout $T rocprofv3 --kernel-trace --stats -d $O/g_block -o trace -- python $R/tools/solver_table.py --solvers PGS_NGS_Block --steps 100 > $O/g_block.log 2>&1
  timeout $T rocprofv3 --kernel-trace --stats -d $O/g_joint -o trace -- python $R/tools/solver_table.py --world joint_grid --base 100 --solvers PGS_NGS --steps 100 > $O/g_joint.log 2>&1
fi
cd $R
for d in g_block g_joint; do
  db=$(find $O/$d -name "*_results.db" 2>/dev/null | head -1)
  [ -n "$db" ] && python tools/rocpd_summary.py $db > $O/$d.txt
done
for d in h_stats h_fetch h_write h_sq_a h_sq_b c5_stats c5_fetch c5_write c5_sq; do
  db=$(find $O/$d -name "*_results.db" 2>/dev/null | head -1)
  [ -n "$db" ] && python tools/rocpd_summary.py $db > $O/$d.txt
done
# the summaries under the names profiles/ keeps
P=$R/gpurun_out/prof3/out
mkdir -p $P
[ -f $O/g_block.txt ] && (echo "## pyramid base 200, s2_solverPGS_NGS_Block 4/2 (tools/solver_table.py --solvers PGS_NGS_Block --steps 100)"; cat $O/g_block.txt; echo; echo "## JointGrid 100x100, s2_solverPGS_NGS 4/2 (tools/solver_table.py --world joint_grid --base 100 --solvers PGS_NGS --steps 100)"; cat $O/g_joint.txt) > $P/r03_generic_kernel_trace.txt
[ -f $O/h_stats.txt ] && cp $O/h_stats.txt $P/r03_persistent_kernel_trace.txt
[ -f $O/h_fetch.txt ] && cp $O/h_fetch.txt $P/r03_persistent_pmc_fetch_size.txt
[ -f $O/h_write.txt ] && cp $O/h_write.txt $P/r03_persistent_pmc_write_size.txt
[ -f $O/h_sq_a.txt ] && cat $O/h_sq_a.txt $O/h_sq_b.txt > $P/r03_persistent_pmc_sq.txt
[ -f $O/c5_stats.txt ] && cp $O/c5_stats.txt $P/r03_config5_kernel_trace.txt
[ -f $O/c5_fetch.txt ] && cp $O/c5_fetch.txt $P/r03_config5_pmc_fetch_size.txt
[ -f $O/c5_write.txt ] && cp $O/c5_write.txt $P/r03_config5_pmc_write_size.txt
[ -f $O/c5_sq.txt ] && cp $O/c5_sq.txt $P/r03_config5_pmc_sq.txt
# keep the merge small: the raw databases stay on the box
rm -rf $O/h_stats $O/h_fetch $O/h_write $O/h_sq_a $O/h_sq_b $O/c5_stats $O/c5_fetch $O/c5_write $O/c5_sq
grep -h "wideStep\|islandStep" $P/*.txt | cut -c1-24,100-180 | head -40
