#!/bin/bash
# rocprofv3 passes behind profiles/r04_*: run on the GPU box from the repo root (gpurun).  Counter passes are separate runs
# (--kernel-trace --pmc X only), never combined with the sys / hip / hsa trace domains; each under `timeout`
# (S2AMD_PROFILE_PASS_SECONDS, 180).  Usage: tools/profile_r04.sh [headline|config5|soft|calib|all]
R=$PWD
export TMPDIR=/tmp
cd /tmp
O=$R/gpurun_out/prof4
mkdir -p $O
what=${1:-all}
B="python $R/bench.py --steps 200 --warmup 40 --no-cpu --no-extras"
C5="python $R/bench.py --config 5 --steps 60 --warmup 10"
T=${S2AMD_PROFILE_PASS_SECONDS:-180}
if [ $what = headline -o $what = all ]; then
  timeout $T rocprofv3 --kernel-trace --stats -d $O/h_stats -o trace -- $B > $O/h_stats.log 2>&1
  timeout $T rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/h_fetch -o pmc -- $B > $O/h_fetch.log 2>&1
  timeout $T rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/h_write -o pmc -- $B > $O/h_write.log 2>&1
  timeout $T rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES -d $O/h_sq_a -o pmc -- $B > $O/h_sq_a.log 2>&1
  timeout $T rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS -d $O/h_sq_b -o pmc -- $B > $O/h_sq_b.log 2>&1
  # the one-launch form of the step (options, off by default): its kernel beside the three launches above
  timeout $T rocprofv3 --kernel-trace --stats -d $O/h_self -o trace -- $B --opt self_contained_strips=1 --opt strip_body_warm=1 > $O/h_self.log 2>&1
fi
if [ $what = config5 -o $what = all ]; then
  timeout $T rocprofv3 --kernel-trace --stats -d $O/c5_stats -o trace -- $C5 > $O/c5_stats.log 2>&1
  timeout $T rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/c5_fetch -o pmc -- $C5 > $O/c5_fetch.log 2>&1
  timeout $T rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/c5_write -o pmc -- $C5 > $O/c5_write.log 2>&1
  timeout $T rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES -d $O/c5_sq -o pmc -- $C5 > $O/c5_sq.log 2>&1
fi
if [ $what = soft -o $what = all ]; then
  # s2Solve_PGS_Soft and s2Solve_SoftStep on the 512-thread kernel (wideStepKernel<..., SOFT_PGS / SOFT_FIXED>) beside TGS_Soft
  timeout $T rocprofv3 --kernel-trace --stats -d $O/soft -o trace -- python $R/tools/solver_table.py --solvers PGS_Soft,SoftStep,TGS_Soft --steps 200 > $O/soft.log 2>&1
fi
if [ $what = calib -o $what = all ]; then
  # FETCH_SIZE against bytes really requested: a coalesced float4 stream, one 152-byte record per lane, one 88-byte record per lane
  timeout $T rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/calib -o pmc -- $R/tools/fetch_calib.bin > $O/calib.log 2>&1
fi
cd $R
for d in h_stats h_fetch h_write h_sq_a h_sq_b h_self c5_stats c5_fetch c5_write c5_sq calib soft; do
  db=$(find $O/$d -name "*_results.db" 2>/dev/null | head -1)
  [ -n "$db" ] && python tools/rocpd_summary.py $db > $O/$d.txt
done
P=$R/gpurun_out/prof4/out
mkdir -p $P
[ -f $O/h_stats.txt ] && cp $O/h_stats.txt $P/r04_persistent_kernel_trace.txt
[ -f $O/h_fetch.txt ] && cp $O/h_fetch.txt $P/r04_persistent_pmc_fetch_size.txt
[ -f $O/h_write.txt ] && cp $O/h_write.txt $P/r04_persistent_pmc_write_size.txt
[ -f $O/h_sq_a.txt ] && cat $O/h_sq_a.txt $O/h_sq_b.txt > $P/r04_persistent_pmc_sq.txt
[ -f $O/h_self.txt ] && cp $O/h_self.txt $P/r04_selfcontained_kernel_trace.txt
[ -f $O/c5_stats.txt ] && cp $O/c5_stats.txt $P/r04_config5_kernel_trace.txt
[ -f $O/c5_fetch.txt ] && cp $O/c5_fetch.txt $P/r04_config5_pmc_fetch_size.txt
[ -f $O/c5_write.txt ] && cp $O/c5_write.txt $P/r04_config5_pmc_write_size.txt
[ -f $O/c5_sq.txt ] && cp $O/c5_sq.txt $P/r04_config5_pmc_sq.txt
[ -f $O/soft.txt ] && cp $O/soft.txt $P/r04_soft_solvers_kernel_trace.txt
[ -f $O/calib.txt ] && (cat $O/calib.log | grep "bytes" ; cat $O/calib.txt) > $P/r04_fetch_size_calibration.txt
rm -rf $O/h_stats $O/h_fetch $O/h_write $O/h_sq_a $O/h_sq_b $O/h_self $O/c5_stats $O/c5_fetch $O/c5_write $O/c5_sq $O/calib $O/soft
grep -h "wideStep\|IslandKernel\|islandStep\|streamFloat4\|recordPerLane" $P/*.txt | cut -c1-30,100-190 | head -40
