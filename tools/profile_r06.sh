#!/bin/bash
# rocprofv3 passes behind profiles/r06_* (round 6; the r05 script with the drop-in's device-tree steps added): run on the GPU box from the repo root (gpurun).  Counter passes are separate runs
# (--kernel-trace --pmc X only), never combined with the sys / hip / hsa trace domains; each under `timeout`
# (S2AMD_PROFILE_PASS_SECONDS, 180).  Usage: tools/profile_r06.sh [headline|fast|config5|jacobi|generic|dropin|all]
# Summaries (tools/rocpd_summary.py) land in gpurun_out/prof6/out/ under the names they are committed as in profiles/.
R=$PWD
export TMPDIR=/tmp
cd /tmp
O=$R/gpurun_out/prof6
mkdir -p $O
what=${1:-all}
B="python $R/bench.py --steps 200 --warmup 40 --no-cpu --no-extras --no-fast"
C5="python $R/bench.py --config 5 --steps 60 --warmup 10"
T=${S2AMD_PROFILE_PASS_SECONDS:-180}
if [ $what = headline -o $what = all ]; then
  timeout $T rocprofv3 --kernel-trace --stats -d $O/h_stats -o trace -- $B > $O/h_stats.log 2>&1
  timeout $T rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/h_fetch -o pmc -- $B > $O/h_fetch.log 2>&1
  timeout $T rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/h_write -o pmc -- $B > $O/h_write.log 2>&1
  timeout $T rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES -d $O/h_sq_a -o pmc -- $B > $O/h_sq_a.log 2>&1
  timeout $T rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS -d $O/h_sq_b -o pmc -- $B > $O/h_sq_b.log 2>&1
fi
if [ $what = fast -o $what = all ]; then
  # the tolerance-mode build (libs2amd_fast.so: FMA contraction on) alone in the process: its wideStepKernel beside the bit-exact one above
  timeout $T rocprofv3 --kernel-trace --stats -d $O/fast -o trace -- python $R/tools/solver_table.py --fast --solvers TGS_Soft,PGS_Soft,SoftStep --steps 200 > $O/fast.log 2>&1
fi
if [ $what = config5 -o $what = all ]; then
  timeout $T rocprofv3 --kernel-trace --stats -d $O/c5_stats -o trace -- $C5 > $O/c5_stats.log 2>&1
  timeout $T rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/c5_fetch -o pmc -- $C5 > $O/c5_fetch.log 2>&1
  timeout $T rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/c5_write -o pmc -- $C5 > $O/c5_write.log 2>&1
  timeout $T rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES -d $O/c5_sq -o pmc -- $C5 > $O/c5_sq.log 2>&1
fi
if [ $what = jacobi -o $what = all ]; then
  # s2Solve_Jacobi as one persistent launch (jacobiStepKernel): BASELINE configs[2] (Tumbler 10k) and the base-200 pyramid
  timeout $T rocprofv3 --kernel-trace --stats -d $O/jac_t -o trace -- python $R/tools/config3_bench.py Jacobi > $O/jac_t.log 2>&1
  timeout $T rocprofv3 --kernel-trace --stats -d $O/jac_p -o trace -- python $R/tools/solver_table.py --solvers Jacobi --steps 200 > $O/jac_p.log 2>&1
fi
if [ $what = generic -o $what = all ]; then
  # the op interpreter (genericStepKernel): the reference's default solver and the joint grid
  timeout $T rocprofv3 --kernel-trace --stats -d $O/gen_b -o trace -- python $R/tools/solver_table.py --solvers PGS_NGS_Block,PGS,TGS_NGS,XPBD --steps 100 > $O/gen_b.log 2>&1
  timeout $T rocprofv3 --kernel-trace --stats -d $O/gen_j -o trace -- python $R/tools/solver_table.py --world joint_grid --base 100 --solvers PGS_NGS --steps 100 > $O/gen_j.log 2>&1
fi
if [ $what = dropin -o $what = all ]; then
  # the public s2World_Step through the product drop-in with everything on the device (pairs, trees): the reference's default solver
  # (every box moves every step: the whole dynamic tree is rebuilt) and the headline solver on the settled pyramid (460 boxes move)
  export S2AMD_DROPIN=step S2AMD_DEVICE_PAIRS=1 S2AMD_LIBRARY=$R/solver2d_amd/libs2amd.so
  timeout $T rocprofv3 --kernel-trace --stats -d $O/drop_b -o trace -- $R/tools/dropin_product_demo.bin 200 60 pyramid 3 4 2 45 > $O/drop_b.log 2>&1
  timeout $T rocprofv3 --kernel-trace --stats -d $O/drop_q -o trace -- $R/tools/dropin_product_demo.bin 200 60 pyramid 7 8 4 45 > $O/drop_q.log 2>&1
  unset S2AMD_DROPIN S2AMD_DEVICE_PAIRS S2AMD_LIBRARY
fi
cd $R
for d in drop_b drop_q h_stats h_fetch h_write h_sq_a h_sq_b fast c5_stats c5_fetch c5_write c5_sq jac_t jac_p gen_b gen_j; do
  db=$(find $O/$d -name "*_results.db" 2>/dev/null | head -1)
  [ -n "$db" ] && python tools/rocpd_summary.py $db > $O/$d.txt
done
P=$O/out
mkdir -p $P
[ -f $O/h_stats.txt ] && cp $O/h_stats.txt $P/r06_persistent_kernel_trace.txt
[ -f $O/h_fetch.txt ] && cp $O/h_fetch.txt $P/r06_persistent_pmc_fetch_size.txt
[ -f $O/h_write.txt ] && cp $O/h_write.txt $P/r06_persistent_pmc_write_size.txt
[ -f $O/h_sq_a.txt ] && cat $O/h_sq_a.txt $O/h_sq_b.txt > $P/r06_persistent_pmc_sq.txt
[ -f $O/fast.txt ] && (echo "# libs2amd_fast.so (-ffp-contract=fast): tools/solver_table.py --fast --solvers TGS_Soft,PGS_Soft,SoftStep --steps 200"; grep -h "^{" $O/fast.log; cat $O/fast.txt) > $P/r06_fast_build_kernel_trace.txt
[ -f $O/c5_stats.txt ] && cp $O/c5_stats.txt $P/r06_config5_kernel_trace.txt
[ -f $O/c5_fetch.txt ] && cp $O/c5_fetch.txt $P/r06_config5_pmc_fetch_size.txt
[ -f $O/c5_write.txt ] && cp $O/c5_write.txt $P/r06_config5_pmc_write_size.txt
[ -f $O/c5_sq.txt ] && cp $O/c5_sq.txt $P/r06_config5_pmc_sq.txt
[ -f $O/jac_t.txt ] && (echo "# tools/config3_bench.py Jacobi (Tumbler 10k)"; grep -h "^{" $O/jac_t.log; cat $O/jac_t.txt; echo; echo "# tools/solver_table.py --solvers Jacobi --steps 200 (pyramid base 200)"; grep -h "^{" $O/jac_p.log; cat $O/jac_p.txt) > $P/r06_jacobi_kernel_trace.txt
[ -f $O/gen_b.txt ] && (echo "# tools/solver_table.py --solvers PGS_NGS_Block,PGS,TGS_NGS,XPBD --steps 100 (pyramid base 200)"; grep -h "^{" $O/gen_b.log; cat $O/gen_b.txt; echo; echo "# tools/solver_table.py --world joint_grid --base 100 --solvers PGS_NGS --steps 100"; grep -h "^{" $O/gen_j.log; cat $O/gen_j.txt) > $P/r06_generic_kernel_trace.txt
[ -f $O/drop_b.txt ] && (echo "# tools/dropin_product_demo.bin 200 60 pyramid 3 4 2 45 (S2AMD_DROPIN=step, device pairs, device trees): the reference's default solver"; grep -h "^route\|per step" $O/drop_b.log; cat $O/drop_b.txt; echo; echo "# tools/dropin_product_demo.bin 200 60 pyramid 7 8 4 45: TGS_Soft on the settled pyramid"; grep -h "^route\|per step" $O/drop_q.log; cat $O/drop_q.txt) > $P/r06_dropin_trees_kernel_trace.txt
rm -rf $O/drop_b $O/drop_q $O/h_stats $O/h_fetch $O/h_write $O/h_sq_a $O/h_sq_b $O/fast $O/c5_stats $O/c5_fetch $O/c5_write $O/c5_sq $O/jac_t $O/jac_p $O/gen_b $O/gen_j
grep -h "wideStep\|IslandKernel\|jacobiStep\|genericStep" $P/*.txt | cut -c1-30,100-190 | head -40
