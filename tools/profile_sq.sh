#!/bin/bash
# SQ counter passes of the headline command (separate runs, --kernel-trace --pmc only): instruction mix and issue activity of
# stripStepKernel, and of the config-5 island kernel
R=$PWD
export TMPDIR=/tmp
cd /tmp
O=$R/gpurun_out
B="python $R/bench.py --steps 100 --warmup 20 --no-cpu --no-extras"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES -d $O/p2_sq_a -o pmc -- $B > $O/p2_sq_a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS -d $O/p2_sq_b -o pmc -- $B > $O/p2_sq_b.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES -d $O/p2_sq_c5 -o pmc -- python $R/bench.py --config 5 --steps 30 --warmup 5 --no-cpu > $O/p2_sq_c5.log 2>&1
cd $R
for d in p2_sq_a p2_sq_b p2_sq_c5; do
  db=$(find $O/$d -name "*_results.db" | head -1)
  [ -n "$db" ] && python tools/rocpd_summary.py $db > $O/$d.txt
done
grep -h "stripStep\|islandStep" $O/p2_sq_a.txt $O/p2_sq_b.txt $O/p2_sq_c5.txt | grep "SQ_" | cut -c1-30,100-175
