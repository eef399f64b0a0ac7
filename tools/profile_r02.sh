#!/bin/bash
# rocprofv3 passes behind profiles/r02_*: run on the GPU box from the repo root (gpurun).  Counter passes are separate runs
# (--kernel-trace --pmc X only), never combined with the sys / hip / hsa trace domains.
R=$PWD
export TMPDIR=/tmp
cd /tmp
O=$R/gpurun_out
B="python $R/bench.py --steps 200 --warmup 40 --no-cpu --no-extras"
rocprofv3 --kernel-trace --stats -d $O/p2_stats -o trace -- $B > $O/p2_stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/p2_fetch -o pmc -- $B > $O/p2_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/p2_write -o pmc -- $B > $O/p2_write.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/p2_c5 -o trace -- python $R/bench.py --config 5 --steps 60 --warmup 10 > $O/p2_c5.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/p2_c5_fetch -o pmc -- python $R/bench.py --config 5 --steps 60 --warmup 10 > $O/p2_c5_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/p2_c5_write -o pmc -- python $R/bench.py --config 5 --steps 60 --warmup 10 > $O/p2_c5_write.log 2>&1
# (rocprofv3 segfaults on tools/churn_bench.py on this image: no churn trace)
cd $R
for d in p2_stats p2_fetch p2_write p2_c5 p2_c5_fetch p2_c5_write; do
  db=$(find $O/$d -name "*_results.db" | head -1)
  [ -n "$db" ] && python tools/rocpd_summary.py $db > $O/$d.txt
done
ls -la $O/*.txt
