#!/bin/bash
# rocprofv3 kernel trace of the whole world loop with the stage-1 pair query after every step (bench.py: whole_step_leg)
R=$PWD
export TMPDIR=/tmp
cd /tmp
O=$R/gpurun_out
rocprofv3 --kernel-trace --stats -d $O/p2_wq -o trace -- python -c "
import sys; sys.path.insert(0, '$R')
import bench
bench.whole_step_leg(0, 200, 8, 4, 30, 120)" > $O/p2_wq.log 2>&1
cd $R
db=$(find $O/p2_wq -name "*_results.db" | head -1)
[ -n "$db" ] && python tools/rocpd_summary.py $db > $O/p2_wq.txt
cat $O/p2_wq.txt | cut -c1-100,101-170 | head -40
