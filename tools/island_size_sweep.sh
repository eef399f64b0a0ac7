#!/bin/sh
# One island of growing size under four solvers: ms per resident step with round 5's thresholds (an LDS group up to 2,048 bodies, strips
# from 4,096 loose bodies on) and round 6's (1,024 -- 896 for the soft solvers, in both rows: it is not an option -- / 768).
# (l launches, s strips, g LDS groups)   tools/island_size_sweep.sh
cd "$(dirname "$0")/.."
for b in 10 20 30 40 44 50 60 70 80 100 140 200; do
  for o in "max_group_bodies=2048 strip_min_bodies=4096" "max_group_bodies=1024 strip_min_bodies=768"; do
    set -- $o
    python tools/solver_table.py --base $b --solvers TGS_Soft,PGS_NGS_Block,SoftStep,XPBD --steps 200 --opt $1 --opt $2 2>/dev/null | python3 -c "
import sys,json
print('base %3d [%s]' % ($b, '$o'), '  '.join('%s %.3f (l%d s%d g%d)'%(d['solver'],d['ms_per_step'],d['launches'],d['strips'],d['groups']) for d in map(json.loads,sys.stdin)))"
  done
done
