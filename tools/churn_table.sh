#!/bin/sh
# The churn table of profiles/: every solver on the wrecked base-200 pyramid and the Tumbler filled from scratch, 200 steps of the whole
# loop each (tools/churn_bench.py), one JSON object per line with what forced the structure builds.   tools/churn_table.sh <out.jsonl>
cd "$(dirname "$0")/.."
out=${1:-/dev/stdout}
tmp=$(mktemp -d)
{
echo "# tools/churn_table.sh: tools/churn_bench.py --world W --solver X --steps 200 (pair query, contact creation, s2amd_world_step, destruction per step), S2AMD_DEBUG_PREP=1 for the reasons"
for spec in "wreck TGS_Soft" "wreck SoftStep" "wreck PGS_Soft" "wreck PGS_NGS_Block" "wreck PGS_NGS" "wreck PGS" "wreck TGS_NGS" "wreck TGS_Sticky" "wreck XPBD" "wreck Jacobi" "tumbler TGS_Soft" "tumbler Jacobi"; do
  set -- $spec
  S2AMD_DEBUG_PREP=1 timeout 300 python tools/churn_bench.py --world $1 --solver $2 --steps 200 > $tmp/o.json 2> $tmp/o.err
  python3 - $tmp/o.json $tmp/o.err "$1" "$2" <<'PY'
import json, sys, collections, re
try:
    d = json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
except Exception as e:
    print(json.dumps({"world": sys.argv[3], "solver": sys.argv[4], "failed": str(e)[:80]}))
    sys.exit(0)
reasons = collections.Counter(m.group(1).strip() or "(upload / first builds)" for m in re.finditer(r"reason: (.*)", open(sys.argv[2]).read()))
a = d["all_steps"]
print(json.dumps({"world": sys.argv[3], "solver": sys.argv[4], "steps": d["steps"], "step_ms": round(a["step_ms"], 3), "world_step_ms": round(a["world_step_ms"], 3),
                  "host_structure_ms": round(a["host_structure_ms"], 3), "solve_device_ms": round(a["solve_device_ms"], 3), "launches": round(a["launches"], 1),
                  "steps_that_rebuilt_the_structure": d["steps_that_rebuilt_the_structure"], "steps_on_persistent_kernel": d["steps_on_persistent_kernel"],
                  "contacts_created": d["contacts_created"], "contacts_placed_without_rebuild": d["contacts_placed_without_rebuild"],
                  "slowest_step_ms": d["slowest_steps_ms"][0] if d["slowest_steps_ms"] else None, "builds_by_reason": dict(reasons)}))
PY
done
} > $out
rm -rf $tmp
