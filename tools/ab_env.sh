#!/bin/bash
# Same-box A/B of environment switches on the default bench (GPU box):
#   bash tools/ab_env.sh "X=1" "RR_NO_DUAL=1" "RR_NO_PREFILTER=1" [-- bench args]
# ("X=1" = a variable nothing reads: the default path.)  One line per variant: q/s, ms per step, scan launch ms, fraction.
VARS=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do VARS+=("$1"); shift; done
[ "$1" == "--" ] && shift
for V in "${VARS[@]}"; do
  env $V python bench.py --no-cpu-baseline --steps 30 "$@" > gpurun_out/ab_tmp.json 2>gpurun_out/ab_tmp.err || { echo "$V failed"; tail -3 gpurun_out/ab_tmp.err; continue; }
  python - "$V" <<PY
import json, sys
d = json.load(open("gpurun_out/ab_tmp.json")); r = d["roofline"]
print(sys.argv[1], d["value"], "q/s", d["ms_per_step"], "ms/step; scan launch", r["avg_launch_ms"], "ms,", r["frac"], "of", r["unit"], "peak;",
      "one set:", d.get("roofline_one_query_set", {}).get("avg_launch_ms"))
PY
done
