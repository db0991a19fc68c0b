# A/B of the filter scan's RR_FLT_TUNE switches on the default bench (GPU box): bash tools/ab_tune.sh "<tunes>" [bench args]
TUNES=${1:-"0 1"}; shift
for T in $TUNES; do
  RR_FLT_TUNE=$T python bench.py --no-cpu-baseline --steps 30 "$@" > gpurun_out/bench_tune$T.json 2>gpurun_out/bench_tune$T.err || { echo "tune $T failed"; tail -3 gpurun_out/bench_tune$T.err; }
done
python - $TUNES <<PY
import json, sys
for T in sys.argv[1:]:
    try:
        d=json.load(open(f"gpurun_out/bench_tune{T}.json")); r=d["roofline"]
        print("tune",T, d["value"], d["ms_per_step"], "scan", r["avg_launch_ms"], r["frac"], r.get("matrix_core_frac_of_2.5PF"))
    except Exception as e: print("tune",T,"error",e)
PY
