#!/bin/bash
# Counter passes of the default bench command on the GPU box (one counter group per pass; never with trace domains
# beyond --kernel-trace).  Usage: bash tools/profile_pmc.sh <tag> "<counters of pass 1>" ["<counters of pass 2>" ...]
#   -> gpurun_out/pmc_<tag>_<i>/ ; extra bench args through RR_PROFILE_BENCH_ARGS.  A pass whose counters the
#   device does not have is reported and skipped.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
i=0
for C in "$@"; do
  i=$((i+1))
  rm -rf $R/gpurun_out/pmc_${TAG}_$i
  if ! timeout -k 10 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${TAG}_$i -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline $RR_PROFILE_BENCH_ARGS > $R/gpurun_out/pmc_${TAG}_$i.log 2>&1; then
    echo "pass $i ($C) failed:"; tail -3 $R/gpurun_out/pmc_${TAG}_$i.log; continue
  fi
  python3 - "$R/gpurun_out/pmc_${TAG}_$i" <<'PY'
import csv, glob, sys, collections
fs = glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)
if not fs:
    print('no counter file'); sys.exit(0)
acc = collections.defaultdict(list)
for r in csv.DictReader(open(max(fs, key=len))):
    k = r['Kernel_Name']
    if 'rr_scan_flt' in k:
        acc[(k[:44], r['Counter_Name'])].append(float(r['Counter_Value']))
for (k, c), v in sorted(acc.items()):
    print(f'{k:44s} {c:28s} n={len(v):3d} mean={sum(v)/len(v):.6g}')
PY
done
