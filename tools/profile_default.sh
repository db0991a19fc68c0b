#!/bin/bash
# Round profile of the default bench command on the GPU box (run through gpurun from the repo root):
#   kernel-trace stats, then FETCH_SIZE and WRITE_SIZE in separate --pmc passes (MI355X_MICROARCH.md).
# Usage: bash tools/profile_default.sh <tag> [bench args...]   -> gpurun_out/prof_<tag>_{stats,fetch,write}/
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_${TAG}_stats $R/gpurun_out/prof_${TAG}_fetch $R/gpurun_out/prof_${TAG}_write
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_stats -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline "$@" > $R/gpurun_out/prof_${TAG}_stats.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof_${TAG}_fetch -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline "$@" > $R/gpurun_out/prof_${TAG}_fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof_${TAG}_write -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline "$@" > $R/gpurun_out/prof_${TAG}_write.log 2>&1
echo profiled $TAG
