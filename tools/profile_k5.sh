#!/bin/bash
# K5 fast-path evidence for profiles/: bench line, rocprofv3 kernel stats, in-kernel phase clocks of the fused kernel,
# config 5 (hybrid + rerank 200 -> 20, batch 64) with the bf16 cross-encoder.   bash tools/profile_k5.sh <tag>
set -o pipefail
tag=${1:-k5}
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out
timeout -k 10 200 python3 $R/tools/k5_bench.py --precision bf16 --reps 15 2>/dev/null | grep forward_ms > $out/${tag}_bench_bf16.json
timeout -k 10 200 python3 $R/tools/k5_bench.py --precision bf16 --reps 9 --len 0 2>/dev/null | grep forward_ms > $out/${tag}_bench_bf16_ragged.json
timeout -k 10 120 python3 $R/tools/k5_stamps.py 2>/dev/null | grep "workgroup\|forward" > $out/${tag}_stamps.txt
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_prof -o k5 -- python3 $R/tools/k5_bench.py --precision bf16 > $out/${tag}_prof.log 2>&1
cd $R
timeout -k 10 500 python3 bench.py --batch 64 --k 20 --rerank-k 200 --steps 5 --warmup 1 --no-cpu-baseline --ce-precision bf16 2>/dev/null | tail -1 > $out/${tag}_config5_bf16.json
cat $out/${tag}_bench_bf16.json $out/${tag}_stamps.txt; cut -c1-400 $out/${tag}_config5_bf16.json
