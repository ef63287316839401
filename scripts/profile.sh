#!/bin/bash
# rocprofv3 evidence for the bench line (run on the GPU box through gpurun):
#   pass 1: --kernel-trace --stats   per-kernel durations of the SAME command bench.py times
#   pass 2/3 (PMC FETCH_SIZE / WRITE_SIZE) are run by bench.py itself (collect_traffic) in their own
#   processes -- counters are never combined with traces.
# Output: gpurun_out/prof_<tag>/ ; copy the *_kernel_stats.csv summary into profiles/.
set -u
TAG=${1:-r01}
shift || true
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o "$TAG" -- python "$REPO/bench.py" --no-traffic --no-cpu-baseline "$@" > "$OUT/bench_under_rocprof.json" 2> "$OUT/trace.log"
echo "rocprof exit $?"
find "$OUT" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$OUT/kernel_stats.csv"
cut -c1-220 "$OUT/kernel_stats.csv" | head -12
# keep the merge small: the raw per-dispatch trace is large
find "$OUT" -name "*kernel_trace.csv" -size +20M -delete
