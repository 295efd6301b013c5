#!/bin/bash
# GPU box: the measurements a round commits under profiles/<tag>/ -- the bench line at the driver's arguments and at the defaults,
# rocprofv3 kernel stats of the default command (+ the step-only summary), and the PMC passes.  usage: tools/measure.sh <tag>
set -o pipefail
TAG=${1:-r03}
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$REPO/gpurun_out
mkdir -p "$O"
cd "$REPO"
python3 bench.py --steps 20 --warmup 5 > "$O/${TAG}_bench_driver_args.json" 2> "$O/${TAG}_bench_driver_args.err" && echo "bench (driver args) ok" || { echo "bench failed"; tail -5 "$O/${TAG}_bench_driver_args.err"; exit 1; }
python3 bench.py > "$O/${TAG}_bench.json" 2> "$O/${TAG}_bench.err" && echo "bench (defaults) ok" || { echo "bench failed"; tail -5 "$O/${TAG}_bench.err"; exit 1; }
bash tools/profile.sh "$TAG" || exit 1
TRACE=$(find "$O/prof_$TAG" -name "*kernel_trace.csv" | head -1)
STATS=$(find "$O/prof_$TAG" -name "*kernel_stats.csv" | head -1)
python3 tools/trace_summary.py "$TRACE" "$O/${TAG}_step_kernel_stats.csv" && cat "$O/${TAG}_step_kernel_stats.csv"
cp "$STATS" "$O/${TAG}_kernel_stats.csv"
rm -f "$TRACE"                                   # tens of MB; the two summaries are what is kept
bash tools/pmc.sh "$TAG" > "$O/${TAG}_pmc_summary.txt" 2>&1; tail -40 "$O/${TAG}_pmc_summary.txt"
