#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel-trace stats of bench.py (timed separately from PMC passes).
# usage: tools/profile.sh <tag> [bench args...]
set -o pipefail
TAG=${1:-r01}; shift
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o bench -- python3 "$REPO/bench.py" --no-cpu-baseline "$@" > "$OUT/bench_under_rocprof.json" 2> "$OUT/rocprof.stderr"
echo "rocprofv3 rc=$?"
find "$OUT" -name "*stats*" | head
