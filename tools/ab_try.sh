#!/bin/bash
# GPU box: A/B two builds of the grid TU that differ by a -D flag (A = no flag, B = $1), alternating
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$REPO"
for rep in 1 2 3 4; do
  for flags in "" "$1"; do
    touch orthogonalsphericalshellgrids.jl_amd/csrc/tpg_grid.hip
    make -C orthogonalsphericalshellgrids.jl_amd/csrc GRID_FLAGS="$flags" > /dev/null 2>&1 || { echo "build failed: $flags"; exit 1; }
    tools/profile.sh ab --no-aux --no-fill-step --steps 60 --warmup 20 > /dev/null || exit 1
    python3 - "$flags" <<'PY'
import csv, sys
for r in csv.DictReader(open("gpurun_out/prof_ab/bench_kernel_stats.csv")):
    if "k_cells" in r["Name"]: print(f"[{sys.argv[1]}]", r["AverageNs"], r["MinNs"])
PY
  done
done
