#!/bin/bash
# GPU box: time the tile form of K1 (TPG_CELLS_VARIANT=3) for each TPG_TILE_ROWS against the default
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$REPO"
report() { python3 - "$1" <<'PY'
import csv, sys
for r in csv.DictReader(open(f"gpurun_out/prof_{sys.argv[1]}/bench_kernel_stats.csv")):
    if "k_cells" in r["Name"]: print(sys.argv[1], r["Name"][:48], r["AverageNs"], r["MinNs"])
PY
}
for rows in 8; do
  TPG_CELLS_VARIANT=3 TPG_TILE_ROWS=$rows tools/profile.sh tile$rows --steps 12 --warmup 2 > /dev/null && report tile$rows
done
tools/profile.sh fast --steps 12 --warmup 2 > /dev/null && report fast
