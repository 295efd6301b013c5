#!/bin/bash
# GPU box: time the cell-kernel forms (TPG_CELLS_VARIANT=3 tile, 2 marching) under rocprofv3
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$REPO"
report() { python3 - "$1" <<'PY'
import csv, sys
for r in csv.DictReader(open(f"gpurun_out/prof_{sys.argv[1]}/bench_kernel_stats.csv")):
    if "k_cells" in r["Name"]: print(sys.argv[1], r["Name"][:48], r["AverageNs"], r["MinNs"])
PY
}
TPG_CELLS_VARIANT=3 tools/profile.sh tile --steps 12 --warmup 2 > /dev/null && report tile
TPG_CELLS_VARIANT=2 tools/profile.sh march --steps 12 --warmup 2 > /dev/null && report march
