#!/bin/bash
# usage: [PMC_KERNELS=k_a,k_b] tools/pmc2.sh <tag> <counters...>   (env passes through to bench.py)
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/pmc2_$TAG; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT" -o p -- python3 "$REPO/bench.py" --no-cpu-baseline --steps 4 --warmup 1 > "$OUT/bench.json" 2> "$OUT/stderr.txt"
python3 - "$OUT" <<'PY'
import csv, sys, glob, collections, re
out = sys.argv[1]
f = glob.glob(out + "/*counter_collection.csv")
if not f: print("no counters; stderr tail:"); print(open(out + "/stderr.txt").read()[-1500:]); sys.exit()
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    m = re.search(r"(k_[a-z_]+)", r["Kernel_Name"]); k = m.group(1) if m else "?"
    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
import os
want = os.environ.get("PMC_KERNELS", "k_cells_march,k_cells_fast,k_cells,k_cells_tile").split(",")
for k in want:
    if k in acc: print(k, {c: round(sum(v[1:]) / max(1, len(v[1:]))) for c, v in acc[k].items()})
PY
