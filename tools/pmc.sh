#!/bin/bash
# GPU box: PMC passes over bench.py (each counter group in its own run, kernel-trace only).
# usage: tools/pmc.sh <tag> [bench args]
TAG=${1:-pmc}; shift
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
run() {  # name counters...
  local name=$1; shift
  local OUT=$REPO/gpurun_out/pmc_${TAG}_$name; rm -rf "$OUT"; mkdir -p "$OUT"
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT" -o p -- python3 "$REPO/bench.py" --no-cpu-baseline --no-aux --no-fill-step --no-cold-onset --preroll 4 --steps 5 --warmup 1 > "$OUT/bench.json" 2> "$OUT/stderr.txt"
  echo "== $name rc=$?"
}
run sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE
run fetch FETCH_SIZE
run write WRITE_SIZE
python3 - "$REPO/gpurun_out" "$TAG" <<'PY'
import csv, sys, glob, collections
root, tag = sys.argv[1], sys.argv[2]
for name in ("sq", "fetch", "write"):
    files = glob.glob(f"{root}/pmc_{tag}_{name}/*counter_collection.csv")
    if not files: print(name, "no counter file"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(files[0])):
        k = r["Kernel_Name"].split("(")[0].split("::")[-1][:28]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        print(name, f"{k:30s}", {c: round(sum(v) / len(v), 1) for c, v in cs.items()}, "n=", len(next(iter(cs.values()))))
PY
