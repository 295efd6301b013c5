#!/bin/bash
# GPU box: exact VALU / SALU instruction counts of ONE 3600x1800 Float64 k_cells_tile launch (deterministic: the figure to compare
# between builds of the cell kernel) + its duration in the same run.  usage: tools/pmc_cells.sh <tag>
TAG=${1:-cells}
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/pmc_cells_$TAG; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT" -o p -- python3 "$REPO/bench.py" --no-cpu-baseline --no-aux --no-fill-step --steps 40 --warmup 40 > "$OUT/bench.json" 2> "$OUT/stderr.txt"
python3 - "$OUT" <<'PY'
import csv, sys, glob, collections
out = sys.argv[1]
rows = collections.defaultdict(dict)
for r in csv.DictReader(open(glob.glob(out + "/*counter_collection.csv")[0])):
    if "k_cells_tile" in r["Kernel_Name"]:
        d = rows[r["Dispatch_Id"]]
        d[r["Counter_Name"]] = float(r["Counter_Value"]); d["dur"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
sel = [d for d in rows.values() if d.get("SQ_WAVES") == 121776.0]
n = len(sel)
avg = lambda k: sum(d[k] for d in sel) / n
tail = sel[n // 2:]
print(f"k_cells_tile 3600x1800 f64: {n} launches; SQ_INSTS_VALU {avg('SQ_INSTS_VALU'):.0f}  SQ_INSTS_SALU {avg('SQ_INSTS_SALU'):.0f}  "
      f"VALU-active/busy {avg('SQ_ACTIVE_INST_VALU') / avg('SQ_BUSY_CYCLES'):.3f}  duration under counters: all {avg('dur'):.1f} us, later half {sum(d['dur'] for d in tail) / len(tail):.1f} us")
PY
