#!/bin/bash
# GPU box: VALU instruction mix of the 3600x1800 Float64 k_cells_tile launch (two counter passes).  usage: tools/pmc_cells_mix.sh
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for pass in "a SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_WAVES" "b SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_IOPS SQ_INSTS_LDS SQ_WAVES SQ_INSTS_BRANCH SQ_INSTS_SMEM"; do
  set -- $pass; tag=$1; shift
  OUT=$REPO/gpurun_out/pmc_mix_$tag; rm -rf "$OUT"; mkdir -p "$OUT"
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT" -o p -- python3 "$REPO/bench.py" --no-cpu-baseline --no-aux --no-fill-step --steps 4 --warmup 2 > "$OUT/bench.json" 2> "$OUT/stderr.txt"
  python3 - "$OUT" <<'PY'
import csv, sys, glob, collections
rows = collections.defaultdict(dict)
f = glob.glob(sys.argv[1] + "/*counter_collection.csv")
if not f: print("no counters:", open(sys.argv[1] + "/stderr.txt").read()[-800:]); sys.exit()
for r in csv.DictReader(open(f[0])):
    if "k_cells_tile" in r["Kernel_Name"]:
        rows[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
sel = [d for d in rows.values() if d.get("SQ_WAVES") == 121776.0]
print({k: round(sum(d[k] for d in sel) / len(sel) / 1e6, 2) for k in sel[0] if k != "SQ_WAVES"}, "M per launch, n =", len(sel))
PY
done
