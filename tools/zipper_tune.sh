#!/bin/bash
# usage (GPU box): tools/zipper_tune.sh 0 1 2 3 4
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/ztune; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$OUT" -o z -- python3 "$REPO/tools/zipper_tune.py" "$@" > "$OUT/log.txt" 2>&1
python3 - "$OUT/z_kernel_trace.csv" "$@" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_zipper" in r["Kernel_Name"]]
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
variants = sys.argv[2:]; R = len(d) // (3 * len(variants)); B = 2 * 17.28e6 + 2 * 19.44e6
for vi, v in enumerate(variants):
    seg = d[vi * 3 * R:(vi + 1) * 3 * R]
    for mi, mode in enumerate(("cold-dirty", "cold-clean", "warm")):
        ts = sorted(seg[mi::3][2:]); med = ts[len(ts) // 2]
        print(f"variant {v} {mode:10s}: median {med:6.2f} us  min {ts[0]:6.2f}  -> {B / med / 1e3:5.0f} GB/s = {B / med / 1e3 / 80:4.1f}% of 8 TB/s")
PY
