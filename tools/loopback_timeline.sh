#!/bin/bash
# GPU box: the N > 1 step of bench.py on the loop-back communicator under rocprofv3 --kernel-trace, for the kernel timeline of
# one overlapped step (which kernel runs when, on which queue), and three A/B runs of the step time:
# overlap off / on at default priority / on at high priority.
# usage: tools/loopback_timeline.sh <tag> [band]
TAG=${1:-lb}; BAND=${2:-3}
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/$TAG; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for cfg in "0 0" "1 0" "1 -1"; do
  set -- $cfg
  TPG_BENCH_OVERLAP=$1 TPG_BENCH_SIDE_PRIORITY=$2 python3 "$REPO/bench.py" --loopback --loopback-band $BAND --steps 100 --warmup 20 --exchange monolithic > "$OUT/ab_overlap$1_prio$2.json" 2> "$OUT/ab_overlap$1_prio$2.err"
  echo "overlap=$1 prio=$2 rc=$?"
done
rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -o lb -- python3 "$REPO/bench.py" --loopback --loopback-band $BAND --steps 10 --warmup 3 --exchange monolithic > "$OUT/trace_bench.json" 2> "$OUT/trace.err"
echo "trace rc=$?"
python3 - "$OUT" <<'PY'
import csv, glob, json, sys, re
out = sys.argv[1]
for f in sorted(glob.glob(out + "/ab_*.json")):
    d = json.loads([l for l in open(f) if l.startswith('{"metric"')][0])
    print(f.split("/")[-1], "ms_per_step %.4f build %.4f local %.4f exch %.4f" % (d["ms_per_step"], d["precompute_ms"], d["fill_bracket_ms"], d["exchange_ms"]))
files = glob.glob(out + "/trace/**/*kernel_trace.csv", recursive=True)
rows = sorted(csv.DictReader(open(files[0])), key=lambda r: int(r["Start_Timestamp"]))
# the timed steps: find the last 60 kernels before the instrumented passes; print a window of the trace relative to its first start
def short(n):
    m = re.search(r"(k_[a-z_0-9]+|rccl\w+|nccl\w+)", n)
    return m.group(1) if m else n[:40]
names = [short(r["Kernel_Name"]) for r in rows]
idx = [i for i, n in enumerate(names) if n == "k_cells_tile"]
i0 = idx[len(idx) // 3] - 6
t_ref = int(rows[i0]["Start_Timestamp"])
with open(out + "/timeline.txt", "w") as fo:
    for r, n in zip(rows[i0:i0 + 40], names[i0:i0 + 40]):
        line = "%9.2f %9.2f %8.2f  q%-3s %s" % ((int(r["Start_Timestamp"]) - t_ref) / 1e3, (int(r["End_Timestamp"]) - t_ref) / 1e3,
                                              (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Queue_Id", "?"), n)
        print(line); fo.write(line + "\n")
PY
