#!/bin/bash
# GPU box: tools/fillbench/run.sh <experiment> [rounds]   -> gpurun_out/fillbench_<experiment>/
# builds the microbenchmark if needed, runs it under rocprofv3 --kernel-trace and summarises the device durations
# per kernel and cache state (cold-dirty / cold-clean / warm: the launch order inside each round).
set -o pipefail
EXP=${1:-all}; ROUNDS=${2:-10}
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
BIN=$REPO/tools/fillbench/fillbench
[ -x "$BIN" ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -o "$BIN" "$REPO/tools/fillbench/fillbench.hip" || exit 1
OUT=$REPO/gpurun_out/fillbench_$EXP; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
if [ "$EXP" = timeline ]; then
  # per-wave stamps inside the fold for 4 (the headline), 8 and 16 fields per launch: what is fixed (ramp, drain) and what scales
  "$BIN" timeline > "$OUT/log.txt" 2>&1; echo "rc=$?"
  for NFB in 8 16; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -DFB_NF=$NFB -o "$BIN.nf$NFB" "$REPO/tools/fillbench/fillbench.hip" || exit 1
    echo "==== $NFB fields per launch ====" >> "$OUT/log.txt"
    "$BIN.nf$NFB" timeline >> "$OUT/log.txt" 2>&1; echo "nf$NFB rc=$?"
  done
  cat "$OUT/log.txt"; exit 0
fi
if [ "$3" = pmc ]; then    # separate counter passes (kernel-trace only beside --pmc): bytes fetched / written per launch, by kernel
  for CNT in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $CNT --output-format csv -d "$OUT/$CNT" -o fb -- "$BIN" "$EXP" 3 > "$OUT/log_$CNT.txt" 2>&1; echo "pmc $CNT rc=$?"
  done
  python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
acc = collections.OrderedDict()
for cnt in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(sys.argv[1] + "/" + cnt + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == cnt and "k_per" in r["Kernel_Name"]:
                acc.setdefault(r["Kernel_Name"][:60], {}).setdefault(cnt, []).append(float(r["Counter_Value"]))
print("PMC per launch (KiB counters -> MB; FETCH_SIZE doubled for gfx950's 128-B requests, MI355X_MICROARCH.md):")
for k, d in acc.items():
    fe = 2 * 1024 * sum(d.get("FETCH_SIZE", [0])) / max(1, len(d.get("FETCH_SIZE", [0]))) / 1e6
    wr = 1024 * sum(d.get("WRITE_SIZE", [0])) / max(1, len(d.get("WRITE_SIZE", [0]))) / 1e6
    print(f"  {k:60s} fetch {fe:8.1f} MB  write {wr:8.1f} MB")
PY
  exit 0
fi
rocprofv3 --kernel-trace --output-format csv -d "$OUT" -o fb -- "$BIN" "$EXP" "$ROUNDS" > "$OUT/log.txt" 2>&1
echo "rocprofv3 rc=$?"
cat "$OUT/log.txt"
python3 - "$OUT" "$EXP" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
rows = sorted(csv.DictReader(open(f[0])), key=lambda r: int(r["Start_Timestamp"]))
acc = collections.OrderedDict()              # kernel -> cache state (= what ran just before) -> durations
prev = ""
for r in rows:
    n = r["Kernel_Name"]
    if not ("k_flush" in n or "k_init" in n or "fillBuffer" in n):
        mode = "cold-dirty" if "k_flush_dirty" in prev else ("cold-clean" if "k_flush_clean" in prev else ("after " + prev.split("(")[0][-28:] if sys.argv[2] == "seq" else "warm"))
        acc.setdefault(n, collections.OrderedDict()).setdefault(mode, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    prev = n
print("device durations (rocprofv3 kernel trace) by kernel and by what ran just before it; first 2 launches per state dropped:")
for n, modes in acc.items():
    out = []
    for mode, d in modes.items():
        t = sorted(d[2:]) or sorted(d)
        out.append(f"{mode} med {t[len(t)//2]:7.2f} min {t[0]:7.2f} n={len(t)}")
    print(f"  {n[:64]:64s} " + " | ".join(out))
PY
