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
# the halo fills at the reference's model halo (5, 5, 5) (bench_halo5.py, headline size only): FETCH / WRITE of the GEN kernels
run5() {
  local name=$1; shift
  local OUT=$REPO/gpurun_out/pmc_${TAG}_halo5_$name; rm -rf "$OUT"; mkdir -p "$OUT"
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT" -o p -- python3 "$REPO/bench_halo5.py" --no-config5 > "$OUT/halo5.json" 2> "$OUT/stderr.txt"
  echo "== halo5 $name rc=$?"
}
run sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE
run fetch FETCH_SIZE
run write WRITE_SIZE
run5 fetch FETCH_SIZE
run5 write WRITE_SIZE
python3 - "$REPO/gpurun_out" "$TAG" <<'PY'
import csv, sys, glob, collections, json, re
root, tag = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for name in ("fetch", "write"):
    for f in glob.glob(f"{root}/pmc_{tag}_halo5_{name}/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            m = re.search(r"(k_(?:fill_merged|zipper_cols)<[^>]*>)", r["Kernel_Name"])
            if m:
                acc[m.group(1).replace(" ", "") + f" fields={r['Grid_Size_Y']}" if 'Grid_Size_Y' in r else m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, cs in sorted(acc.items()):
    f = cs.get("FETCH_SIZE", [0])[1:] or [0]; w = cs.get("WRITE_SIZE", [0])[1:] or [0]
    fb, wb = sum(f) / len(f) * 1024, sum(w) / len(w) * 1024
    out[k] = {"fetch_bytes_raw": fb, "write_bytes": wb, "hbm_bytes_per_launch": 2 * fb + wb, "launches": len(f)}
    print("halo5", k, {kk: round(v / 1e6, 2) if kk != "launches" else v for kk, v in out[k].items()})
json.dump({"_source": "tools/pmc.sh: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over bench_halo5.py --no-config5; KiB counters, FETCH doubled (gfx950), first launch dropped", "kernels": out},
          open(f"{root}/{tag}_halo5_traffic.json", "w"), indent=1)
PY
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
