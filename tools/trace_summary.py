#!/usr/bin/env python3
"""Per-kernel duration summary of a rocprofv3 --kernel-trace CSV of bench.py (N = 1), restricted to the launches that belong
to bench STEPS: the plain `--stats` table also averages the auxiliary launches of the same kernels (cold / warm probes, the
Float32 and config-5 fills), which have other sizes and cache states.  Two launch sequences are recognised by their
neighbours in the trace:
  step       k_tables, k_cells_tile, k_halos, k_fill_merged                      (the timed steps and the instrumented pass)
  fold pass  k_tables, k_cells_tile, k_halos, k_zipper_cols, k_periodic_x_vec    (the pass behind `roofline_fold`)
A `step` whose kernels follow each other within 5 us is a TIMED step (the instrumented pass has stream markers between
its phases, ~10 us each); only those enter the `timed_step` rows, which are the figures to compare with `roofline`.  Marker-free
steps that come AFTER the fold pass are the K cold-onset steps of round 5 (`ms_per_step_cold_onset`: started right after 100 passes
over 1 GiB): they get their own `cold_onset_step` rows.
Beside those step sequences every launch of a HALO-FILL kernel gets a row of its own kind: the launches are grouped by kernel, template
arguments (element type, chunk width, Hy, GEN), number of fields (Grid_Size_Y) and grid width -- so the 8- and 16-field batched folds
behind `roofline_fold_batched` and the halo-(5, 5, 5) launches behind `fill_step_halo5` are rows of their own (`fill_launch` rows; their
medians are the figures to compare with bench.py's medians, both being cold launches).
usage: tools/trace_summary.py <bench_kernel_trace.csv> [out.csv]"""
import csv
import re
import statistics
import sys

STEP = ["k_tables", "k_cells_tile", "k_halos", "k_fill_merged"]
FOLD = ["k_tables", "k_cells_tile", "k_halos", "k_zipper_cols", "k_periodic_x_vec"]


def short(name):
    m = re.search(r"(k_[a-z_0-9]+)", name)
    return m.group(1) if m else name[:40]


def is_copy_probe(kernel_name):
    """k_zipper_cols<T, W, HY, COPY, GEN>: the same-shape copy probe of the test library has COPY = true"""
    args = kernel_name.split("k_zipper_cols<")[1].split(">")[0].replace(" ", "").split(",")
    return len(args) > 3 and args[3] == "true"


def main():
    rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
    names = [short(r["Kernel_Name"]) for r in rows]
    t0 = [int(r["Start_Timestamp"]) for r in rows]
    t1 = [int(r["End_Timestamp"]) for r in rows]
    dur = [(b - a) / 1e3 for a, b in zip(t0, t1)]
    acc = {"timed_step": {k: [] for k in STEP}, "instrumented_step": {k: [] for k in STEP}, "fold_pass": {k: [] for k in FOLD},
           "cold_onset_step": {k: [] for k in STEP}}
    i, seen_fold = 0, False
    while i < len(rows):
        if names[i:i + len(FOLD)] == FOLD and not is_copy_probe(rows[i + 3]["Kernel_Name"]):
            for k, d in zip(FOLD, dur[i:i + len(FOLD)]):
                acc["fold_pass"][k].append(d)
            i += len(FOLD)
            seen_fold = True
        elif names[i:i + len(STEP)] == STEP:
            gaps = [(t0[i + j + 1] - t1[i + j]) / 1e3 for j in range(len(STEP) - 1)]
            kind = ("cold_onset_step" if seen_fold else "timed_step") if max(gaps) < 5.0 else "instrumented_step"
            for k, d in zip(STEP, dur[i:i + len(STEP)]):
                acc[kind][k].append(d)
            i += len(STEP)
        else:
            i += 1
    # every halo-fill launch by (kernel<template>, fields, grid width)
    FILL = ("k_zipper_cols", "k_fill_merged", "k_fill_fused_vec", "k_fill_fused", "k_zipper_scalar", "k_zipper_vec", "k_periodic_x")
    fills = {}
    for r, d in zip(rows, dur):
        m = re.search(r"(k_[a-z_0-9]+)(<[^>]*>)?", r["Kernel_Name"])
        if not m or not m.group(1).startswith(FILL):
            continue
        key = (m.group(1) + (m.group(2) or "").replace(" ", ""), int(r["Grid_Size_Y"]), int(r["Grid_Size_X"]))
        fills.setdefault(key, []).append(d)
    out = [("sequence", "kernel", "launches", "avg_us", "median_us", "min_us", "max_us")]
    for kind, order in (("timed_step", STEP), ("instrumented_step", STEP), ("fold_pass", FOLD), ("cold_onset_step", STEP)):
        for k in order:
            d = acc[kind][k]
            if d:
                out.append((kind, k, len(d), round(statistics.mean(d), 3), round(statistics.median(d), 3), round(min(d), 3), round(max(d), 3)))
    for (name, ny, nx), d in sorted(fills.items()):
        out.append(("fill_launch", f"{name} fields={ny} grid_x={nx}", len(d), round(statistics.mean(d), 3), round(statistics.median(d), 3),
                    round(min(d), 3), round(max(d), 3)))
    w = csv.writer(open(sys.argv[2], "w", newline="") if len(sys.argv) > 2 else sys.stdout)
    w.writerows(out)


if __name__ == "__main__":
    main()
