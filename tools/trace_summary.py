#!/usr/bin/env python3
"""Per-kernel duration summary of a rocprofv3 --kernel-trace CSV of bench.py, restricted to the launches that belong to
bench STEPS (build -> zipper -> periodic x): the plain `--stats` table also averages the auxiliary launches of the same
kernels (cold / warm probes, the config-5 fill_step), which have other sizes and cache states.
A step launch is recognised by its neighbours in the trace: k_tables, k_cells_tile, k_halos, k_zipper_cols, k_periodic_x_vec
in this order.  usage: tools/trace_summary.py <bench_kernel_trace.csv> [out.csv]"""
import csv
import re
import statistics
import sys

ORDER = ["k_tables", "k_cells_tile", "k_halos", "k_zipper_cols", "k_periodic_x_vec"]


def short(name):
    m = re.search(r"(k_[a-z_0-9]+)", name)
    return m.group(1) if m else name[:40]


def main():
    rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
    names = [short(r["Kernel_Name"]) for r in rows]
    dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
    acc = {k: [] for k in ORDER}
    gaps = []
    i = 0
    while i + len(ORDER) <= len(rows):
        if names[i:i + len(ORDER)] == ORDER and "true>" not in rows[i + 3]["Kernel_Name"].split("k_zipper_cols")[1][:24]:
            for k, d in zip(ORDER, dur[i:i + len(ORDER)]):
                acc[k].append(d)
            gaps.append((int(rows[i + 4]["End_Timestamp"]) - int(rows[i]["Start_Timestamp"])) / 1e3 - sum(dur[i:i + 5]))
            i += len(ORDER)
        else:
            i += 1
    out = [("kernel", "step_launches", "avg_us", "median_us", "min_us", "max_us")]
    for k in ORDER:
        d = acc[k]
        out.append((k, len(d), round(statistics.mean(d), 3), round(statistics.median(d), 3), round(min(d), 3), round(max(d), 3)))
    out.append(("(gaps between the 5 kernels of a step)", len(gaps), round(statistics.mean(gaps), 3), round(statistics.median(gaps), 3), round(min(gaps), 3), round(max(gaps), 3)))
    w = csv.writer(open(sys.argv[2], "w", newline="") if len(sys.argv) > 2 else sys.stdout)
    w.writerows(out)


if __name__ == "__main__":
    main()
