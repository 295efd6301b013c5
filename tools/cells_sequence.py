#!/usr/bin/env python3
"""Sequence of k_cells_tile launch durations (and the gap before each) from a rocprofv3 --kernel-trace CSV of bench.py:
how long do the first steps after an idle device take, compared with the steady state?  usage: cells_sequence.py <kernel_trace.csv>"""
import csv, re, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
prev_end = None
print("t_ms  kernel  dur_us  gap_before_us")
for r in rows:
    m = re.search(r"(k_[a-z_0-9]+)", r["Kernel_Name"])
    n = m.group(1) if m else r["Kernel_Name"][:24]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if n in ("k_cells_tile", "k_fill_merged", "k_zipper_cols", "k_convert_frame_vec"):
        print(f"{(s - t0) / 1e6:10.3f} {n:20s} {(e - s) / 1e3:8.1f} {((s - prev_end) / 1e3) if prev_end else 0:10.1f}")
    prev_end = e
