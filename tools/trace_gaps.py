#!/usr/bin/env python3
"""Kernel timeline of a rocprofv3 --kernel-trace CSV: per plan() the launches, their durations and the gaps between them.
usage: trace_gaps.py <kernel_trace.csv> [first_row] [n_rows]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
first = int(sys.argv[2]) if len(sys.argv) > 2 else len(rows) // 2
n = int(sys.argv[3]) if len(sys.argv) > 3 else 12
prev_end = None
for r in rows[first:first + n]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    print(f"gap {gap:8.1f} us  dur {(e - s) / 1e3:7.1f} us  {r['Kernel_Name'][:70]}  grid {r.get('Grid_Size_X', r.get('Grid_Size',''))} wg {r.get('Workgroup_Size_X', r.get('Workgroup_Size',''))}")
    prev_end = e
