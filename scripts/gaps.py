#!/usr/bin/env python3
"""From a rocprofv3 kernel_trace.csv of bench.py (FM): the idle time between the step's two launches (inside a
graph replay) and between one step's last launch and the next step's first (between replays).
usage: gaps.py kernel_trace.csv"""
import csv
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'fm_fwd_sort_kernel' in r['Kernel_Name'] or 'FmDenseExtra' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
inner, outer = [], []
for a, b in zip(rows, rows[1:]):
    gap = (int(b['Start_Timestamp']) - int(a['End_Timestamp'])) / 1e3
    if gap > 50:
        continue
    if 'fm_fwd_sort' in a['Kernel_Name'] and 'FmDenseExtra' in b['Kernel_Name']:
        inner.append(gap)
    elif 'FmDenseExtra' in a['Kernel_Name'] and 'fm_fwd_sort' in b['Kernel_Name']:
        outer.append(gap)
for name, v in (('launch 1 -> launch 2 (inside a replay)', inner), ('launch 2 -> next launch 1 (between replays)', outer)):
    v = sorted(v)
    if v:
        print("%-46s n=%4d median %.2f us  p10 %.2f  p90 %.2f" % (name, len(v), v[len(v) // 2], v[len(v) // 10], v[9 * len(v) // 10]))
