#!/usr/bin/env python3
"""Median per-launch value of every counter in rocprofv3 counter_collection.csv files, per kernel whose name
matches a pattern.  usage: pmc_diag.py PATTERN file.csv [file.csv ...]"""
import csv
import re
import statistics
import sys
from collections import defaultdict

pat = re.compile(sys.argv[1])
by = defaultdict(lambda: defaultdict(list))
for path in sys.argv[2:]:
    for r in csv.DictReader(open(path)):
        name = r['Kernel_Name']
        if not pat.search(name):
            continue
        m = re.findall(r'(\w+_kernel)', name)
        short = (m[0] if m else name[:40]) + (' +dense' if 'FmDenseExtra' in name else '')
        by[short][r['Counter_Name']].append(float(r['Counter_Value']))
for k in sorted(by):
    print(k)
    for c in sorted(by[k]):
        v = by[k][c]
        print("  %-44s median %16.1f  (n=%d)" % (c, statistics.median(v), len(v)))
