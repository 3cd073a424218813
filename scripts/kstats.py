#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel_trace.csv: per kernel name (and grid size) count / mean / min us."""
import csv
import re
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
agg = defaultdict(list)
for r in rows:
    name = r['Kernel_Name']
    if 'at::native' in name and len(sys.argv) < 3:
        continue
    m = re.findall(r'(\w+_kernel)', name)
    short = (m[0] if m else name[:40]) + ('<long>' if '<long' in name else '') + ('<9>' if '<9>' in name else '')
    key = (short, int(r['Grid_Size_X']) if 'Grid_Size_X' in r else int(r['Grid_Size']))
    agg[key].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for (name, grid), v in sorted(agg.items()):
    v2 = sorted(v)
    print("%-62s grid=%9d n=%4d mean=%8.2f med=%8.2f min=%8.2f us" % (name, grid, len(v), sum(v) / len(v), v2[len(v2) // 2], v2[0]))
