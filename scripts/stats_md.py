#!/usr/bin/env python3
"""rocprofv3 *_kernel_stats.csv -> markdown table of this library's kernels.
usage: stats_md.py stats.csv "title" "command" "workload" > out.md"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
print("# %s\n" % sys.argv[2])
print("Command (on the GPU box): `%s`." % sys.argv[3])
print("Workload: %s" % sys.argv[4])
print("Only this library's kernels are listed (torch's fill/random/copy kernels that build the synthetic data are omitted).\n")
print("| kernel | calls | avg us | min us | max us | total % |")
print("|---|---|---|---|---|---|")
for r in rows:
    n = r['Name']
    if 'at::native' in n or 'rocclr' in n or 'hiprand' in n.lower() or 'rocprim' in n:
        continue
    n = n.replace('(anonymous namespace)::', '').replace('void ', '')
    n = re.sub(r'\(.*', '', n)
    print("| %s | %s | %.2f | %.2f | %.2f | %s |" % (n, r['Calls'], float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3,
                                                   float(r['MaxNs']) / 1e3, r['Percentage']))
