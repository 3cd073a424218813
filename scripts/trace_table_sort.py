#!/usr/bin/env python3
"""Phase times of the per-table LDS sort's workgroups (library built with -DTS_TRACE, MI355REC_LIB points at it):
start, cuts known, counted, filled, radix passes, written - microseconds from the launch's first start.
usage: MI355REC_LIB=.../libmi_tstrace.so trace_table_sort.py [batch]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
import torch

from bench import CRITEO_KAGGLE, gen_batches
from recommendation_amd import _lib, ops

dev = torch.device('cuda', 0)
rows = [int(x) for x in os.environ['ROWS'].split(',')] if os.environ.get('ROWS') else CRITEO_KAGGLE
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
T = len(rows)
toffs = ops.table_offsets(rows, dev)
batches = gen_batches(2, B, rows, 13, 'uniform', 64, dev, seed=1000)
lib = _lib.load()
lib.mi_debug_ts_trace.restype = ctypes.c_int
lib.mi_debug_ts_trace.argtypes = [ctypes.c_void_p, ctypes.c_int]
ids = batches[0][0]
tm = ids.t().contiguous()
plan = ops.SortedPlan(toffs, T, int(sum(rows)), ids, bag_len=1, runs=False)
for i in range(3):
    ops.fm_sort_ids(toffs, ids, plan, ids_tm=tm)
    torch.cuda.synchronize()
buf = np.zeros(1024 * 16, dtype=np.uint64)
assert lib.mi_debug_ts_trace(buf.ctypes.data, 1024 * 16) == 0
t = buf.reshape(1024, 16).astype(np.int64)
live = t[:, 8] > 0
t0 = t[live, 0].min()
names = {0: 'start', 1: 'cuts', 2: 'counted', 3: 'filled', 4: 'pass 1', 5: 'pass 2', 6: 'pass 3', 7: 'pass 4', 8: 'written'}
print("B=%d: %d workgroups wrote; launch span %.2f us" % (B, int(live.sum()), (t[live, 8].max() - t0) / 100.0))
for k in sorted(names):
    v = (t[live, k] - t0) / 100.0
    ok = t[live, k] > 0
    if ok.any():
        print("  %-8s median %7.2f  p90 %7.2f  max %7.2f   (%d workgroups)" % (names[k], np.median(v[ok]), np.percentile(v[ok], 90), v[ok].max(), int(ok.sum())))
print("  entries per workgroup: median %d max %d; key bits median %d max %d" % (np.median(t[live, 10]), t[live, 10].max(), np.median(t[live, 9]), t[live, 9].max()))
late = np.argsort(-t[:, 8])[:6]
for b in late:
    print("  block %3d:" % b, " ".join("%s %.2f" % (names[k], (t[b, k] - t0) / 100.0) for k in sorted(names) if t[b, k] > 0), "m", t[b, 10], "bits", t[b, 9])
