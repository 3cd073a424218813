#!/usr/bin/env python3
"""Phase times inside the FM step's first launch (forward workgroups + per-table sort workgroups), library built with
-DTS_TRACE by scripts/build_fm_trace.sh and MI355REC_LIB pointing at it.  usage: MI355REC_LIB=.../libmi_tstrace.so trace_fm_launch1.py [batch]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
import torch

from bench import CRITEO_KAGGLE, gen_batches
from recommendation_amd import _lib
from recommendation_amd.fm import FactorizationMachine

dev = torch.device('cuda', 0)
rows = CRITEO_KAGGLE
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
model = FactorizationMachine(rows, 16, 13, dev, opt='adam_tf', lr=1e-3, seed=0)
batches = gen_batches(4, B, rows, 13, 'uniform', 64, dev, seed=1000)
tms = [b[0].t().contiguous() for b in batches]
lib = _lib.load()
lib.mi_debug_ts_trace.restype = ctypes.c_int
lib.mi_debug_ts_trace.argtypes = [ctypes.c_void_p, ctypes.c_int]
names = {0: 'start', 1: 'ids', 2: 'cuts', 3: 'filled', 4: 'pass 1', 5: 'pass 2', 6: 'pass 3', 7: 'pass 4', 8: 'done', 11: 'ranked', 12: 'rank barrier', 13: 'p1 fill barrier', 14: 'p1 ranked', 15: 'p1 barrier'}
for it in range(6):
    b = batches[it % 4]
    model.train_step(b[0], b[1], b[2], ids_tm=tms[it % 4])
    torch.cuda.synchronize()
    if it < 3:
        continue
    buf = np.zeros(1024 * 16, dtype=np.uint64)
    assert lib.mi_debug_ts_trace(buf.ctypes.data, 1024 * 16) == 0
    t = buf.reshape(1024, 16).astype(np.int64)
    nf = (B + 63) // 64
    live = t[:, 8] > 0
    t0 = t[live, 0].min()
    f = np.arange(1024) < nf  # the forward workgroups come first in the grid
    fw, so = live & f, live & ~f
    print("step %d: launch span %.2f us; forward workgroups %d: start median %.2f max %.2f, done median %.2f max %.2f; sort workgroups %d" % (
        it, (t[live, 8].max() - t0) / 100.0, int(fw.sum()), np.median(t[fw, 0] - t0) / 100.0, (t[fw, 0].max() - t0) / 100.0,
        np.median(t[fw, 8] - t0) / 100.0, (t[fw, 8].max() - t0) / 100.0, int(so.sum())))
    for k, nm in ((1, 'ids known'), (2, 'rows summed'), (3, 'dense added')):
        ok = fw & (t[:, k] > 0)
        if ok.any():
            v = (t[ok, k] - t0) / 100.0
            print("  forward %-12s median %6.2f  p90 %6.2f  max %6.2f" % (nm, np.median(v), np.percentile(v, 90), v.max()))
    for k in sorted(names):
        ok = so & (t[:, k] > 0)
        if ok.any():
            v = (t[ok, k] - t0) / 100.0
            print("  %-12s median %6.2f  p90 %6.2f  max %6.2f   (%d workgroups)" % (names[k], np.median(v), np.percentile(v, 90), v.max(), int(ok.sum())))
    late = [b_ for b_ in np.argsort(-t[:, 8]) if so[b_]][:5]
    for b_ in late:
        print("  block %3d:" % b_, " ".join("%s %.2f" % (names[k], (t[b_, k] - t0) / 100.0) for k in sorted(names) if t[b_, k] > 0), "m", t[b_, 10], "bits", t[b_, 9])
