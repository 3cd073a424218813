#!/usr/bin/env python3
"""Per-workgroup start/end times of the FM step's second launch (sparse update + dense side), from a library
built with -DSEG_TRACE (MI355REC_LIB points at it).  Prints, per class of workgroup (dense side, long-run
scan, 64-entry waves), when its workgroups start and finish relative to the launch's first start.
usage: MI355REC_LIB=.../libmi355rec_trace.so trace_fm_update.py [zipf]"""
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
B = 8192
dist = sys.argv[1] if len(sys.argv) > 1 else 'uniform'
m = FactorizationMachine(rows, 16, 13, device=dev, opt='adam_tf', lr=1e-3, seed=0)
batches = gen_batches(8, B, rows, 13, dist, 64, dev, seed=1000)
tms = [b[0].t().contiguous() for b in batches]
lib = _lib.load()
lib.mi_debug_seg_trace.restype = ctypes.c_int
lib.mi_debug_seg_trace.argtypes = [ctypes.c_void_p, ctypes.c_int]
n_dense_wg = 13 + 2
n_long = -(-B * len(rows) // int(os.environ.get('LONG_CHUNK', '1024')))
res = []
for i in range(24):
    j = i % 8
    m.train_step(*batches[j], ids_tm=tms[j])
    torch.cuda.synchronize()
    buf = np.zeros(4096 * 2, dtype=np.uint64)
    assert lib.mi_debug_seg_trace(buf.ctypes.data, 4096 * 2) == 0
    t = buf.reshape(4096, 2).astype(np.int64)
    n_wg = n_dense_wg + n_long + -(-B * len(rows) // 256)
    t = t[:n_wg]
    t0 = t[:, 0].min()
    st, en = (t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0       # 100 MHz counter -> us
    if i >= 8:
        res.append((st, en))
st = np.mean([r[0] for r in res], axis=0)
en = np.mean([r[1] for r in res], axis=0)
if os.environ.get('EXTRA_LAST'):      # library built with -DMI_EXTRA_LAST: the dense-side workgroups are the launch's last
    order = np.concatenate([np.arange(len(st) - n_dense_wg, len(st)), np.arange(0, len(st) - n_dense_wg)])
    st, en = st[order], en[order]
for name, lo, hi in (('dense side', 0, n_dense_wg), ('long-run scan', n_dense_wg, n_dense_wg + n_long),
                     ('64-entry waves', n_dense_wg + n_long, len(st))):
    s, e = st[lo:hi], en[lo:hi]
    print("%-16s %5d workgroups: start %5.2f .. %5.2f us, end median %5.2f p90 %5.2f max %5.2f us, duration median %5.2f max %5.2f"
          % (name, hi - lo, s.min(), s.max(), np.median(e), np.percentile(e, 90), e.max(), np.median(e - s), (e - s).max()))
lng = np.argsort(-(en[n_dense_wg:n_dense_wg + n_long]))[:12]
print("latest long-run workgroups (chunk: end us, duration us):", [(int(c), round(float(en[n_dense_wg + c]), 2), round(float(en[n_dense_wg + c] - st[n_dense_wg + c]), 2)) for c in lng])
sh = np.argsort(-(en[n_dense_wg + n_long:]))[:12]
print("latest wave workgroups (index: end us, duration us):", [(int(c), round(float(en[n_dense_wg + n_long + c]), 2), round(float(en[n_dense_wg + n_long + c] - st[n_dense_wg + n_long + c]), 2)) for c in sh])
print("launch span %.2f us" % en.max())
if hasattr(lib, 'mi_debug_seg_phases'):
    lib.mi_debug_seg_phases.restype = ctypes.c_int
    lib.mi_debug_seg_phases.argtypes = [ctypes.c_void_p, ctypes.c_int]
    ph = np.zeros(4096 * 8, dtype=np.uint64)
    assert lib.mi_debug_seg_phases(ph.ctypes.data, 4096 * 8) == 0
    ph = ph.reshape(4096, 8).astype(np.int64)
    buf = np.zeros(4096 * 2, dtype=np.uint64)
    lib.mi_debug_seg_trace(buf.ctypes.data, 4096 * 2)
    t = buf.reshape(4096, 2).astype(np.int64)
    print("long-run workgroups of the last step (chunk: start, scan done, heads ordered, wave-dealt heads done, end; us from launch start):")
    t0 = t[:n_dense_wg + n_long + 832, 0].min()
    for c in list(lng[:6]) + [0, 100]:
        w = (0 if os.environ.get('EXTRA_LAST') else n_dense_wg) + int(c)
        print("  chunk %3d: %5.2f %5.2f %5.2f %5.2f %5.2f" % (c, (t[w, 0] - t0) / 100.0, (ph[w, 0] - t0) / 100.0, (ph[w, 1] - t0) / 100.0,
                                                          (ph[w, 2] - t0) / 100.0, (t[w, 1] - t0) / 100.0))
