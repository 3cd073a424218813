#!/usr/bin/env python3
"""Random-segment read rate of the HBM as a function of segment size: the ceiling of a row gather at
that row width.  Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC scripts/seg_probe.hip -o scripts/exp/libsegprobe.so
usage: probe_segments.py [table_GiB=8] [GiB_read_per_launch=2]"""
import ctypes
import os
import sys

import torch

here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, 'exp', 'libsegprobe.so'))
lib.seg_probe.restype = ctypes.c_int
lib.seg_probe.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                          ctypes.c_void_p, ctypes.c_void_p]
gib = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0
per = float(sys.argv[2]) if len(sys.argv) > 2 else 2.0
dev = torch.device('cuda', 0)
nbytes = int(gib * (1 << 30))
table = torch.empty(nbytes // 4, dtype=torch.float32, device=dev)
for lo in range(0, table.numel(), 1 << 28):
    table[lo:lo + (1 << 28)].uniform_(0.0, 1.0)
out = torch.zeros(4, dtype=torch.float32, device=dev)
stream = torch.cuda.current_stream().cuda_stream
print("table %.1f GiB, %.1f GiB read per launch; GB/s by segment size (rows) and segments in flight per lane group (columns)" % (gib, per))
for mode, what in ((0, "read"), (1, "read-modify-write (bytes = read + written)"), (2, "write")):
  print("-- %s" % what)
  print("%8s %10s %10s %10s %10s   %s" % ("seg B", "R=1", "R=2", "R=4", "R=8", "best: G segments/s"))
  for S in (128, 512, 1024, 2048):
    n_seg = int(per * (1 << 30)) // S
    idx = torch.randperm(nbytes // S, device=dev)[:n_seg].to(torch.int32)   # distinct segments, random places
    row = []
    for R in (1, 2, 4, 8):
        def go():
            rc = lib.seg_probe(table.data_ptr(), idx.data_ptr(), n_seg, S, R, mode, out.data_ptr(), stream)
            assert rc == 0, rc
        go()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            go()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        row.append(n_seg * S * (2 if mode == 1 else 1) / ms / 1e6)
    print("%8d %10.0f %10.0f %10.0f %10.0f   %.2f" % (S, row[0], row[1], row[2], row[3], max(row) / S / (2 if mode == 1 else 1)), flush=True)
