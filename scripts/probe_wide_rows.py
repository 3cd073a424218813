#!/usr/bin/env python3
"""Read-modify-write rate of sorted distinct 512-byte rows with a per-row fp32 state word, by where the word lives.
Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC scripts/wide_probe.hip -o scripts/exp/libwideprobe.so
usage: probe_wide_rows.py [rows_millions=125]"""
import ctypes
import os
import sys

import torch

here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, 'exp', 'libwideprobe.so'))
lib.wide_probe.restype = ctypes.c_int
lib.wide_probe.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                           ctypes.c_void_p]
rows = int(float(sys.argv[1]) * 1e6) if len(sys.argv) > 1 else 125000000
dev = torch.device('cuda', 0)
stream = torch.cuda.current_stream().cuda_stream
NSET = 8
U = 65536 * 26
g = torch.Generator(device=dev)
g.manual_seed(1)
sets = [torch.unique(torch.randint(0, rows, (U,), device=dev, generator=g, dtype=torch.int64)).to(torch.int32).contiguous()
        for _ in range(NSET)]
n = sets[0].numel()
print("%d sorted distinct rows of %d M; us a launch, TB/s of the 1032 algorithmic bytes a row (512 + 4, read and written)" % (n, rows // 1000000))
for name, stride, mode in (("rows only, no state", 128, 0), ("rows + separate state array", 128, 1),
                           ("record of 132 floats, state behind the row", 132, 2),
                           ("record of 160 floats (line-aligned), state behind the row", 160, 2)):
    w = torch.empty((rows * stride,), dtype=torch.float32, device=dev)
    for lo in range(0, w.numel(), 1 << 28):
        w[lo:lo + (1 << 28)].fill_(0.5)
    state = torch.ones((rows,), dtype=torch.float32, device=dev)

    def go(i):
        s = sets[i % NSET]
        rc = lib.wide_probe(w.data_ptr(), state.data_ptr(), s.data_ptr(), s.numel(), stride, mode, stream)
        assert rc == 0, rc
    for i in range(NSET):
        go(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(2 * NSET):
        go(i)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / (2 * NSET) * 1e3
    print("%-62s %8.1f us  %5.2f TB/s" % (name, us, n * 1032 / us / 1e6))
    del w, state
    torch.cuda.empty_cache()
