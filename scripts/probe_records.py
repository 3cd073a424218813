#!/usr/bin/env python3
"""Read-modify-write rate of sorted distinct 256-byte FM records, and the forward's gather rate, by access shape.
Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC scripts/rec_probe.hip -o scripts/exp/librecprobe.so
usage: probe_records.py [rows_millions=33.76]"""
import ctypes
import os
import sys

import torch

here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, 'exp', 'librecprobe.so'))
lib.rec_probe.restype = ctypes.c_int
lib.rec_probe.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                          ctypes.c_void_p, ctypes.c_void_p]
rows = int(float(sys.argv[1]) * 1e6) if len(sys.argv) > 1 else 33762577
dev = torch.device('cuda', 0)
rec = torch.empty((rows, 64), dtype=torch.float32, device=dev)
for lo in range(0, rows, 1 << 22):
    rec[lo:lo + (1 << 22)].uniform_(0.0, 1.0)
out = torch.zeros(4, dtype=torch.float32, device=dev)
stream = torch.cuda.current_stream().cuda_stream
NSET = 16


def timeit(sets, shape, R, gather):
    def go(i):
        s = sets[i % NSET]
        rc = lib.rec_probe(rec.data_ptr(), s.data_ptr(), s.numel(), shape, R, gather, out.data_ptr(), stream)
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
    return e0.elapsed_time(e1) / (2 * NSET) * 1e3   # us


for U in (94053, 530919):
    g = torch.Generator(device=dev)
    g.manual_seed(U)
    sets = []
    for i in range(NSET):
        r = torch.randint(0, rows, (U,), device=dev, generator=g, dtype=torch.int64)
        r = torch.unique(r)
        sets.append(r.to(torch.int32).contiguous())   # sorted, distinct
    n = sets[0].numel()
    print("== read-modify-write of %d sorted distinct records (%.1f MB read + written at 512 B a record); us a launch, TB/s" % (n, n * 512 / 1e6))
    for shape, name in ((0, "4 lanes: V,m,v float4 + 3 dwords"), (1, "4 lanes: V,m,v float4 only"),
                        (2, "16 lanes: one float4 each"), (3, "16 lanes, 13 of them store")):
        line = []
        for R in (1, 2, 4, 8, 16):
            if shape <= 1 and R > 4:
                continue
            us = timeit(sets, shape, R, 0)
            line.append("R=%d %6.2f us %5.2f" % (R, us, n * 512 / us / 1e6))
        print("  %-36s %s" % (name, " | ".join(line)), flush=True)
    usets = [s[torch.randperm(s.numel(), device=dev)].contiguous() for s in sets]
    n2 = 26 * (8192 if U < 200000 else 65536)
    gsets = []
    for i in range(NSET):
        gsets.append(torch.randint(0, rows, (n2,), device=dev, generator=g, dtype=torch.int64).to(torch.int32).contiguous())
    print("== gather of %d random records' V + w1 (%.1f MB at 68 B); us a launch, TB/s" % (n2, n2 * 68 / 1e6))
    for shape, name in ((0, "4 lanes: float4 + dword w1"), (1, "4 lanes: float4 only"), (2, "5 of 8 lanes: float4")):
        line = []
        for R in (1, 2, 4, 8):
            us = timeit(gsets, shape, R, 1)
            line.append("R=%d %6.2f us %5.2f" % (R, us, n2 * 68 / us / 1e6))
        print("  %-36s %s" % (name, " | ".join(line)), flush=True)
