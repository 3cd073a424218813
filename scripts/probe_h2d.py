#!/usr/bin/env python3
"""Host-to-device copy time of pinned blocks by size (torch copy_ non_blocking on one stream; 20 back-to-back copies,
then one synchronize): the fixed cost per copy and the rate of the link as this box gives them."""
import time

import torch

dev = torch.device('cuda', 0)
for nbytes in (64 << 10, 1310720, 4 << 20, 16 << 20, 64 << 20, 256 << 20):
    h = torch.empty((nbytes,), dtype=torch.uint8).pin_memory()
    d = torch.empty((nbytes,), dtype=torch.uint8, device=dev)
    for _ in range(3):
        d.copy_(h, non_blocking=True)
    torch.cuda.synchronize()
    n = 20
    t0 = time.perf_counter()
    for _ in range(n):
        d.copy_(h, non_blocking=True)
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print("%10d bytes: %8.1f us per copy (host issue %6.1f us)  %6.1f GB/s" % (nbytes, dt * 1e6, t_issue / n * 1e6, nbytes / dt / 1e9), flush=True)
