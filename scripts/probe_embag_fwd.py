#!/usr/bin/env python3
"""Multi-hot embedding-bag forward: the wave-per-bag kernel (fixed bag length >= 16, few bags) against the
lane-group-per-bag kernel (reached here through the CSR form of the same bags) over batch sizes.
usage: probe_embag_fwd.py [dim]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch

from recommendation_amd import ops

dev = torch.device('cuda', 0)
D = int(sys.argv[1]) if len(sys.argv) > 1 else 128
T = 8
rows = [4000000] * T
toffs = ops.table_offsets(rows, dev)
g = torch.Generator(device=dev)
g.manual_seed(1)
W = torch.randn((sum(rows), D), generator=g, dtype=torch.float32, device=dev)
err = ops.new_error_word(dev)


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


print("dim %d, %d tables; us per call and algorithmic TB/s (rows + ids + output)" % (D, T))
for L in (20, 100):
    for B in (64, 256, 1024, 4096, 16384, 65536):
        if B * T * L > 120e6:
            continue
        ids = torch.randint(0, rows[0], (B, T, L), generator=g, dtype=torch.int64, device=dev)
        out = torch.empty((B, T * D), dtype=torch.float32, device=dev)
        offs = torch.arange(B * T + 1, dtype=torch.int64, device=dev) * L
        nbytes = B * T * L * (D * 4 + 8) + B * T * D * 4
        t_fixed = timed(lambda: ops.embag_fwd(W, toffs, T, ids, bag_len=L, check_flags=ops.MI_CHECK_RANGE, err=err, out=out))
        a = out.clone()
        t_csr = timed(lambda: ops.embag_fwd(W, toffs, T, ids.view(-1), offsets=offs, check_flags=ops.MI_CHECK_RANGE, err=err, out=out))
        rel = float((a - out).abs().max() / out.abs().max())
        G = 64 // (D // 4)
        wave = L >= 16 and L >= 4 * G and (B * T * (D // 4) < 8 * 64 * 4 * 256 or L >= 6 * G)
        print("L %3d B %6d: fixed-length call (%s) %8.1f us %5.2f TB/s | lane-group kernel %8.1f us %5.2f TB/s | max rel diff %.1e"
              % (L, B, 'wave per bag' if wave else 'lane group  ', t_fixed, nbytes / t_fixed / 1e6, t_csr, nbytes / t_csr / 1e6, rel))
ops.check_error_word(err)
