#!/usr/bin/env python3
"""Which part of the wide-row sparse update costs what: the same sorted plan through
  (a) mi_embag_bwd_dense_grad   - gradient rows read, summed rows WRITTEN (no read-modify-write),
  (b) mi_embag_bwd_apply_sgd    - + read-modify-write of the weight rows,
  (c) ..._adagrad_rowwise / (d) ..._adam_tf.
usage: probe_embag_update.py [rows_total=15600000] [batch=65536]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recommendation_amd import ops  # noqa: E402

total_req = int(sys.argv[1]) if len(sys.argv) > 1 else 15600000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
T, D = 26, 128
dev = torch.device('cuda', 0)
rows = [total_req // T] * T
total = sum(rows)
toffs = ops.table_offsets(rows, dev)
g = torch.Generator(device=dev)
g.manual_seed(1)
W = torch.randn((total, D), generator=g, device=dev) * 0.05
ids = torch.stack([torch.randint(0, r, (B,), generator=g, device=dev) for r in rows], dim=1).contiguous()
grad = torch.randn((B, T * D), generator=g, device=dev) * 1e-3
plan = ops.SortedPlan(toffs, T, total, ids, bag_len=1, runs=False)
n = B * T
U = int(torch.unique(plan.sorted_rows[:n]).numel())
st = torch.zeros((total,), device=dev)
m, v = torch.zeros_like(W), torch.zeros_like(W)
dW = torch.empty((total, D), device=dev)


def dense_grad():
    from recommendation_amd import _lib
    _lib.call('mi_embag_bwd_dense_grad', plan.ref(), grad.data_ptr(), grad.shape[-1], D, total, dW.data_ptr(),
              torch.cuda.current_stream().cuda_stream)


def timed(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


print("rows %d x dim %d, batch %d, entries %d, distinct rows %d" % (total, D, B, n, U))
seg = D * 4
ms0 = timed(lambda: dW.zero_())
print("%-40s %8.3f ms  (the zero fill mi_embag_bwd_dense_grad starts with; subtract it from the next line)" % ('memset dW', ms0))
for name, fn, nbytes in (
        ('dense_grad (zero fill + read grad, write rows)', dense_grad, n * (8 + seg) + U * seg),
        ('sgd (+ RMW weights)', lambda: ops.bwd_apply_sgd(plan, grad, W, 0.01), n * (8 + seg) + U * seg * 2),
        ('adagrad_rowwise', lambda: ops.bwd_apply_adagrad_rowwise(plan, grad, W, st, 0.01, 1e-8),
         n * (8 + seg) + U * (seg + 4) * 2),
        ('adam_tf', lambda: ops.bwd_apply_adam_tf(plan, grad, W, m, v, 1e-3, 0.9, 0.999, 1e-8),
         n * (8 + seg) + U * seg * 6)):
    ms = timed(fn)
    print("%-40s %8.3f ms  %7.0f GB/s algorithmic" % (name, ms, nbytes / ms / 1e6), flush=True)
