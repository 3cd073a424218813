"""Wide-row update: the same launch in fresh processes, with the table and the state array allocated in different
orders / ways.  usage: probe_adagrad_modes.py VARIANT   (a: table then state, temporaries while filling;
b: state first; c: table filled in place, no temporaries; d: one allocation holding table + state)"""
import sys
import time

import torch

sys.path.insert(0, '.')
from recommendation_amd import ops  # noqa: E402

variant = sys.argv[1] if len(sys.argv) > 1 else 'a'
dev = torch.device('cuda', 0)
T, L, D, B = 26, 1, 128, 65536
rows = [125000000 // T] * T
total = sum(rows)
toffs = ops.table_offsets(rows, dev)
g = torch.Generator(device=dev)
g.manual_seed(1000)
state = None
pre = None
if variant == 'e':  # every small buffer before the table
    state = torch.zeros((total,), dtype=torch.float32, device=dev)
    pre = (torch.empty((B, T * D), dtype=torch.float32, device=dev), torch.empty((B, T * D), dtype=torch.float32, device=dev))
if variant == 'b':
    state = torch.zeros((total,), dtype=torch.float32, device=dev)
if variant == 'd':
    blob = torch.empty((total * (D + 1) + 64,), dtype=torch.float32, device=dev)
    W = blob[:total * D].view(total, D)
    state = blob[total * D:total * D + total]
    state.zero_()
else:
    W = torch.empty((total, D), dtype=torch.float32, device=dev)
chunk = 1 << 22
for lo in range(0, total, chunk):
    hi = min(lo + chunk, total)
    if variant in ('c', 'd'):
        W[lo:hi].normal_(generator=g).mul_(0.05)
    else:
        W[lo:hi] = torch.randn((hi - lo, D), generator=g, dtype=torch.float32, device=dev).mul_(0.05)
if state is None:
    state = torch.zeros((total,), dtype=torch.float32, device=dev)
card = torch.tensor(rows, dtype=torch.float64, device=dev).view(1, T, 1)
u = torch.rand((B, T, L), generator=g, dtype=torch.float64, device=dev)
ids = torch.minimum(torch.floor(u * card).clamp_(min=0), card - 1).to(torch.int64).contiguous()
if pre is not None:
    grad, out = pre
    grad.normal_(generator=g).mul_(1e-3)
else:
    grad = torch.randn((B, T * D), generator=g, dtype=torch.float32, device=dev).mul_(1e-3)
    out = torch.empty((B, T * D), dtype=torch.float32, device=dev)
err = ops.new_error_word(dev)
plan = ops.SortedPlan(toffs, T, total, ids, bag_len=L, runs=False)
plan.prepare(ids)


def timed(fn, reps=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


f = timed(lambda: ops.embag_fwd(W, toffs, T, ids, bag_len=L, check_flags=ops.MI_CHECK_RANGE, err=err, out=out))
u_ = timed(lambda: ops.bwd_apply_adagrad_rowwise(plan, grad, W, state, 0.01, 1e-8))
print('variant %s  W %#x  state %#x  grad %#x  fwd %.4f ms  update %.4f ms' % (variant, W.data_ptr(), state.data_ptr(), grad.data_ptr(), f, u_),
      flush=True)
