#!/usr/bin/env python3
"""Times the four launches of the sharded FM step of ONE rank of an N-rank job on a single GPU:
the other ranks' traffic is produced by routing N different batches locally.
usage: probe_shard.py [N] [B]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch

from bench import CRITEO_KAGGLE, gen_batches
from recommendation_amd.sharded import ShardedFactorizationMachine

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
dev = torch.device('cuda', 0)
rows = CRITEO_KAGGLE
if len(sys.argv) > 3:      # probe_shard.py N B MINROWS: tables smaller than MINROWS are enlarged to it
    rows = [max(r, int(sys.argv[3])) for r in rows]
m = ShardedFactorizationMachine(rows, 16, 13, device=dev, opt='adam_tf', lr=1e-3, rank=0, world=N)
batches = gen_batches(N, B, rows, 13, 'uniform', 64, dev, seed=5)
for s, (ids, dense, labels) in enumerate(batches):      # what rank s would send to rank 0
    m.phase_route(ids)
    m.recv[s].copy_(m.send[0])
ids, dense, labels = batches[0]


def timed(name, fn, reps=30):
    for _ in range(3):
        fn()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record()
    for i in range(reps):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    ts = sorted(ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(reps))
    print("%-46s median %7.1f us  min %7.1f us" % (name, ts[len(ts) // 2], ts[0]))


timed("route", lambda: m.phase_route(ids))
be = m.be
timed("owner forward alone", lambda: be.owner_forward(m.recv, B, m.rs, m.partial, None, None, None, None, m.lr, m.beta1, m.beta2))
timed("owner forward + sort", lambda: m.phase_owner_forward())
m.partials_in.copy_(m.partial)
timed("combine", lambda: m.phase_combine(dense, labels))
for s in range(N):
    m.X_all[s * m.rs:(s + 1) * m.rs].copy_(m.X)
timed("owner backward + dense side", lambda: m.phase_update())
valid = int(m.n_total.item())
keep = m.n_total.clone()
m.n_total.zero_()
timed("  dense side alone (no sparse entries)", lambda: m.phase_update())
m.n_total.copy_(keep)
print("N=%d B=%d valid entries at this owner: %d (%.3f of N*B*T)" % (N, B, valid, valid / (N * B * len(rows))))
m.be.check()
nc = -(-B // 64)
cnt = m.recv[:, :nc * len(rows)].reshape(N, nc, len(rows)).sum((0, 1)).cpu().tolist()
print("entries per table:", cnt)
