#!/usr/bin/env python3
"""Where the fixed cost of one timed graph replay goes (FM, batch 8192, K consecutive steps in one graph): wall time of
replay + synchronise against the time between two events recorded around the replay on its stream.
usage: probe_replay_overhead.py [K]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
import torch

from bench import CRITEO_KAGGLE, gen_batches
from recommendation_amd.fm import FactorizationMachine

K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device('cuda', 0)
rows = CRITEO_KAGGLE
model = FactorizationMachine(rows, 16, 13, dev, opt='adam_tf', lr=1e-3, seed=0)
batches = gen_batches(8, 8192, rows, 13, 'uniform', 64, dev, seed=1000)
tms = [b[0].t().contiguous() for b in batches]
full = [batches[j % 8] for j in range(K)]
ftm = [tms[j % 8] for j in range(K)]
for _ in range(3):
    model.train_steps(full, ids_tms=ftm)
torch.cuda.synchronize()
wall, issue, gpu = [], [], []
for _ in range(30):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    a.record()
    model.train_steps(full, ids_tms=ftm)
    b.record()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    wall.append((t2 - t0) * 1e6)
    issue.append((t1 - t0) * 1e6)
    gpu.append(a.elapsed_time(b) * 1e3)
print("K=%d: wall %.1f us (%.2f a step), host until the replay call returned %.1f us, event to event on the stream %.1f us (%.2f a step)"
      % (K, np.median(wall), np.median(wall) / K, np.median(issue), np.median(gpu), np.median(gpu) / K))
