#!/usr/bin/env python3
"""FM step stages with the small tables enlarged: probe_fm_rows.py MINROWS (how much of launch 2 is long runs)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch

from bench import CRITEO_KAGGLE, gen_batches, stage_report
from recommendation_amd.fm import FactorizationMachine

minrows = int(sys.argv[1]) if len(sys.argv) > 1 else 1
rows = [max(r, minrows) for r in CRITEO_KAGGLE]
dev = torch.device('cuda', 0)
m = FactorizationMachine(rows, 16, 13, device=dev, opt='adam_tf', lr=1e-3, seed=0)
batches = gen_batches(2, 8192, rows, 13, 'uniform', 64, dev, seed=5)
for _ in range(5):
    m.train_step(*batches[0])
stages, U = stage_report(m, batches[0], 50, 64)
print("min rows", minrows, "unique", U)
for s in stages:
    print("%-75s %7.2f us" % (s['name'], s['ms'] * 1e3))
