#!/usr/bin/env python3
"""Host-side profile (cProfile) of the row-wise sharded FM step at world size 1: where the time the host needs to
enqueue a step goes (ctypes calls, torch.distributed collectives, stream / event bookkeeping).
usage: host_profile_sharded.py [steps]"""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch

from bench import CRITEO_KAGGLE, gen_batches

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
os.environ.setdefault('MASTER_PORT', '29578')
os.environ.setdefault('RANK', '0')
os.environ.setdefault('WORLD_SIZE', '1')
torch.cuda.set_device(0)
dev = torch.device('cuda', 0)
torch.distributed.init_process_group('nccl', device_id=dev)
from recommendation_amd.sharded import ShardedFactorizationMachine

m = ShardedFactorizationMachine(CRITEO_KAGGLE, 16, 13, device=dev, opt='adam_tf', lr=1e-3, seed=0)
batches = gen_batches(8, 8192, CRITEO_KAGGLE, 13, 'uniform', 64, dev, seed=1000)
for i in range(50):
    m.train_step(*batches[i % 8])
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for i in range(steps):
    m.train_step(*batches[i % 8])
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr, stream=sys.stderr)
st.sort_stats('cumulative').print_stats(28)
torch.distributed.destroy_process_group()
