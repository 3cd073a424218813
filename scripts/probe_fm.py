#!/usr/bin/env python3
"""Probe: run the FM step stages at several batch sizes so that a rocprofv3 --kernel-trace of
this script shows how each kernel's duration scales (latency- or throughput-bound).

  rocprofv3 --kernel-trace --stats --output-format csv -d out -- python scripts/probe_fm.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import CRITEO_KAGGLE, gen_batches  # noqa: E402
from recommendation_amd.fm import FactorizationMachine  # noqa: E402


def main():
    dev = torch.device('cuda:0')
    opt = sys.argv[1] if len(sys.argv) > 1 else 'adam_tf'
    sizes = [int(x) for x in sys.argv[2].split(',')] if len(sys.argv) > 2 else [2048, 8192, 32768, 131072]
    fm = FactorizationMachine(CRITEO_KAGGLE, 16, 13, device=dev, opt=opt)
    for B in sizes:
        batches = gen_batches(4, B, CRITEO_KAGGLE, 13, 'uniform', 64, dev, seed=B)
        for i in range(12):
            fm.train_step(*batches[i % 4])
        torch.cuda.synchronize()
        print("B", B, "loss", float(fm.loss.item()))


if __name__ == '__main__':
    main()
