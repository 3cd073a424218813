"""Time one DCN-v2 training step at the benchmark shape (Criteo Kaggle tables, B = 65536)."""
import sys
import time
import numpy as np
import torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recommendation_amd.dcn import DCNv2

CRITEO_KAGGLE = [1460, 583, 10131227, 2202608, 305, 24, 12517, 633, 3, 93145, 5683, 8351593, 3194, 27, 14992,
                 5461306, 10, 5652, 2173, 4, 7046547, 18, 15, 286181, 105, 142572]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
dev = torch.device('cuda:0')
m = DCNv2(CRITEO_KAGGLE, 16, 13, dev)
g = torch.Generator(device='cpu').manual_seed(0)
ids = torch.stack([torch.randint(0, r, (B,), generator=g) for r in CRITEO_KAGGLE], dim=1).to(dev)
dense = torch.rand((B, 13), generator=g).to(dev)
labels = (torch.rand((B,), generator=g) < 0.25).float().to(dev)
for _ in range(3):
    loss = m.train_step(ids, dense, labels)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 10
for _ in range(n):
    loss = m.train_step(ids, dense, labels)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
fl = m.flops_per_sample() * B
print("B=%d step %.3f ms  %.2f M samples/s  %.0f TFLOP/s (contractions only)  loss %.4f" %
      (B, dt * 1e3, B / dt / 1e6, fl / dt / 1e12, float(loss.item())))
