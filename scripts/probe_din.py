"""Time one DIN training step at the benchmark shape (1 M items, dim 64, L = 100, B = 8192)."""
import os
import sys
import time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recommendation_amd.din import DIN

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
L, ROWS = 100, 1000000
dev = torch.device('cuda:0')
m = DIN(ROWS, dev, max_len=L)
g = torch.Generator(device='cpu').manual_seed(0)
hist = torch.randint(0, ROWS, (B, L), generator=g).to(dev)
qid = torch.randint(0, ROWS, (B,), generator=g).to(dev)
lens = torch.randint(1, L + 1, (B,), generator=g).to(torch.int32).to(dev)
labels = (torch.rand((B,), generator=g) < 0.3).float().to(dev)
for _ in range(3):
    loss = m.train_step(hist, qid, lens, labels)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 10
for _ in range(n):
    loss = m.train_step(hist, qid, lens, labels)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print("B=%d step %.3f ms  %.2f M samples/s  %.0f TFLOP/s (contractions, padded widths)  loss %.4f" %
      (B, dt * 1e3, B / dt / 1e6, m.flops_per_sample() * B / dt / 1e12, float(loss.item())))
