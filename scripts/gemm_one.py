"""One shape of the forward contraction, a few launches (for counter passes and traces).
usage: gemm_one.py M N K [reps]"""
import sys
import os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recommendation_amd import ops

M, N, K = (int(x) for x in sys.argv[1:4])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
dev = torch.device('cuda:0')
X = (torch.rand((M, K), device=dev) * 2 - 1).to(torch.bfloat16)
W = (torch.rand((N, K), device=dev) * 2 - 1).to(torch.bfloat16)
out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
bias = torch.zeros(N, device=dev)
for _ in range(reps):
    ops.gemm_bf16(X, W, ops.MI_EPI_BIAS_ACT, flags=ops.MI_EPI_RELU, out=out, bias=bias)
torch.cuda.synchronize()
print("done")
