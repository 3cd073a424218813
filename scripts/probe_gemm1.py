"""One contraction shape, a few launches: for counter collection."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recommendation_amd import ops
dev = torch.device('cuda:0')
M, N, K = 65536, 1024, 1024
mode = sys.argv[1] if len(sys.argv) > 1 else 'fwd'
X = (torch.rand((M, K), device=dev) * 2 - 1).to(torch.bfloat16)
W = (torch.rand((N, K), device=dev) * 2 - 1).to(torch.bfloat16)
out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
bias = torch.zeros(N, device=dev)
for _ in range(5):
    if mode == 'fwd':
        ops.gemm_bf16(X, W, ops.MI_EPI_BIAS_ACT, flags=ops.MI_EPI_RELU, out=out, bias=bias)
    else:
        ops.gemm_bf16(out, W, ops.MI_EPI_MASK, b_rm=True, out=X)
torch.cuda.synchronize()
