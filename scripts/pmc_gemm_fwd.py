#!/usr/bin/env python3
"""Only the roofline kernel of `bench.py --workload dcn`: the 1024x1024 forward contraction at M=65536
with its bias+ReLU epilogue, 20 launches - the target of the rocprofv3 --pmc passes."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch

from recommendation_amd import ops

dev = torch.device('cuda', 0)
M, N, K = 65536, 1024, 1024
X = (torch.rand((M, K), device=dev) * 2 - 1).to(torch.bfloat16)
W = (torch.rand((N, K), device=dev) * 2 - 1).to(torch.bfloat16)
out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
bias = torch.zeros(N, device=dev)
for _ in range(20):
    ops.gemm_bf16(X, W, ops.MI_EPI_BIAS_ACT, flags=ops.MI_EPI_RELU, out=out, bias=bias)
torch.cuda.synchronize()
