import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recommendation_amd import ops
dev = torch.device('cuda:0')
M, N, K = 65536, 1024, 1024
X = (torch.rand((M, K), device=dev) * 2 - 1).to(torch.bfloat16)
W = (torch.rand((N, K), device=dev) * 2 - 1).to(torch.bfloat16)
out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
bias = torch.zeros(N, device=dev)
f = lambda: ops.gemm_bf16(X, W, ops.MI_EPI_BIAS_ACT, flags=ops.MI_EPI_RELU, out=out, bias=bias)
for _ in range(3): f()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(10): f()
b.record(); torch.cuda.synchronize()
t = a.elapsed_time(b) / 10 * 1e-3
print("big=%s ablate=%s  %.1f us %.0f TF" % (os.environ.get('MI_GEMM_BIG'), os.environ.get('MI_GEMM_ABLATE'), t * 1e6, 2 * M * N * K / t / 1e12))
