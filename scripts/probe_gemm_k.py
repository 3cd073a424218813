"""Where a 256x256 tile's time goes: the forward contraction over K and over tiles per CU."""
import sys
import os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recommendation_amd import ops

dev = torch.device('cuda:0')


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e-3


def rnd(*shape):
    return (torch.rand(shape, device=dev) * 2 - 1).to(torch.bfloat16)


N = 1024
for M in (16384, 32768, 65536, 131072):
    for K in (128, 256, 512, 1024, 2048, 4096):
        X, W = rnd(M, K), rnd(N, K)
        out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
        bias = torch.zeros(N, device=dev)
        t = timeit(lambda: ops.gemm_bf16(X, W, ops.MI_EPI_BIAS_ACT, flags=ops.MI_EPI_RELU, out=out, bias=bias))
        ref = timeit(lambda: torch.matmul(X, W.t()))
        tiles_per_cu = M // 256 * (N // 256) / 256
        print("M=%6d K=%4d tiles/CU %.0f  ours %7.1f us %5.0f TF | vendor %7.1f us %5.0f TF | per tile %.2f us, per k-tile %.3f us"
              % (M, K, tiles_per_cu, t * 1e6, 2 * M * N * K / t / 1e12, ref * 1e6, 2 * M * N * K / ref / 1e12,
                 t * 1e6 / tiles_per_cu, t * 1e6 / tiles_per_cu / (K / 64)), flush=True)
