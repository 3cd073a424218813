"""Data-gradient contractions of a DCN-v2 step (row-stored transposed weights) with their epilogues."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recommendation_amd import ops

dev = torch.device('cuda:0')
M = 65536


def timeit(fn, reps=10):
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


for (n_out, k_in, mode) in [(512, 1024, 'mask'), (1024, 1024, 'mask'), (1024, 448, 'cross'), (448, 448, 'cross')]:
    dY, WT, X = rnd(M, n_out), rnd(k_in, n_out), rnd(M, k_in)
    dX = torch.empty((M, k_in), dtype=torch.bfloat16, device=dev)
    if mode == 'mask':
        fn = lambda: ops.gemm_bf16(dY, WT, ops.MI_EPI_MASK, out=dX, in0=X)
    else:
        du, nxt = torch.empty_like(dX), rnd(M, k_in)
        fn = lambda: ops.gemm_bf16(dY, WT, ops.MI_EPI_CROSS_BWD, out=dX, in1=nxt, out1=du, in0=X)
    t = timeit(fn)
    print("dX %-5s out %4d <- %4d  %.1f us  %.0f TFLOP/s" % (mode, k_in, n_out, t * 1e6, 2 * M * n_out * k_in / t / 1e12))
    fw = lambda: ops.gemm_bf16(X, rnd(n_out, k_in), ops.MI_EPI_BIAS_ACT, flags=ops.MI_EPI_RELU, out=dY, bias=torch.zeros(n_out, device=dev))

print("-- epilogue cost on M=65536, N=1024, K=1024 and K=512")
for K in (1024, 512):
    A, W, X = rnd(M, K), rnd(1024, K), rnd(M, 1024)
    out = torch.empty((M, 1024), dtype=torch.bfloat16, device=dev)
    out1 = torch.empty_like(out)
    bias = torch.zeros(1024, device=dev)
    fl = 2 * M * 1024 * K
    for name, fn in [
        ("plain store (mask, no input)", lambda: ops.gemm_bf16(A, W, ops.MI_EPI_MASK, out=out)),
        ("bias + relu", lambda: ops.gemm_bf16(A, W, ops.MI_EPI_BIAS_ACT, flags=ops.MI_EPI_RELU, out=out, bias=bias)),
        ("mask with input tile", lambda: ops.gemm_bf16(A, W, ops.MI_EPI_MASK, out=out, in0=X)),
        ("cross bwd: 2 inputs, 2 outputs", lambda: ops.gemm_bf16(A, W, ops.MI_EPI_CROSS_BWD, out=out, in1=X, out1=out1, in0=X)),
    ]:
        t = timeit(fn)
        print("K=%4d %-34s %.1f us  %.0f TFLOP/s" % (K, name, t * 1e6, fl / t / 1e12))
