"""Time the bf16 MFMA contraction on the shapes a DCN-v2 step uses (B = 65536)."""
import sys
import torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recommendation_amd import ops

dev = torch.device('cuda:0')
Bsz = 65536


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


for (N, K) in [(1024, 1024), (448, 448), (1024, 448), (512, 1024)]:
    X, W = rnd(Bsz, K), rnd(N, K)
    out = torch.empty((Bsz, N), dtype=torch.bfloat16, device=dev)
    bias = torch.zeros(N, device=dev)
    t = timeit(lambda: ops.gemm_bf16(X, W, ops.MI_EPI_BIAS_ACT, flags=ops.MI_EPI_RELU, out=out, bias=bias))
    print("fwd  M=%d N=%d K=%d  %.1f us  %.0f TFLOP/s" % (Bsz, N, K, t * 1e6, 2 * Bsz * N * K / t / 1e12))
    ref = timeit(lambda: torch.matmul(X, W.t()))
    print("     torch.matmul (hipBLASLt)   %.1f us  %.0f TFLOP/s" % (ref * 1e6, 2 * Bsz * N * K / ref / 1e12))
    dY = rnd(Bsz, N)
    dX = torch.empty((Bsz, K), dtype=torch.bfloat16, device=dev)
    t = timeit(lambda: ops.gemm_bf16(dY, W, ops.MI_EPI_MASK, b_rm=True, out=dX, in0=X))
    print("dX   M=%d N=%d K=%d  %.1f us  %.0f TFLOP/s" % (Bsz, K, N, t * 1e6, 2 * Bsz * N * K / t / 1e12))
    for splits in (8, 16, 32, 64):
        slab = torch.empty((splits, N, K), dtype=torch.float32, device=dev)
        t = timeit(lambda: ops.gemm_bf16(dY, X, ops.MI_EPI_SLAB, a_rm=True, b_rm=True, k_splits=splits, out=slab))
        print("dW   M=%d N=%d K=%d splits=%d  %.1f us  %.0f TFLOP/s" % (N, K, Bsz, splits, t * 1e6, 2 * Bsz * N * K / t / 1e12))
