#!/usr/bin/env python3
"""What the vendor library (through torch.matmul) reaches on the contraction shapes of a DCN-v2 step at batch 65536,
bf16 operands on random data: the comparator for mi_gemm_bf16.  Prints TFLOP/s per shape and operand layout."""
import torch

dev = torch.device('cuda', 0)
B = 65536


def bench(f, flops, reps=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    return ms, flops / ms / 1e9


for name, n_in, n_out in (("cross 448x448", 448, 448), ("mlp0 448->1024", 448, 1024), ("mlp1 1024->1024", 1024, 1024),
                          ("mlp2 1024->512", 1024, 512)):
    x = torch.randn(B, n_in, device=dev, dtype=torch.bfloat16)
    w = torch.randn(n_out, n_in, device=dev, dtype=torch.bfloat16) * 0.05          # [out][in]: y = x w^T
    wt = w.t().contiguous()                                                        # [in][out]
    dy = torch.randn(B, n_out, device=dev, dtype=torch.bfloat16)
    flops = 2.0 * B * n_in * n_out
    ms, tf = bench(lambda: torch.matmul(x, w.t()), flops)
    print("%-18s forward  y = x w^T      %7.1f us %7.1f TFLOP/s" % (name, ms * 1e3, tf))
    ms, tf = bench(lambda: torch.matmul(dy, w), flops)
    print("%-18s data     dx = dy w      %7.1f us %7.1f TFLOP/s" % (name, ms * 1e3, tf))
    ms, tf = bench(lambda: torch.matmul(dy.t(), x), flops)
    print("%-18s weights  dw = dy^T x    %7.1f us %7.1f TFLOP/s" % (name, ms * 1e3, tf), flush=True)
