import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recommendation_amd import ops, _lib
dev = torch.device('cuda:0')
M = 65536
def rnd(*shape):
    return (torch.rand(shape, device=dev) * 2 - 1).to(torch.bfloat16)
lib = _lib.load()
dbg = torch.zeros(1024 * 4, dtype=torch.int64, device=dev)
lib.mi_debug_gemm.argtypes = [ctypes.c_void_p]
for K in (1024, 512):
    A, W, X = rnd(M, K), rnd(1024, K), rnd(M, 1024)
    out = torch.empty((M, 1024), dtype=torch.bfloat16, device=dev)
    bias = torch.zeros(1024, device=dev)
    for name, fn in [
        ("plain", lambda: ops.gemm_bf16(A, W, ops.MI_EPI_MASK, out=out)),
        ("bias+relu", lambda: ops.gemm_bf16(A, W, ops.MI_EPI_BIAS_ACT, flags=ops.MI_EPI_RELU, out=out, bias=bias)),
        ("mask+input", lambda: ops.gemm_bf16(A, W, ops.MI_EPI_MASK, out=out, in0=X)),
    ]:
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        lib.mi_debug_gemm(ctypes.c_void_p(dbg.data_ptr()))
        dbg.zero_()
        fn()
        torch.cuda.synchronize()
        lib.mi_debug_gemm(ctypes.c_void_p(0))
        d = dbg.cpu().numpy().reshape(1024, 4).astype(float) / 100.0
        t0 = d[:, 0].min()
        start, main, epi = d[:, 0] - t0, d[:, 1] - d[:, 0], d[:, 2] - d[:, 1]
        import numpy as np
        for lo in (0, 256, 512, 768):
            sl = slice(lo, lo + 256)
            print("K=%4d %-11s wgs %4d-%4d: start %6.1f..%6.1f  main %5.1f (%.1f..%.1f)  epilogue %5.1f (%.1f..%.1f)  end max %.1f" % (
                K, name, lo, lo + 255, start[sl].min(), start[sl].max(), main[sl].mean(), main[sl].min(), main[sl].max(),
                epi[sl].mean(), epi[sl].min(), epi[sl].max(), (d[sl, 2] - t0).max()))
