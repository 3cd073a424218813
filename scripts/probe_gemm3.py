import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recommendation_amd import ops
dev = torch.device('cuda:0')
M = 65536
def timeit(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e-3
rnd = lambda *s: (torch.rand(s, device=dev) * 2 - 1).to(torch.bfloat16)
for (N, K) in [(1024, 1024), (512, 1024), (1024, 512)]:      # W [N, K]; dX = dY[M,N] * W -> [M,K]
    W, dY, X = rnd(N, K), rnd(M, N), rnd(M, K)
    dX = torch.empty((M, K), dtype=torch.bfloat16, device=dev)
    Wt = W.t().contiguous()                                     # [K, N]: the same product with a row-stored operand
    fl = 2 * M * N * K
    t = timeit(lambda: ops.gemm_bf16(dY, W, ops.MI_EPI_MASK, b_rm=True, out=dX, in0=X))
    print("dX tr-B  mask   red=%d out=%d %.1f us %.0f TF" % (N, K, t * 1e6, fl / t / 1e12))
    t = timeit(lambda: ops.gemm_bf16(dY, W, ops.MI_EPI_MASK, b_rm=True, out=dX))
    print("dX tr-B  plain  red=%d out=%d %.1f us %.0f TF" % (N, K, t * 1e6, fl / t / 1e12))
    t = timeit(lambda: ops.gemm_bf16(dY, Wt, ops.MI_EPI_MASK, out=dX, in0=X))
    print("dX row-B mask   red=%d out=%d %.1f us %.0f TF" % (N, K, t * 1e6, fl / t / 1e12))
    t = timeit(lambda: ops.gemm_bf16(dY, Wt, ops.MI_EPI_MASK, out=dX))
    print("dX row-B plain  red=%d out=%d %.1f us %.0f TF" % (N, K, t * 1e6, fl / t / 1e12))
