import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import torch
from recommendation_amd import ops
dev = torch.device('cuda:0')
B, L = 32, 100
rows, D, H, H2 = 3000, 64, 128, 64
g = torch.Generator(device='cpu').manual_seed(23 + L)
f32 = dict(dtype=torch.float32, device=dev); bf = dict(dtype=torch.bfloat16, device=dev)
def rnd(*shape, scale=0.1):
    return (torch.randn(shape, generator=g) * scale).to(dev)
E_ = rnd(rows, D, scale=0.3)
W1f, Wq, b1 = rnd(H, 2 * D).to(torch.bfloat16), rnd(H, D), rnd(H)
W2, b2 = rnd(H2, H).to(torch.bfloat16), rnd(H2)
w3 = torch.zeros((H2 + 4,), **f32); w3[:H2 + 1] = rnd(H2 + 1, scale=0.2)
W2T, W1fT = W2.t().contiguous(), W1f.t().contiguous()
hist = torch.randint(0, rows, (B, L), generator=g); hist[0, 0] = -1
qid = torch.randint(0, rows, (B,), generator=g)
lengths = torch.randint(0, L + 1, (B,), generator=g).to(torch.int32); lengths[0] = L; lengths[1] = 0
hist, qid, lengths = hist.to(dev), qid.to(dev), lengths.to(dev)
M = B * L
A, H1t, H2t = torch.empty((M, 2 * D), **bf), torch.empty((M, H), **bf), torch.empty((M, H2), **bf)
a, ids_all = torch.empty((M,), **f32), torch.empty((B, L + 1), dtype=torch.int32, device=dev)
feat = torch.empty((B, 2 * D), **f32)
ops.din_lau_fwd_fused(E_, hist, qid, lengths, W1f, Wq, b1, W2, b2, w3, A, H1t, H2t, a, ids_all, feat, torch.empty((B, H), **f32))
dfeat = rnd(B, 2 * D, scale=1.0)
P = ops.din_lau_bwd_part_floats()
x = dict(dZ2=torch.full((M, H2), 7.0, **bf), dZ1=torch.full((M, H), 7.0, **bf), dE=torch.full((B * (L + 1), D), 7.0, **f32),
         S=torch.empty((B, H), **f32), part=torch.empty((B, P), **f32), sums=torch.empty((8, P), **f32))
ops.din_lau_bwd_fused(A, H1t, H2t, a, ids_all, feat, dfeat, w3, W2T, W1fT, x['dZ2'], x['dZ1'], x['dE'], x['S'], x['part'], x['sums'])
da, dA = torch.empty((M,), **f32), torch.empty((M, 2 * D), **bf)
y = dict(dZ2=torch.empty((M, H2), **bf), dZ1=torch.empty((M, H), **bf), dE=torch.empty((B * (L + 1), D), **f32), S=torch.empty((B, H), **f32))
vs = max(1, min(1024, M // 64)); part = torch.zeros((vs, H2 + 4), **f32)
ops.din_lau_pool_bwd(A, dfeat, ids_all, D, da)
ops.head_bwd(da, w3[:H2], H2t, y['dZ2'], vs, part)
ops.gemm_bf16(y['dZ2'], W2T, ops.MI_EPI_MASK, out=y['dZ1'], in0=H1t)
ops.gemm_bf16(y['dZ1'], W1fT, ops.MI_EPI_MASK, out=dA)
ops.din_lau_scatter_fact(A, dA, a, dfeat, dfeat[:, D:], feat[:, D:], ids_all, D, y['dE'])
torch.cuda.synchronize()
for k in ('dZ2', 'dZ1'):
    d = (x[k].float() - y[k].float()).abs()
    print(k, 'equal', torch.equal(x[k], y[k]), 'n diff', int((d > 0).sum()), 'max', float(d.max()), 'scale', float(y[k].float().abs().max()))
xe, ye = x['dE'].view(B, L + 1, D), y['dE'].view(B, L + 1, D)
d = (xe[:, :L] - ye[:, :L]).abs()
print('dE hist equal', torch.equal(xe[:, :L], ye[:, :L]), 'n diff', int((d > 0).sum()), 'of', d.numel(), 'max', float(d.max()), 'scale', float(ye.abs().max()))
nz = (d > 0).nonzero()[:5]
print(nz)
# compare da: recompute in torch
