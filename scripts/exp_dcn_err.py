import sys
import numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from oracle import dcn as odcn, fm as ofm
from recommendation_amd.dcn import DCNv2
import test_dcn_gpu as T
dev = torch.device('cuda:0')
B, mlp = 512, (256, 128)
st = T._state(3, mlp)
m = DCNv2(T.ROWS, T.DIM, T.ND, dev, mlp=mlp, keep_grads=True)
m.load_state(st)
ids, dense, labels = ofm.synthetic_batch(B, T.ROWS, T.ND, seed=4, dist='zipf')
loss_want, gr, uniq, dV, k = odcn.grads(st, ids, dense, labels, storage='bf16')
m.train_step(torch.from_numpy(ids).to(dev), torch.from_numpy(dense).to(dev), torch.from_numpy(labels).to(dev))
d = m.d
for name, p in m.dense_params().items():
    if name == 'out/wb': continue
    g = gr[name]; got = p.grad.cpu().numpy()
    got = got[:g.shape[0], :g.shape[1]] if g.ndim == 2 else got[:g.shape[0]]
    print(name, 'max-rel %.4f' % T.rel(got, g), 'rms-rel %.4f' % (np.sqrt(((got - g) ** 2).mean()) / np.sqrt((g ** 2).mean())))
print('dx0', T.rel(m.dx0.cpu().numpy()[:, :d], k['dx0']))
