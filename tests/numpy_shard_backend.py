"""CPU stand-in for one rank's local compute in the sharded FM, written on the numpy oracle.
TEST INFRASTRUCTURE ONLY: lets the exchange logic of recommendation_amd.sharded run under gloo
without a GPU.  The product backend is HipBackend."""
import numpy as np
import torch

from oracle import embedding as oemb
from oracle import fm as ofm


def _np(t):
    return t.numpy()


class NumpyBackend(object):
    def __init__(self, rows_per_table, dim, n_dense, opt, rank, world):
        self.device = torch.device('cpu')
        self.T, self.dim, self.n_dense, self.opt = len(rows_per_table), dim, n_dense, opt
        self.rank, self.world = rank, world
        self.toffs = np.concatenate([[0], np.cumsum(rows_per_table)]).astype(np.int64)
        total = int(self.toffs[-1])
        self.local_rows = (total - rank + world - 1) // world
        self.V = np.zeros((self.local_rows, dim), np.float32)
        self.w1 = np.zeros((self.local_rows,), np.float32)
        self.aV = np.zeros((self.local_rows,), np.float32)        # row-wise Adagrad accumulators
        self.aw = np.zeros((self.local_rows,), np.float32)
        self.bad = False

    def init_rows(self, V, w1):
        self.V[:], self.w1[:] = _np(V), _np(w1)

    def rows(self):
        return torch.from_numpy(self.V), torch.from_numpy(self.w1)

    def zeros(self, shape, dtype):
        return torch.zeros(shape, dtype=dtype)

    def bwd_scratch(self, n_slots):
        return torch.zeros((1,), dtype=torch.float32)

    def row_floats(self):
        return -(-(self.dim + 2 + self.n_dense) // 4) * 4      # this stand-in reduces the dense side from the rows

    def extra_rows(self, B, ld):
        return 0

    def first_local(self, t):
        return (int(self.toffs[t]) - self.rank + self.world - 1) // self.world

    def _split(self, block, B):
        """block [blk] -> (cnt [n_chunks][T], slots [T][B])"""
        nc = -(-B // 64)
        return block[:nc * self.T].reshape(nc, self.T), block[nc * self.T:].reshape(self.T, B)

    def route(self, ids, send):
        ids = _np(ids).astype(np.int64)
        B = ids.shape[0]
        g = ids + self.toffs[:-1][None, :]                      # [B][T]
        out = _np(send)                                          # [N][blk]
        for d in range(self.world):
            cnt, slots = self._split(out[d], B)
            mine = (g % self.world) == d
            slots[:] = np.where(mine, g // self.world, -1).T
            for c in range(cnt.shape[0]):
                cnt[c] = mine[64 * c:64 * c + 64].sum(0)

    def owner_forward(self, recv, B, row_stride, partial, sorted_rows, perm, n_total, sched, lr, beta1, beta2):
        assert row_stride == B
        R, P = _np(recv), _np(partial)
        slots_all = np.stack([self._split(R[s], B)[1] for s in range(self.world)])   # [N][T][B]
        D = self.dim
        P[:] = 0
        for s in range(self.world):
            for b in range(B):
                seg = slots_all[s, :, b]
                seg = seg[seg >= 0]
                if seg.size == 0:
                    continue
                v = self.V[seg]
                P[s, b, :D] = v.sum(0, dtype=np.float32)
                P[s, b, D] = (v * v).sum(dtype=np.float32)
                P[s, b, D + 1] = self.w1[seg].sum(dtype=np.float32)
        so, pe = _np(sorted_rows), _np(perm)
        at = 0
        for t in range(self.T):
            slots = slots_all[:, t, :].reshape(-1)               # stream order e = s*B + b
            e = np.nonzero(slots >= 0)[0]
            order = np.argsort(slots[e], kind='stable')
            n = e.size
            assert np.all(slots[e] >= self.first_local(t)) and np.all(slots[e] < self.first_local(t + 1))
            assert n == sum(int(self._split(R[s], B)[0][:, t].sum()) for s in range(self.world))
            so[at:at + n], pe[at:at + n] = slots[e][order], e[order]
            at += n
        _np(n_total)[0] = at
        _np(sched)[3] += 1

    def combine(self, partials, B, dense, wd, bias, labels, gscale, X):
        P = _np(partials)
        D = self.dim
        tot = P.sum(0, dtype=np.float32)                         # ranks in order
        Sn = tot[:, :D]
        z = 0.5 * ((Sn * Sn).sum(1) - tot[:, D]) + tot[:, D + 1] + float(_np(bias)[0])
        if dense is not None:
            z = z + _np(dense) @ _np(wd)
        z = z.astype(np.float32)
        y = _np(labels).astype(np.float32)
        per, sig = ofm.bce_with_logits(z, y)
        x = _np(X)
        x[:] = 0
        x[:, :D] = Sn
        x[:, D] = (sig - y) * np.float32(gscale)
        x[:, D + 1] = per
        if dense is not None:
            x[:, D + 2:D + 2 + self.n_dense] = _np(dense)

    def update(self, sorted_rows, perm, n_total, X_all, B, row_stride, dp, dm, dv, lr, beta1, beta2, eps, sched,
               scratch, loss):
        x = _np(X_all)
        D, nd = self.dim, self.n_dense
        g = x[:, D]
        grad = np.concatenate([g @ x[:, D + 2:D + 2 + nd], [g.sum(dtype=np.float32)]]).astype(np.float32)
        _np(loss)[0] = x[:, D + 1].mean(dtype=np.float32)
        alpha, omb1, omb2 = oemb.tf_adam_scalars(lr, beta1, beta2, int(_np(sched)[3]))
        w, m, v = oemb.adam_tf_update(_np(dp).copy(), _np(dm).copy(), _np(dv).copy(), grad, alpha, omb1, omb2, eps)
        _np(dp)[:], _np(dm)[:], _np(dv)[:] = w, m, v
        # sparse side: runs of equal rows in the sorted entries, padding skipped
        n = int(_np(n_total)[0])
        ga, Sa = x[:, D], x[:, :D]
        rows, es = _np(sorted_rows)[:n].astype(np.int64), _np(perm)[:n]
        uniq, start = np.unique(rows, return_index=True)
        assert np.all(np.diff(rows) >= 0)
        end = np.append(start[1:], rows.size)
        for u, lo, hi in zip(uniq, start, end):
            e = es[lo:hi]
            gv = (ga[e, None] * (Sa[e] - self.V[u][None, :])).sum(0, dtype=np.float32)
            gw = ga[e].sum(dtype=np.float32)
            if self.opt == 'sgd':
                self.V[u] -= np.float32(lr) * gv
                self.w1[u] -= np.float32(lr) * gw
            elif self.opt == 'adagrad_rowwise':
                self.aV[u] += (gv * gv).mean(dtype=np.float32)
                self.V[u] -= np.float32(lr) * gv / (np.sqrt(self.aV[u]) + np.float32(eps))
                self.aw[u] += gw * gw
                self.w1[u] -= np.float32(lr) * gw / (np.sqrt(self.aw[u]) + np.float32(eps))
            else:
                raise NotImplementedError(self.opt)
