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

    def route(self, ids, send):
        ids = _np(ids).astype(np.int64)
        B = ids.shape[0]
        g = ids + self.toffs[:-1][None, :]
        out = _np(send)
        for d in range(self.world):
            mask = (g % self.world) == d
            out[d, 0] = 0
            out[d, 1:B + 1] = np.cumsum(mask.sum(1))
            loc = (g // self.world)[mask]                        # row-major: sample, then field
            out[d, B + 1:B + 1 + loc.size] = loc

    def partial_fwd(self, recv, B, partial, rows_out, entry_sample, n_dev):
        R, P = _np(recv), _np(partial)
        ro, es = _np(rows_out), _np(entry_sample)
        P[:] = 0
        D = self.dim
        base = 0
        for s in range(self.world):
            off, rows = R[s, :B + 1], R[s, B + 1:]
            for b in range(B):
                lo, hi = off[b], off[b + 1]
                if hi == lo:
                    continue
                seg = rows[lo:hi]
                v = self.V[seg]
                P[s, b, :D] = v.sum(0, dtype=np.float32)
                P[s, b, D] = (v * v).sum(dtype=np.float32)
                P[s, b, D + 1] = self.w1[seg].sum(dtype=np.float32)
                ro[base + lo:base + hi] = seg
                es[base + lo:base + hi] = s * B + b
            base += off[B]
        _np(n_dev)[0] = base
        self._n = int(base)

    def received(self):
        return self._n

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

    def dense_step(self, X_all, dp, dm, dv, lr, beta1, beta2, eps, sched, loss):
        x = _np(X_all)
        D, nd = self.dim, self.n_dense
        g = x[:, D]
        grad = np.concatenate([g @ x[:, D + 2:D + 2 + nd], [g.sum(dtype=np.float32)]]).astype(np.float32)
        _np(loss)[0] = x[:, D + 1].mean(dtype=np.float32)
        s = _np(sched)
        s[3] += 1
        alpha, omb1, omb2 = oemb.tf_adam_scalars(lr, beta1, beta2, int(s[3]))
        w, m, v = oemb.adam_tf_update(_np(dp).copy(), _np(dm).copy(), _np(dv).copy(), grad, alpha, omb1, omb2, eps)
        _np(dp)[:], _np(dm)[:], _np(dv)[:] = w, m, v

    def owner_backward(self, rows, n_recv, entry_sample, X_all, lr, beta1, beta2, eps, sched):
        if n_recv == 0:
            return
        rows, es = _np(rows)[:n_recv].astype(np.int64), _np(entry_sample)[:n_recv]
        x = _np(X_all)
        ga, Sa = x[:, self.dim], x[:, :self.dim]
        order = np.argsort(rows, kind='stable')
        uniq, start = np.unique(rows[order], return_index=True)
        end = np.append(start[1:], n_recv)
        for u, lo, hi in zip(uniq, start, end):
            e = es[order[lo:hi]]
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
