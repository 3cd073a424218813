"""CPU stand-in for one rank's local compute in the sharded embedding bag (numpy).
TEST INFRASTRUCTURE ONLY: lets the exchange logic of recommendation_amd.sharded_bag run under gloo
without a GPU and serves as the checker of the HIP kernels' index work.  The product backend is
HipBagBackend."""
import numpy as np
import torch


def slots(toffs, T, ids, L, n_bags, n_parts, owner, N, cap):
    """numpy statement of mi_shard_bag_slots: (local_ids, row_of_bag, bag_of_row, cnt, overflow)."""
    ids = np.asarray(ids).reshape(-1, n_bags, L).astype(np.int64)
    t = np.arange(n_bags) % T
    base = np.asarray(toffs)[t][None, :, None]
    rows = (np.asarray(toffs)[t + 1] - np.asarray(toffs)[t])[None, :, None]
    local_ids = np.full((n_parts, n_bags, L), -1, np.int32)
    row_of_bag = np.full((n_parts, n_bags), -1, np.int32)
    bag_of_row = np.full((n_parts * cap,), -1, np.int32)
    cnt = np.zeros(n_parts, np.int32)
    overflow = False
    for p in range(n_parts):
        o = owner if owner >= 0 else p
        part = ids[p if owner >= 0 else 0][None][0]
        valid = (part >= 0) & (part < rows[0])
        g = base[0] + np.where(valid, part, 0)
        own = valid & (g % N == o)
        lstart = (np.asarray(toffs)[t] + N - 1 - o) // N
        local_ids[p] = np.where(own, g // N - lstart[:, None], -1)
        f = own.any(axis=1)
        slot = np.cumsum(f) - 1
        ok = f & (slot < cap)
        overflow |= bool((f & ~ok).any())
        row_of_bag[p][ok] = p * cap + slot[ok]
        bag_of_row[p * cap + slot[ok]] = np.nonzero(ok)[0]
        cnt[p] = min(int(f.sum()), cap)
    return local_ids, row_of_bag, bag_of_row, cnt, overflow


def bucket(toffs, T, ids, L, n_bags, rank, N, cap, cap_ids, row_of_bag):
    """numpy statement of mi_shard_bag_bucket: (send_ids [N][cap_ids], send_off [N][cap], overflow)."""
    ids = np.asarray(ids).reshape(n_bags, L).astype(np.int64)
    toffs = np.asarray(toffs)
    t = np.arange(n_bags) % T
    base, rows = toffs[t][:, None], (toffs[t + 1] - toffs[t])[:, None]
    valid = (ids >= 0) & (ids < rows)
    g = base + np.where(valid, ids, 0)
    send_ids = np.full((N, cap_ids), -1, np.int32)
    send_off = np.zeros((N, cap), np.int64)
    overflow = False
    for p in range(N):
        own = valid & (g % N == p)
        pos, slots = 0, 0
        for bag in range(n_bags):
            r = int(row_of_bag[p][bag])
            if r < 0:
                continue
            mine = (g[bag][own[bag]] // N).astype(np.int32)
            send_off[p][r - p * cap] = rank * cap_ids + min(pos, cap_ids)
            if pos + mine.size <= cap_ids:
                send_ids[p][pos:pos + mine.size] = mine
            else:
                overflow = True
            pos += mine.size
            slots += 1
        send_off[p][slots:] = rank * cap_ids + min(pos, cap_ids)
    return send_ids, send_off, overflow


class NumpyBagBackend(object):
    def __init__(self, rows_per_table, dim, bag_len, opt, rank, world):
        self.device = torch.device('cpu')
        self.T, self.dim, self.L, self.opt = len(rows_per_table), dim, bag_len, opt
        self.rank, self.world = rank, world
        self.toffs = np.concatenate([[0], np.cumsum(rows_per_table)]).astype(np.int64)
        total = int(self.toffs[-1])
        self.local_rows = (total - rank + world - 1) // world
        self.ltoffs = (self.toffs + world - 1 - rank) // world
        self.W = torch.zeros((max(self.local_rows, 1), dim), dtype=torch.float32)
        self.state = torch.zeros((max(self.local_rows, 1),), dtype=torch.float32)
        self.overflow = False

    def zeros(self, shape, dtype):
        return torch.zeros(shape, dtype=dtype)

    def slots_workspace(self, n_bags, n_parts):
        return torch.zeros((1,), dtype=torch.uint8)

    def _slots(self, ids, n_bags, n_parts, owner, cap, local_ids, row_of_bag, bag_of_row, cnt):
        li, rb, br, c, ov = slots(self.toffs, self.T, ids.numpy(), self.L, n_bags, n_parts, owner, self.world, cap)
        self.overflow |= ov
        if local_ids is not None:
            local_ids.copy_(torch.from_numpy(li.reshape(-1, self.L)))
        row_of_bag.copy_(torch.from_numpy(rb))
        bag_of_row.copy_(torch.from_numpy(br))
        cnt.copy_(torch.from_numpy(c))

    def owner_slots(self, ids_all, n_bags, cap, local_ids, row_of_bag, bag_of_row, cnt, ws):
        self._slots(ids_all, n_bags, self.world, self.rank, cap, local_ids, row_of_bag, bag_of_row, cnt)

    def requester_slots(self, ids, n_bags, cap, row_of_bag, bag_of_row, cnt, ws):
        self._slots(ids, n_bags, self.world, -1, cap, None, row_of_bag, bag_of_row, cnt)

    def bucket(self, ids, n_bags, cap, cap_ids, row_of_bag, cnt, send_ids, send_off, ws):
        si, so, ov = bucket(self.toffs, self.T, ids.numpy(), self.L, n_bags, self.rank, self.world, cap, cap_ids,
                            row_of_bag.numpy())
        self.overflow |= ov
        send_ids.copy_(torch.from_numpy(si))
        send_off.copy_(torch.from_numpy(so))

    def gather_csr(self, recv_ids, recv_off, out_rows):
        W, ids, off = self.W.numpy(), recv_ids.numpy(), recv_off.numpy()
        out = out_rows.numpy().reshape(-1, self.dim)
        for i in range(off.size - 1):
            acc = np.zeros(self.dim, np.float32)
            for k in ids[off[i]:off[i + 1]]:
                if k >= 0:
                    acc = acc + W[k]
            out[i] = acc

    def update_csr(self, recv_ids, recv_off, grad_rows, lr, eps):
        ids, off = recv_ids.numpy(), recv_off.numpy()
        g = grad_rows.numpy().reshape(-1, self.dim)
        W, st = self.W.numpy(), self.state.numpy()
        bag = np.repeat(np.arange(off.size - 1), np.diff(off))
        live = ids[off[0]:off[-1]] >= 0
        rows, src = ids[off[0]:off[-1]][live], bag[live]
        order = np.argsort(rows, kind='stable')
        rows, src = rows[order], src[order]
        uniq, start = np.unique(rows, return_index=True)
        end = np.append(start[1:], rows.size)
        for r, lo, hi in zip(uniq, start, end):
            s = np.zeros(self.dim, np.float64)
            for k in src[lo:hi]:
                s += g[k]
            if self.opt == 'sgd':
                W[r] = (W[r] - lr * s).astype(np.float32)
            else:
                st[r] = np.float32(st[r] + np.mean(s * s))
                W[r] = (W[r] - lr * s / (np.sqrt(st[r]) + eps)).astype(np.float32)

    def gather(self, local_ids, n_bags, cap, bag_of_row, cnt, out_rows):
        W, li = self.W.numpy(), local_ids.numpy().reshape(self.world, n_bags, self.L)
        out = out_rows.numpy()
        for p in range(self.world):
            for slot in range(int(cnt[p])):
                bag = int(bag_of_row[p * cap + slot])
                acc = np.zeros(self.dim, np.float32)
                for l in range(self.L):
                    if li[p, bag, l] >= 0:
                        acc = acc + W[self.ltoffs[bag % self.T] + li[p, bag, l]]
                out[p, slot] = acc

    def combine(self, rows, row_of_bag, n_bags, out):
        flat = rows.numpy().reshape(-1, self.dim)
        rb = row_of_bag.numpy()
        o = np.zeros((n_bags, self.dim), np.float32)
        for p in range(self.world):
            has = rb[p] >= 0
            o[has] = o[has] + flat[rb[p][has]]
        out.copy_(torch.from_numpy(o.reshape(-1, self.T * self.dim)))

    def pack(self, grad, bag_of_row, cnt, cap, rows_out):
        g = grad.numpy().reshape(-1, self.dim)                       # [n_bags][D]
        out = rows_out.numpy()
        for p in range(self.world):
            n = int(cnt[p])
            out[p, :n] = g[bag_of_row.numpy()[p * cap:p * cap + n]]

    def update(self, local_ids, grad_rows, row_of_bag, lr, eps):
        li = local_ids.numpy()                                        # [N*n_bags][L]
        rb = row_of_bag.numpy().reshape(-1)
        g = grad_rows.numpy().reshape(-1, self.dim)
        W, st = self.W.numpy(), self.state.numpy()
        bag = np.repeat(np.arange(li.shape[0]), self.L)
        flat = li.reshape(-1)
        live = flat >= 0
        rows = self.ltoffs[bag[live] % self.T] + flat[live]
        src = rb[bag[live]]
        order = np.argsort(rows, kind='stable')
        rows, src = rows[order], src[order]
        uniq, start = np.unique(rows, return_index=True)
        end = np.append(start[1:], rows.size)
        for r, lo, hi in zip(uniq, start, end):
            s = np.zeros(self.dim, np.float64)
            for k in src[lo:hi]:
                s += g[k]
            if self.opt == 'sgd':
                W[r] = (W[r] - lr * s).astype(np.float32)
            else:
                st[r] = np.float32(st[r] + np.mean(s * s))
                W[r] = (W[r] - lr * s / (np.sqrt(st[r]) + eps)).astype(np.float32)

    def check(self):
        if self.overflow:
            raise ValueError("capacity exceeded")
