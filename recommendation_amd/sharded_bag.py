"""Row-wise sharded embedding bags across the GPUs of one node (one process per GPU): the generic
sharded table for models that consume pooled rows (embedding bags of any width, DCN-v2, DIN).

The reference distributes with TensorFlow's asynchronous parameter server (2 workers + 1 ps, every
variable read and optimizer apply a gRPC hop: recommendation/code/model_trainer.py:124-136,
recommendation/train_job.py:41-42).  That scheme is not reproduced.  Here global row
r = table_offsets[t] + id lives on rank r % N as local row r / N, and a synchronous step is

  forward   A. all-gather of the ids            [N][B][T][L]      (8 B an id; rows are 512 B)
            owner: slots (which bags of which rank I hold ids of) + partial pools -> blocks
            B. all-to-all of the blocks         [N][cap][D] fp32  (one-id bags: B*T rows in all)
            requester: pooled[b][t] = sum of the owners' rows, rank order
  backward  requester: the gradient row of each bag -> blocks in the same compact order
            C. all-to-all of the blocks
            owner: sort-by-row + fused optimizer on its local rows (gradient rows read in place)

``id_exchange='bucketed'`` replaces A by an all-to-all of the ids bucketed by owner (the reference's workers ask
each parameter server only for the rows it holds, model_trainer.py:124-136): the requester sends owner p the CSR
form of "my bags you contribute to" - your ids as YOUR local rows, int32, packed in bag order, and the first id of
every slot - 4 bytes an id + 8 per (bag, owner) instead of 8*N bytes an id; the owner then runs the library's
ordinary CSR embedding bag (mi_embag_fwd / mi_embag_bwd_prepare with offsets) over its local rows: no owner-side
slot work at all.  ``cap_ids`` = capacity_factor * B*T*L / N ids per rank pair.

Three (bucketed: four) collectives of fixed size, nothing on the host waits for the device.  ``cap`` =
capacity_factor * B*T / N rows per rank pair (one-id bags; all B*T for bags of several ids, which
nearly every owner contributes to).  A pair that needs more raises MI_IDERR_CAPACITY in the error
word (checked by ``check()``): its surplus bags would contribute nothing, so the step is invalid.
Local compute goes through a backend object so that the exchange logic runs under gloo on CPU with
a numpy stand-in (tests); the product backend is ``HipBagBackend`` and there is no other.
"""
import torch
import torch.distributed as dist

from . import ops
from .sharded import TorchComm, local_row_count

_OPT = {'sgd': ops.MI_OPT_SGD, 'adagrad_rowwise': ops.MI_OPT_ADAGRAD_ROWWISE}


def local_table_offsets(rows_per_table, rank, world):
    """Local rows that belong to tables before t: ceil((table_offsets[t] - rank) / N), T+1 entries."""
    out, off = [], 0
    for r in list(rows_per_table) + [0]:
        out.append((off + world - 1 - rank) // world)
        off += int(r)
    return out


class HipBagBackend(object):
    """Local compute of one rank: C-ABI calls only."""

    def __init__(self, rows_per_table, dim, bag_len, device, opt, rank, world):
        self.device = torch.device(device)
        self.T, self.dim, self.L, self.opt = len(rows_per_table), int(dim), int(bag_len), opt
        self.rank, self.world = rank, world
        self.total_rows = int(sum(rows_per_table))
        self.local_rows = local_row_count(self.total_rows, rank, world)
        self.toffs = ops.table_offsets(rows_per_table, self.device)
        self.ltoffs = torch.tensor(local_table_offsets(rows_per_table, rank, world), dtype=torch.int64,
                                   device=self.device)
        self.W = torch.zeros((max(self.local_rows, 1), self.dim), dtype=torch.float32, device=self.device)
        self.state = torch.zeros((max(self.local_rows, 1),), dtype=torch.float32, device=self.device) \
            if opt == 'adagrad_rowwise' else None
        self.err = ops.new_error_word(self.device)
        self._plan = None

    def zeros(self, shape, dtype):
        return torch.zeros(shape, dtype=dtype, device=self.device)

    def slots_workspace(self, n_bags, n_parts):
        n = max(ops.shard_bag_slots_workspace_bytes(n_bags, n_parts), ops.shard_bag_bucket_workspace_bytes(n_bags, n_parts))
        return self.zeros((max(n, 1),), torch.uint8)

    # bucketed id exchange: sender side, then the owner's side as the ordinary CSR embedding bag on its local rows
    def bucket(self, ids, n_bags, cap, cap_ids, row_of_bag, cnt, send_ids, send_off, ws):
        ops.shard_bag_bucket(self.toffs, self.T, ids, self.L, n_bags, self.rank, self.world, cap, cap_ids, row_of_bag, cnt,
                             send_ids, send_off, ws, self.err)

    def _one_table(self):
        if getattr(self, '_toffs1', None) is None:
            self._toffs1 = ops.table_offsets([max(self.local_rows, 1)], self.device)
        return self._toffs1

    def gather_csr(self, recv_ids, recv_off, out_rows):
        ops.embag_fwd(self.W, self._one_table(), 1, recv_ids, offsets=recv_off, out=out_rows.view(-1, self.dim),
                      out_ld=self.dim, err=self.err)

    def update_csr(self, recv_ids, recv_off, grad_rows, lr, eps):
        if self._plan is None or self._plan.offsets is not recv_off or self._plan.n_entries != recv_ids.numel():
            self._plan = ops.SortedPlan(self._one_table(), 1, max(self.local_rows, 1), recv_ids, offsets=recv_off,
                                        runs=False)
        else:
            self._plan.prepare(recv_ids)
        g = grad_rows.view(-1, self.dim)
        if self.opt == 'sgd':
            ops.bwd_apply_sgd(self._plan, g, self.W, lr)
        else:
            ops.bwd_apply_adagrad_rowwise(self._plan, g, self.W, self.state, lr, eps)

    def owner_slots(self, ids_all, n_bags, cap, local_ids, row_of_bag, bag_of_row, cnt, ws):
        ops.shard_bag_slots(self.toffs, self.T, ids_all, self.L, n_bags, self.world, self.rank, self.world, cap,
                            local_ids, row_of_bag, bag_of_row, cnt, ws, self.err)

    def requester_slots(self, ids, n_bags, cap, row_of_bag, bag_of_row, cnt, ws):
        ops.shard_bag_slots(self.toffs, self.T, ids, self.L, n_bags, self.world, -1, self.world, cap, None,
                            row_of_bag, bag_of_row, cnt, ws, self.err)

    def gather(self, local_ids, n_bags, cap, bag_of_row, cnt, out_rows):
        ops.shard_bag_gather(self.W, self.ltoffs, self.T, local_ids, self.L, n_bags, self.world, cap, bag_of_row, cnt,
                             out_rows)

    def combine(self, rows, row_of_bag, n_bags, out):
        ops.shard_bag_combine(rows, row_of_bag, self.world, n_bags, self.T, self.dim, out)

    def pack(self, grad, bag_of_row, cnt, cap, rows_out):
        ops.shard_bag_pack(grad, self.T, self.dim, bag_of_row, cnt, self.world, cap, rows_out)

    def update(self, local_ids, grad_rows, row_of_bag, lr, eps):
        """Sort the owner's entries by local row and apply the fused update; entries = the local ids of
        all ranks' bags ([N*B*T][L], -1 where another rank holds the id)."""
        if self._plan is None or self._plan.n_entries != local_ids.numel():
            self._plan = ops.SortedPlan(self.ltoffs, self.T, max(self.local_rows, 1), local_ids, bag_len=self.L,
                                        runs=False)
        else:
            self._plan.prepare(local_ids)
        ops.bwd_apply_rows(self._plan, grad_rows, row_of_bag, self.W, _OPT[self.opt], lr, state=self.state, eps=eps)

    def check(self):
        ops.check_error_word(self.err, "sharded embedding bag")


class ShardedEmbeddingBag(object):
    def __init__(self, rows_per_table, dim, bag_len, device, opt='adagrad_rowwise', lr=1e-2, eps=1e-8, seed=0,
                 init_scale=0.05, capacity_factor=2.0, comm=None, backend=None, rank=None, world=None,
                 id_exchange='all_gather'):
        if opt not in _OPT:
            raise ValueError("opt must be one of %s" % sorted(_OPT))
        if id_exchange not in ('all_gather', 'bucketed'):
            raise ValueError("id_exchange must be 'all_gather' or 'bucketed'")
        self.bucketed = id_exchange == 'bucketed' 
        self.comm = comm if comm is not None else (TorchComm() if rank is None else None)
        self.rank = self.comm.rank if rank is None else rank
        self.world = self.comm.world if world is None else world
        self.rows_per_table = [int(r) for r in rows_per_table]
        self.T, self.dim, self.L = len(self.rows_per_table), int(dim), int(bag_len)
        self.opt, self.lr, self.eps = opt, lr, eps
        self.capacity_factor = float(capacity_factor)
        self.total_rows = sum(self.rows_per_table)
        self.be = backend if backend is not None else HipBagBackend(self.rows_per_table, self.dim, self.L, device, opt,
                                                                    self.rank, self.world)
        self.device = self.be.device
        if backend is None:
            g = torch.Generator(device=self.device)
            g.manual_seed(seed * 1000003 + self.rank)
            chunk = 1 << 22
            for lo in range(0, self.be.local_rows, chunk):
                hi = min(lo + chunk, self.be.local_rows)
                self.be.W[lo:hi] = torch.randn((hi - lo, self.dim), generator=g, dtype=torch.float32,
                                               device=self.device) * init_scale
        self._key = None
        self.steps = 0

    # -- parameters ---------------------------------------------------------------------------
    def load_weights(self, W_full):
        """Take this rank's rows (global rows rank, rank+N, ...) of a full table (numpy [total_rows, dim])."""
        self.be.W[:self.be.local_rows] = torch.from_numpy(W_full[self.rank::self.world].copy()).to(self.device)

    def capacity(self, B):
        n_bags = B * self.T
        if self.L > 1 or self.world == 1:
            return n_bags                      # bags of several ids: nearly every owner contributes to every bag
        return min(n_bags, int(self.capacity_factor * n_bags / self.world) + 64)

    def id_capacity(self, B):
        """Ids one rank pair may exchange in a step (bucketed id exchange)."""
        n = B * self.T * self.L
        return n if self.world == 1 else min(n, int(self.capacity_factor * n / self.world) + 64)

    # -- buffers ------------------------------------------------------------------------------
    def _buffers(self, ids):
        B = ids.shape[0]
        key = (B, ids.dtype)
        if self._key == key:
            return
        N, T, L, D, be = self.world, self.T, self.L, self.dim, self.be
        i32, f32 = torch.int32, torch.float32
        n_bags = B * T
        self.n_bags, self.cap = n_bags, self.capacity(B)
        cap = self.cap
        if self.bucketed:
            self.cap_ids = self.id_capacity(B)
            self.send_ids, self.recv_ids = be.zeros((N, self.cap_ids), i32), be.zeros((N * self.cap_ids,), i32)
            self.send_off = be.zeros((N, cap), torch.int64)
            self.recv_off = be.zeros((N * cap + 1,), torch.int64)
            self.recv_off[N * cap] = N * self.cap_ids          # the closing offset of the CSR bags, set once
        else:
            self.ids_all = be.zeros((N, B, T, L), ids.dtype)
            self.local_ids = be.zeros((N * n_bags, L), i32)
            self.own_row_of_bag, self.own_bag_of_row = be.zeros((N, n_bags), i32), be.zeros((N * cap,), i32)
            self.own_cnt = be.zeros((N,), i32)
        self.req_row_of_bag, self.req_bag_of_row = be.zeros((N, n_bags), i32), be.zeros((N * cap,), i32)
        self.req_cnt = be.zeros((N,), i32)
        self.ws = be.slots_workspace(n_bags, N)
        self.rows_out, self.rows_in = be.zeros((N, cap, D), f32), be.zeros((N, cap, D), f32)      # forward blocks
        self.grad_out, self.grad_in = be.zeros((N, cap, D), f32), be.zeros((N, cap, D), f32)      # backward blocks
        self.pooled = be.zeros((B, T * D), f32)
        self._key = key

    # -- the phases of a step (public so that a test can drive several ranks in lockstep) -------
    def phase_bucket(self, ids):
        """Bucketed id exchange, sender side: this rank's ids -> send_ids [N][cap_ids], send_off [N][cap] (two
        all-to-alls into recv_ids / recv_off[:N*cap])."""
        be = self.be
        be.requester_slots(ids, self.n_bags, self.cap, self.req_row_of_bag, self.req_bag_of_row, self.req_cnt, self.ws)
        be.bucket(ids, self.n_bags, self.cap, self.cap_ids, self.req_row_of_bag, self.req_cnt, self.send_ids,
                  self.send_off, self.ws)

    def phase_owner_forward(self):
        """self.ids_all (exchange A done) -> self.rows_out [N][cap][D] (exchange B: all-to-all into rows_in)."""
        be = self.be
        if self.bucketed:       # the received blocks ARE CSR bags over this rank's local rows
            be.gather_csr(self.recv_ids, self.recv_off, self.rows_out)
            return self.rows_out
        be.owner_slots(self.ids_all, self.n_bags, self.cap, self.local_ids, self.own_row_of_bag, self.own_bag_of_row,
                       self.own_cnt, self.ws)
        be.gather(self.local_ids, self.n_bags, self.cap, self.own_bag_of_row, self.own_cnt, self.rows_out)
        return self.rows_out

    def phase_combine(self, ids):
        """self.rows_in -> self.pooled [B][T*D]."""
        be = self.be
        if not self.bucketed:   # (bucketed: done before the id exchange)
            be.requester_slots(ids, self.n_bags, self.cap, self.req_row_of_bag, self.req_bag_of_row, self.req_cnt, self.ws)
        be.combine(self.rows_in, self.req_row_of_bag, self.n_bags, self.pooled)
        return self.pooled

    def phase_grad_pack(self, grad):
        """grad [B][T*D] of the pooled output -> self.grad_out [N][cap][D] (exchange C: all-to-all into grad_in)."""
        self.be.pack(grad, self.req_bag_of_row, self.req_cnt, self.cap, self.grad_out)
        return self.grad_out

    def phase_update(self):
        """self.grad_in -> fused sparse update of this rank's rows."""
        if self.bucketed:
            self.be.update_csr(self.recv_ids, self.recv_off, self.grad_in, self.lr, self.eps)
        else:
            self.be.update(self.local_ids, self.grad_in, self.own_row_of_bag, self.lr, self.eps)
        self.steps += 1

    # -- one step over torch.distributed --------------------------------------------------------
    def forward(self, ids):
        """ids [B][T][L] (this rank's batch) -> pooled [B][T*D] (sum pooling; a negative id is padding)."""
        ids = ids.reshape(ids.shape[0], self.T, self.L).contiguous()
        self._buffers(ids)
        self._ids = ids
        if self.bucketed:
            self.phase_bucket(ids)
            self.comm.all_to_all(self.recv_ids, self.send_ids.view(-1))                          # A: ids, by owner
            self.comm.all_to_all(self.recv_off[:self.world * self.cap], self.send_off.view(-1))  # A: slot offsets
        else:
            self.comm.all_gather(self.ids_all.view(-1), ids.view(-1))      # A
        self.phase_owner_forward()
        self.comm.all_to_all(self.rows_in.view(-1), self.rows_out.view(-1))     # B
        return self.phase_combine(ids)

    def backward(self, grad, check=False):
        """grad [B][T*D] = dL/d pooled of the last forward: the owners update their rows."""
        self.phase_grad_pack(grad)
        self.comm.all_to_all(self.grad_in.view(-1), self.grad_out.view(-1))     # C
        self.phase_update()
        if check:
            self.be.check()


def lockstep_step(models, ids_list, grads):
    """Drive the models of ALL ranks in one process, phase by phase, every exchange a tensor copy: the
    protocol and the kernels of ``forward`` / ``backward`` on a single GPU.  Returns the pooled outputs."""
    for m, ids in zip(models, ids_list):
        m._buffers(ids.reshape(ids.shape[0], m.T, m.L))
    if models[0].bucketed:
        for m, ids in zip(models, ids_list):
            m.phase_bucket(ids.reshape(ids.shape[0], m.T, m.L).contiguous())
        for o, mo in enumerate(models):                   # exchange A, bucketed: block o of every sender
            for s, ms in enumerate(models):
                mo.recv_ids.view(len(models), -1)[s].copy_(ms.send_ids[o])
                mo.recv_off[:len(models) * mo.cap].view(len(models), -1)[s].copy_(ms.send_off[o])
            mo.phase_owner_forward()
    else:
        for m in models:                                  # exchange A
            for s, ids in enumerate(ids_list):
                m.ids_all[s].copy_(ids.reshape(ids.shape[0], m.T, m.L))
            m.phase_owner_forward()
    for s, ms in enumerate(models):                       # exchange B
        for o, mo in enumerate(models):
            ms.rows_in[o].copy_(mo.rows_out[s])
    pooled = [m.phase_combine(ids.reshape(ids.shape[0], m.T, m.L).contiguous()) for m, ids in zip(models, ids_list)]
    if grads is None:
        return pooled
    for m, g in zip(models, grads):
        m.phase_grad_pack(g)
    for o, mo in enumerate(models):                       # exchange C
        for s, ms in enumerate(models):
            mo.grad_in[s].copy_(ms.grad_out[o])
    for m in models:
        m.phase_update()
    return pooled
