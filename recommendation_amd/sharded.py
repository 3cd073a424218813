"""Row-wise sharded embedding tables across the GPUs of one node (one process per GPU).

The reference distributes with TensorFlow's asynchronous parameter server: 2 workers + 1 ps,
every variable read and Adam apply a gRPC hop (recommendation/code/model_trainer.py:124-136,
recommendation/train_job.py:41-42).  That scheme is not reproduced.  Here

* the rows of all tables are dealt round-robin over the N ranks (global row r = offset[t] + id
  lives on rank r % N as local row r / N), so any id distribution loads the ranks evenly;
* the small dense side (dense-feature weights, bias) is replicated;
* a step is synchronous: every rank contributes its own batch of B samples, the loss is the mean
  over the N*B samples, and after the step all replicas agree bit for bit.

A step has THREE collectives (RCCL over xGMI through torch.distributed, backend "nccl"), all of
fixed size, and FOUR launches; nothing on the host ever waits for the device:

  launch 1  route: ids [B][T] -> per owner o: slot[t][b] = local row when o owns it, else -1, and
            how many valid slots every 64-sample chunk has per table
  A. all-to-all  [N][blk] int32       requester -> owner (blk = chunk counts + B*T slots)
  launch 2  owner: partial forward per (source rank, sample) | per-table LDS sort of the valid
            slots (payload = row of X_all) | Adam schedule step
  B. all-to-all  [N][B][D+2] fp32     owner -> requester: per SAMPLE sum v, sum |v|^2, sum w1
                                      (not per (sample, field))
  launch 3  combine: sum the owners' partials in rank order, logit, loss, dL/dlogit -> X rows
  C. all-gather  [B+e][D+2 -> 20] fp32 every sample's S, dL/dlogit and loss row, and behind them the
                                      rank's dense-side partial sums (the dense features stay home)
  launch 4  owner: sparse backward + fused optimizer | dense side: all ranks' partials added

After C every owner has what its sparse backward needs and every replica adds the same dense-side
partials in the same order, so there is no all-reduce and the replicas stay bit-identical.  The sparse
backward and the fused optimizer run entirely on the owner: no gradient row crosses the fabric.
Local compute goes through a small backend object so that the exchange logic can be exercised
with gloo on CPU (tests plug in a numpy stand-in); the product backend is ``HipBackend`` and there
is no other.
"""
import torch
import torch.distributed as dist

from . import ops
from .fm import _OPT


def local_row_count(total_rows, rank, world):
    return (total_rows - rank + world - 1) // world


class HipBackend(object):
    """Local compute of one rank: every step piece is one C-ABI call, one launch."""

    def __init__(self, rows_per_table, dim, n_dense, device, opt, rank, world):
        self.device = torch.device(device)
        self.T, self.dim, self.n_dense, self.opt = len(rows_per_table), dim, n_dense, opt
        self.rank, self.world = rank, world
        self.total_rows = int(sum(rows_per_table))
        self.local_rows = local_row_count(self.total_rows, rank, world)
        self.toffs = ops.table_offsets(rows_per_table, self.device)
        self.table = ops.FmRecords(max(self.local_rows, 1), dim, _OPT[opt], self.device)
        self.err = ops.new_error_word(self.device)

    # parameters ------------------------------------------------------------------------------
    def init_rows(self, V, w1):
        """V [local_rows, dim], w1 [local_rows] for this shard."""
        self.table.rec[:self.local_rows, :self.dim] = V.to(self.device)
        self.table.rec[:self.local_rows, self.table.off_w1] = w1.to(self.device)

    def rows(self):
        return (self.table.rec[:self.local_rows, :self.dim], self.table.rec[:self.local_rows, self.table.off_w1])

    def zeros(self, shape, dtype):
        return torch.zeros(shape, dtype=dtype, device=self.device)

    def bwd_scratch(self, n_slots):
        return self.zeros((ops.shard_bwd_scratch_floats(n_slots),), torch.float32)

    def row_floats(self):
        """Floats per gathered sample row: S, dL/dlogit, loss (the dense features stay at home)."""
        return ops.shard_row_floats(self.dim)

    def extra_rows(self, B, ld):
        """Rows behind the B sample rows of X that carry the combine launch's dense-side partials."""
        return -(-ops.shard_dense_part_floats(B, self.dim, self.n_dense) // ld)

    # step pieces: one launch each ---------------------------------------------------------------
    def route(self, ids, send):
        ops.shard_route(self.toffs, ids, self.world, send, self.err)

    def owner_forward(self, recv, B, row_stride, partial, sorted_rows, perm, n_total, sched, lr, beta1, beta2):
        ops.shard_owner_fwd_sort(self.table, max(self.local_rows, 1), self.toffs, self.rank, self.world, recv, B,
                                 partial, sorted_rows, perm, n_total, sched, lr, beta1, beta2, self.err,
                                 row_stride=row_stride)

    def combine(self, partials, B, dense, wd, bias, labels, gscale, X):
        ops.fm_combine(partials, self.world, B, self.dim, dense, wd, bias, labels, gscale, X, part_out=X[B:])

    def update(self, sorted_rows, perm, n_total, X_all, B, row_stride, dp, dm, dv, lr, beta1, beta2, eps, sched,
               scratch, loss):
        adam = self.opt == 'adam_tf'
        ops.shard_owner_bwd_dense(self.table, max(self.local_rows, 1), sorted_rows, perm, n_total,
                                  sorted_rows.numel(), X_all, self.world, B, row_stride,
                                  _OPT[self.opt], lr, beta1, beta2, eps, self.n_dense, dp, dm, dv, 0.0, scratch, loss,
                                  lr_dev=sched if adam else None, dense_lr_dev=sched)

    def check(self):
        ops.check_error_word(self.err, "sharded FM ids")


class TorchComm(object):
    """The collectives of a step, on torch.distributed (RCCL when the backend is nccl)."""

    def __init__(self, group=None):
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)

    def all_to_all(self, out, inp):
        dist.all_to_all_single(out, inp, group=self.group)

    def all_gather(self, out, inp):
        dist.all_gather_into_tensor(out, inp, group=self.group)

    def all_reduce(self, buf):
        """buf (fp32, in place) <- the sum of every rank's buf."""
        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group)


class RcclComm(object):
    """The same collectives through the library's own RCCL entry points (mi_comm_*, mi_a2a_*,
    mi_allgather_bytes, mi_allreduce_dense) instead of torch.distributed: what a host without
    PyTorch's process groups would call.  The 128-byte RCCL id is made by rank 0 and handed round
    by ``exchange_id(id_bytes_or_None) -> id_bytes`` (default: torch.distributed's object
    broadcast when a process group is up; a single rank needs no exchange)."""

    def __init__(self, rank=None, world=None, exchange_id=None):
        import ctypes
        from . import _lib
        if rank is None:
            up = dist.is_available() and dist.is_initialized()
            rank, world = (dist.get_rank(), dist.get_world_size()) if up else (0, 1)
        self.rank, self.world = int(rank), int(world)
        buf = ctypes.create_string_buffer(_lib.MI_COMM_ID_BYTES)
        if self.rank == 0:
            _lib.call('mi_comm_unique_id', buf)
        if self.world > 1:
            if exchange_id is None:
                box = [bytes(buf.raw) if self.rank == 0 else None]
                dist.broadcast_object_list(box, src=0)
                ident = box[0]
            else:
                ident = exchange_id(bytes(buf.raw) if self.rank == 0 else None)
            buf = ctypes.create_string_buffer(ident, _lib.MI_COMM_ID_BYTES)
        self._h = ctypes.c_void_p()
        _lib.call('mi_comm_init', buf, self.world, self.rank, ctypes.byref(self._h))

    @staticmethod
    def _stream():
        return ops._stream()

    def all_to_all(self, out, inp):
        from . import _lib
        n = inp.numel() // self.world
        if inp.dtype == torch.float32:
            _lib.call('mi_a2a_pooled', self._h, inp.data_ptr(), out.data_ptr(), n, self._stream())
        elif inp.dtype == torch.int32:
            _lib.call('mi_a2a_indices', self._h, inp.data_ptr(), out.data_ptr(), n, self._stream())
        elif inp.dtype == torch.int64:          # as pairs of int32
            _lib.call('mi_a2a_indices', self._h, inp.data_ptr(), out.data_ptr(), 2 * n, self._stream())
        else:
            raise ops.MiError("RcclComm.all_to_all: int32 or float32 blocks only")

    def all_gather(self, out, inp):
        from . import _lib
        _lib.call('mi_allgather_bytes', self._h, inp.data_ptr(), out.data_ptr(), inp.numel() * inp.element_size(),
                  self._stream())

    def all_reduce(self, buf):
        from . import _lib
        _lib.call('mi_allreduce_dense', self._h, buf.data_ptr(), buf.numel(), self._stream())

    def close(self):
        from . import _lib
        if self._h:
            _lib.call('mi_comm_destroy', self._h)
            self._h = None


class ShardedFactorizationMachine(object):
    def __init__(self, rows_per_table, dim, n_dense, device, opt='adam_tf', lr=1e-3, beta1=0.9, beta2=0.999,
                 eps=1e-8, seed=0, init_scale=0.05, comm=None, backend=None, rank=None, world=None):
        self.comm = comm if comm is not None else (TorchComm() if rank is None else None)
        self.rank = self.comm.rank if rank is None else rank
        self.world = self.comm.world if world is None else world
        self.rows_per_table = [int(r) for r in rows_per_table]
        self.T, self.dim, self.n_dense = len(rows_per_table), int(dim), int(n_dense)
        self.opt, self.lr, self.beta1, self.beta2, self.eps = opt, lr, beta1, beta2, eps
        self.total_rows = sum(self.rows_per_table)
        self.be = backend if backend is not None else HipBackend(self.rows_per_table, self.dim, self.n_dense, device,
                                                                 opt, self.rank, self.world)
        self.device = self.be.device
        f32 = torch.float32
        # replicated dense side; same seed on every rank -> identical replicas
        gen = torch.Generator(device='cpu')
        gen.manual_seed(seed)
        dp = torch.zeros(self.n_dense + 1, dtype=f32)
        dp[:self.n_dense] = torch.randn(self.n_dense, generator=gen, dtype=f32) * init_scale
        self.dp = dp.to(self.device)
        self.dm, self.dv = torch.zeros_like(self.dp), torch.zeros_like(self.dp)
        self.sched = torch.tensor([0.0, 1.0, 1.0, 0.0], dtype=f32, device=self.device)
        # this rank's rows: every rank seeds its own generator
        if backend is None:
            g = torch.Generator(device=self.device)
            g.manual_seed(seed * 1000003 + self.rank)
            n = self.be.local_rows
            chunk = 1 << 22
            for lo in range(0, n, chunk):
                hi = min(lo + chunk, n)
                self.be.table.rec[lo:hi, :self.dim + 1] = torch.randn((hi - lo, self.dim + 1), generator=g,
                                                                      dtype=f32, device=self.device) * init_scale
        self.err = getattr(self.be, 'err', None)
        self._B = None
        self.loss = None
        self._prefetched, self._side = None, None

    # -- parameters ---------------------------------------------------------------------------
    @property
    def wd(self):
        return self.dp[:self.n_dense]

    @property
    def bias(self):
        return self.dp[self.n_dense:]

    def load_state(self, st):
        """Take this rank's rows (global rows rank, rank+N, ...) from a full-table state."""
        V = torch.from_numpy(st.V[self.rank::self.world].copy())
        w1 = torch.from_numpy(st.w1[self.rank::self.world].copy())
        self.be.init_rows(V, w1)
        self.dp[:self.n_dense] = torch.from_numpy(st.wd).to(self.device)
        self.dp[self.n_dense] = float(st.bias)

    # -- buffers ------------------------------------------------------------------------------
    def _buffers(self, B):
        if self._B == B:
            return
        N, D, T, be = self.world, self.dim, self.T, self.be
        i32, f32 = torch.int32, torch.float32
        self.ld = be.row_floats()
        blk = ops.shard_block_ints(B, T)
        self.send, self.recv = be.zeros((N, blk), i32), be.zeros((N, blk), i32)
        # room for every entry of every rank (an owner can receive them all)
        self.sorted_rows, self.perm = be.zeros((N * B * T,), i32), be.zeros((N * B * T,), i32)
        self.n_total = be.zeros((1,), i32)
        self.partial, self.partials_in = be.zeros((N, B, D + 2), f32), be.zeros((N, B, D + 2), f32)
        # per rank: B sample rows, then the rows that carry its dense-side partials
        self.rs = B + be.extra_rows(B, self.ld)
        self.X, self.X_all = be.zeros((self.rs, self.ld), f32), be.zeros((N * self.rs, self.ld), f32)
        self.scratch = be.bwd_scratch(N * B * T)
        self.loss = be.zeros((1,), f32)
        self._B = B

    # -- the phases of a step (public so that a test can drive several ranks in lockstep) -------
    def phase_route(self, ids):
        """ids [B][T] -> self.send [N][blk] (exchange A: all-to-all into self.recv)."""
        self._buffers(ids.shape[0])
        self.be.route(ids, self.send)
        return self.send

    def phase_owner_forward(self):
        """self.recv -> self.partial [N][B][D+2] (exchange B: all-to-all into self.partials_in), and
        the sorted entries the backward will take."""
        self.be.owner_forward(self.recv, self._B, self.rs, self.partial, self.sorted_rows, self.perm, self.n_total,
                              self.sched, self.lr, self.beta1, self.beta2)
        return self.partial

    def phase_combine(self, dense, labels):
        """self.partials_in -> self.X [rs][ld]: B sample rows + dense-side partials (exchange C:
        all-gather into self.X_all)."""
        B = self._B
        self.be.combine(self.partials_in, B, dense if self.n_dense else None, self.wd if self.n_dense else None,
                        self.bias, labels, 1.0 / (B * self.world), self.X)
        return self.X

    def phase_update(self):
        """Sparse backward + optimizer on the owner, dense step on every replica."""
        self.be.update(self.sorted_rows, self.perm, self.n_total, self.X_all, self._B, self.rs, self.dp, self.dm,
                       self.dv, self.lr, self.beta1, self.beta2, self.eps, self.sched, self.scratch, self.loss)

    # -- one step over torch.distributed --------------------------------------------------------
    def _route_and_send(self, ids):
        self.phase_route(ids)
        self.comm.all_to_all(self.recv, self.send)         # A

    def train_step(self, ids, dense, labels, check_ids=False, next_ids=None):
        """One synchronous step.  ``next_ids``: the ids of the batch the NEXT call will be given;
        their routing and exchange A depend on no model state, so they are issued on a side stream
        as soon as this step's owner forward has consumed the receive buffer and run beside this
        step's update instead of in front of the next step."""
        c = self.comm
        cuda = self.device.type == 'cuda'
        routed = None
        if self._prefetched is not None:
            routed, arrived = self._prefetched
            torch.cuda.current_stream(self.device).wait_event(arrived)   # the prefetched exchange A is done
            self._prefetched = None
        if routed != (ids.data_ptr(), ids.shape[0]):
            self._route_and_send(ids)
        self.phase_owner_forward()
        if cuda and next_ids is not None:
            if self._side is None:
                self._side = torch.cuda.Stream(self.device)
                self._consumed, self._arrived = torch.cuda.Event(), torch.cuda.Event()
            self._consumed.record()                         # recv and send are free again after this point
        c.all_to_all(self.partials_in, self.partial)       # B
        self.phase_combine(dense, labels)
        c.all_gather(self.X_all, self.X)                   # C
        if cuda and next_ids is not None:
            # issued after C so that every rank enqueues its collectives in the same order
            with torch.cuda.stream(self._side):
                self._side.wait_event(self._consumed)
                self._route_and_send(next_ids)
                self._arrived.record()
            self._prefetched = ((next_ids.data_ptr(), next_ids.shape[0]), self._arrived)
        self.phase_update()
        if check_ids and hasattr(self.be, 'check'):
            self.be.check()
        return self.loss


def lockstep_train_step(models, batches):
    """Drive the models of ALL ranks in one process, phase by phase, doing each exchange with plain
    tensor copies.  Same phases and buffers as ``train_step`` over torch.distributed; used to check
    the protocol and the kernels on a single GPU."""
    B = batches[0][0].shape[0]
    for m, (ids, _, _) in zip(models, batches):
        m.phase_route(ids)
    for r, m in enumerate(models):                       # exchange A
        for s, ms in enumerate(models):
            m.recv[s].copy_(ms.send[r])
        m.phase_owner_forward()
    for s, ms in enumerate(models):                      # exchange B
        for r, m in enumerate(models):
            ms.partials_in[r].copy_(m.partial[s])
    for m, (_, dense, labels) in zip(models, batches):
        m.phase_combine(dense, labels)
    for m in models:                                     # exchange C
        for s, ms in enumerate(models):
            m.X_all[s * m.rs:(s + 1) * m.rs].copy_(ms.X)
    for m in models:
        m.phase_update()
    return models[0].loss
