"""DCN-v2 over several GPUs: the tables row-wise sharded, the interaction network data parallel.

The reference distributes EVERY variable over parameter servers and runs the same graph on every worker
(recommendation/code/model_trainer.py:124-136 replica_device_setter + :135-158, recommendation/train_job.py:41-63):
a worker pulls the rows it needs, pushes their gradients, and the dense variables' gradients of all workers meet
on the servers.  Here a rank owns the rows ``r % world == rank`` of every table (ShardedEmbeddingBag: ids out,
rows back, row gradients to the owners, which apply them), keeps a full replica of the cross network and the MLP,
and the replicas stay identical because every step adds the ranks' dense gradients before applying them:

  pooled   = bag.forward(ids)                       exchanges A, B
  loss     = engine.step(pooled, dense, labels)     forward + backward on this rank's B samples, gradients only;
                                                    the loss gradient is scaled by 1 / (B * world)
  comm.all_reduce(engine.grad_arena)                ONE collective over every dense gradient (sum = global mean's)
  engine.apply()                                    the same Adam step on every rank
  bag.backward(engine.dpooled)                      exchange C, owners update their rows

``engine`` is the local arithmetic: DCNv2(embeddings=False, grads_only=True) on the GPU; tests replace it (and the
bag's backend) by numpy stand-ins to run the protocol over gloo on CPUs."""
import torch

from . import ops
from .sharded import TorchComm
from .sharded_bag import ShardedEmbeddingBag


class HipDcnEngine(object):
    """The interaction network of one rank on its GPU."""

    def __init__(self, rows_per_table, dim, n_dense, device, world, **kw):
        from .dcn import DCNv2
        self.net = DCNv2(rows_per_table, dim, n_dense, device, embeddings=False, grads_only=True, loss_ranks=world, **kw)
        self.T, self.dim = self.net.T, self.net.dim
        self.grad_arena = self.net.grad_arena

    def step(self, ids, pooled, dense, labels):
        """Forward and backward over this rank's samples; returns the local mean loss (1-element tensor)."""
        loss = self.net.train_step(ids, dense, labels, pooled=pooled)
        self.dpooled = self.net.dx0[:, :self.T * self.dim]          # rows dp floats apart
        return loss

    def apply(self):
        self.net.apply_dense()


class ShardedDCN(object):
    def __init__(self, rows_per_table, dim, n_dense, device, comm=None, rank=None, world=None, engine=None,
                 bag_backend=None, emb_lr=1e-2, emb_eps=1e-8, seed=0, capacity_factor=2.0, **kw):
        self.comm = comm if comm is not None else (TorchComm() if rank is None else None)
        self.rank = self.comm.rank if rank is None else rank
        self.world = self.comm.world if world is None else world
        self.bag = ShardedEmbeddingBag(rows_per_table, dim, 1, device, opt='adagrad_rowwise', lr=emb_lr, eps=emb_eps,
                                       seed=seed, comm=self.comm, backend=bag_backend, rank=self.rank, world=self.world,
                                       capacity_factor=capacity_factor)
        self.engine = engine if engine is not None else HipDcnEngine(rows_per_table, dim, n_dense, device, self.world,
                                                                     seed=seed, **kw)
        self.steps = 0

    # -- the phases between the collectives (public so that a test can drive several ranks in lockstep) --------
    def phase_dense(self, ids, pooled, dense, labels):
        return self.engine.step(ids, pooled, dense, labels)

    def phase_apply(self):
        self.engine.apply()
        self.steps += 1

    # -- one step over the communicator -------------------------------------------------------------------------
    def train_step(self, ids, dense, labels, check=False):
        """ids [B][T], dense [B][n_dense], labels [B]: this rank's part of the global batch.  Returns this rank's
        mean loss."""
        pooled = self.bag.forward(ids)
        loss = self.phase_dense(ids, pooled, dense, labels)
        if self.world > 1:
            self.comm.all_reduce(self.engine.grad_arena)
        self.phase_apply()
        self.bag.backward(self.engine.dpooled, check=check)
        return loss


def lockstep_dcn_step(models, batches):
    """Drive the models of ALL ranks in one process, every exchange a tensor copy and the all-reduce a sum in rank
    order: the protocol and the kernels of ``train_step`` on a single GPU.  Returns the ranks' mean losses."""
    from .sharded_bag import lockstep_step
    bags = [m.bag for m in models]
    ids_list = [b[0] for b in batches]
    pooled = lockstep_step(bags, ids_list, None)
    losses = [m.phase_dense(b[0], p, b[1], b[2]).clone() for m, b, p in zip(models, batches, pooled)]
    total = models[0].engine.grad_arena.clone()
    for m in models[1:]:
        total += m.engine.grad_arena
    for m in models:
        m.engine.grad_arena.copy_(total)
        m.phase_apply()
    for m in models:
        m.bag.phase_grad_pack(m.engine.dpooled)
    for o, mo in enumerate(bags):                       # exchange C
        for s, ms in enumerate(bags):
            mo.grad_in[s].copy_(ms.grad_out[o])
    for m in bags:
        m.phase_update()
    return losses
