"""Row-wise sharded embedding bags on the GPU: the ranks' models driven in lockstep on one device
(every exchange a tensor copy, kernels the real ones) must reproduce the single-GPU embedding bag on
the full tables; the slot artefacts (index work) bit-exact against numpy; the real collectives over
RCCL at world size 1."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from numpy_bag_backend import bucket as np_bucket  # noqa: E402
from numpy_bag_backend import slots as np_slots  # noqa: E402
from oracle import embedding as oe  # noqa: E402

pytestmark = pytest.mark.gpu


def rel_err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def _ids(rng, B, rows, L, dist, pad):
    cols = []
    for r in rows:
        if dist == 'zipf':
            x = np.minimum((rng.pareto(1.05, size=(B, L))).astype(np.int64), r - 1)
        else:
            x = rng.integers(0, r, size=(B, L))
        cols.append(x)
    ids = np.stack(cols, axis=1).astype(np.int64)
    if pad:
        ids[rng.random(ids.shape) < pad] = -1
    return ids


@pytest.mark.parametrize("N,L,idt", [(2, 1, np.int64), (8, 1, np.int32), (8, 3, np.int64), (3, 2, np.int64)])
def test_slots_bit_exact(dev, N, L, idt):
    from recommendation_amd import ops
    rng = np.random.default_rng(5)
    rows = [1000, 3, 50000, 17, 250]
    T, B = len(rows), 777
    n_bags = B * T
    toffs = np.concatenate([[0], np.cumsum(rows)]).astype(np.int64)
    ids_all = np.stack([_ids(rng, B, rows, L, 'zipf' if s % 2 else 'uniform', 0.2 if L > 1 else 0.0)
                        for s in range(N)]).astype(idt)
    td = torch.from_numpy(toffs).to(dev)
    i32 = dict(dtype=torch.int32, device=dev)
    for cap in (n_bags, n_bags // N + 40):
        ws = torch.zeros(ops.shard_bag_slots_workspace_bytes(n_bags, N), dtype=torch.uint8, device=dev)
        for owner, src in [(o, ids_all) for o in range(N)] + [(-1, ids_all[1:2])]:
            err = ops.new_error_word(dev)
            li = torch.full((N * n_bags, L), -9, **i32) if owner >= 0 else None
            rb, br, cnt = torch.full((N, n_bags), -9, **i32), torch.full((N * cap,), -9, **i32), torch.full((N,), -9, **i32)
            ops.shard_bag_slots(td, T, torch.from_numpy(np.ascontiguousarray(src)).to(dev), L, n_bags, N, owner, N, cap,
                                li, rb, br, cnt, ws, err)
            wli, wrb, wbr, wcnt, overflow = np_slots(toffs, T, src, L, n_bags, N, owner, N, cap)
            assert np.array_equal(cnt.cpu().numpy(), wcnt)
            assert np.array_equal(rb.cpu().numpy(), wrb)
            got_br = br.cpu().numpy()
            for p in range(N):
                assert np.array_equal(got_br[p * cap:p * cap + wcnt[p]], wbr[p * cap:p * cap + wcnt[p]])
            if owner >= 0:
                assert np.array_equal(li.cpu().numpy().reshape(N, n_bags, L), wli)
            assert bool(int(err.item()) & ops.MI_IDERR_CAPACITY) == overflow


@pytest.mark.parametrize("N,L,idt", [(2, 1, np.int64), (8, 1, np.int32), (8, 5, np.int64), (3, 20, np.int32)])
def test_bucketed_id_blocks_bit_exact(dev, N, L, idt):
    """mi_shard_bag_bucket against its numpy statement: every owner's block of local rows and slot offsets, for
    every sending rank; a too small id capacity is reported."""
    from recommendation_amd import ops
    rng = np.random.default_rng(6)
    rows = [1000, 3, 50000, 17, 250]
    T, B = len(rows), 777
    n_bags = B * T
    toffs = np.concatenate([[0], np.cumsum(rows)]).astype(np.int64)
    td = torch.from_numpy(toffs).to(dev)
    i32 = dict(dtype=torch.int32, device=dev)
    ids = _ids(rng, B, rows, L, 'zipf', 0.2 if L > 1 else 0.0).astype(idt)
    idd = torch.from_numpy(ids).to(dev)
    cap = n_bags if L > 1 else n_bags // N + 400
    for cap_ids in (n_bags * L, n_bags * L // N + 4000, 50):
        for rank in (0, N - 1):
            ws = torch.zeros(max(ops.shard_bag_slots_workspace_bytes(n_bags, N), ops.shard_bag_bucket_workspace_bytes(n_bags, N)),
                             dtype=torch.uint8, device=dev)
            err = ops.new_error_word(dev)
            rb, br, cnt = torch.full((N, n_bags), -9, **i32), torch.full((N * cap,), -9, **i32), torch.full((N,), -9, **i32)
            ops.shard_bag_slots(td, T, idd, L, n_bags, N, -1, N, cap, None, rb, br, cnt, ws, err)
            send_ids = torch.full((N, cap_ids), 7, **i32)
            send_off = torch.full((N, cap), -7, dtype=torch.int64, device=dev)
            err2 = ops.new_error_word(dev)         # (the slots above may have run out of bag capacity on their own)
            ops.shard_bag_bucket(td, T, idd, L, n_bags, rank, N, cap, cap_ids, rb, cnt, send_ids, send_off, ws, err2)
            want_ids, want_off, overflow = np_bucket(toffs, T, ids, L, n_bags, rank, N, cap, cap_ids, rb.cpu().numpy())
            assert bool(int(err2.item()) & ops.MI_IDERR_CAPACITY) == overflow
            assert overflow == (cap_ids == 50)
            assert np.array_equal(send_off.cpu().numpy(), want_off)
            if not overflow:
                assert np.array_equal(send_ids.cpu().numpy(), want_ids)


@pytest.mark.parametrize("exchange", ["all_gather", "bucketed"])
@pytest.mark.parametrize("N,L,dim,opt", [(2, 1, 128, "adagrad_rowwise"), (4, 1, 16, "sgd"), (8, 1, 128, "adagrad_rowwise"),
                                         (8, 4, 128, "adagrad_rowwise"), (3, 2, 32, "sgd"), (8, 20, 64, "adagrad_rowwise")])
def test_lockstep_ranks_equal_the_single_gpu_embedding_bag(dev, N, L, dim, opt, exchange):
    """Pooled rows of every rank == mi_embag_fwd on the full table (one-id bags: the same bits); after two
    steps the union of the shards == the single-GPU sort-by-row update of the concatenated batch.  With either id
    exchange: the all-gather of ids, or the all-to-all of ids bucketed by owner."""
    from recommendation_amd import ops
    from recommendation_amd.sharded_bag import ShardedEmbeddingBag, lockstep_step
    rng = np.random.default_rng(11)
    rows = [1000, 3, 50000, 17, 250, 8]
    T, B, lr = len(rows), 512, 0.05
    toffs = np.concatenate([[0], np.cumsum(rows)]).astype(np.int64)
    total = int(toffs[-1])
    W0 = (rng.standard_normal((total, dim)) * 0.1).astype(np.float32)
    models = [ShardedEmbeddingBag(rows, dim, L, dev, opt=opt, lr=lr, rank=r, world=N, capacity_factor=3.0,
                                  id_exchange=exchange) for r in range(N)]
    for m in models:
        m.load_weights(W0)
    Wd = torch.from_numpy(W0.copy()).to(dev)
    st = torch.zeros((total,), device=dev)
    td = torch.from_numpy(toffs).to(dev)
    for step in range(2):
        ids = [_ids(rng, B, rows, L, 'zipf' if (s + step) % 2 else 'uniform', 0.15 if L > 1 else 0.0) for s in range(N)]
        grads = [rng.standard_normal((B, T * dim)).astype(np.float32) for _ in range(N)]
        idd = [torch.from_numpy(x).to(dev) for x in ids]
        pooled = lockstep_step(models, idd, [torch.from_numpy(g).to(dev) for g in grads])
        for s in range(N):
            want = ops.embag_fwd(Wd, td, T, idd[s], bag_len=L)
            if L == 1 and step == 0:          # gathered rows: the same bits (later steps compare tables that
                assert torch.equal(pooled[s], want)   # two differently ordered run sums have updated)
            else:
                assert rel_err(pooled[s].cpu().numpy(), want.cpu().numpy()) <= 1e-5
        cat = torch.cat(idd)
        plan = ops.SortedPlan(td, T, total, cat, bag_len=L, runs=False)
        g = torch.from_numpy(np.concatenate(grads)).to(dev)
        if opt == 'sgd':
            ops.bwd_apply_sgd(plan, g, Wd, lr)
        else:
            ops.bwd_apply_adagrad_rowwise(plan, g, Wd, st, lr, 1e-8)
        for m in models:
            m.be.check()
    full = np.zeros_like(W0)
    for r, m in enumerate(models):
        full[r::N] = m.be.W[:m.be.local_rows].cpu().numpy()
    assert rel_err(full, Wd.cpu().numpy()) <= 1e-6
    # and against the oracle's formulas on the first step's pooled rows of rank 0 (gathered rows)
    assert (full != W0).any()


def test_capacity_overflow_is_reported_not_hidden(dev):
    from recommendation_amd import ops
    from recommendation_amd.sharded_bag import ShardedEmbeddingBag, lockstep_step
    rows, N, B = [64, 64], 4, 256
    models = [ShardedEmbeddingBag(rows, 16, 1, dev, opt='sgd', rank=r, world=N, capacity_factor=1.0) for r in range(N)]
    ids = [torch.zeros((B, 2, 1), dtype=torch.int64, device=dev) for _ in range(N)]       # every id is row 0 / row 64: rank 0 owns all
    lockstep_step(models, ids, None)
    with pytest.raises(ops.InvalidArgumentError):
        models[0].be.check()


@pytest.mark.parametrize("exchange", ["all_gather", "bucketed"])
def test_forward_backward_over_rccl_world1_equals_lockstep(dev, exchange):
    import torch.distributed as dist
    from recommendation_amd.sharded_bag import ShardedEmbeddingBag, lockstep_step
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29613")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(dev))
    try:
        rng = np.random.default_rng(3)
        rows, L, dim, B = [1000, 3, 50000, 17], 2, 128, 300
        W0 = (rng.standard_normal((sum(rows), dim)) * 0.1).astype(np.float32)
        a = ShardedEmbeddingBag(rows, dim, L, dev, lr=0.05, id_exchange=exchange)
        b = ShardedEmbeddingBag(rows, dim, L, dev, lr=0.05, rank=0, world=1, id_exchange=exchange)
        for m in (a, b):
            m.load_weights(W0)
        for step in range(3):
            ids = torch.from_numpy(_ids(rng, B, rows, L, 'zipf', 0.1)).to(dev)
            g = torch.from_numpy(rng.standard_normal((B, len(rows) * dim)).astype(np.float32)).to(dev)
            pa = a.forward(ids)
            a.backward(g, check=True)
            pb = lockstep_step([b], [ids], [g])[0]
            assert torch.equal(pa, pb)
        assert torch.equal(a.be.W, b.be.W) and torch.equal(a.be.state, b.be.state)
    finally:
        dist.destroy_process_group()
