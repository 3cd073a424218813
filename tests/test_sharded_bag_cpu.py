"""The exchange logic of the row-wise sharded embedding bag over torch.distributed (gloo, world 2
and 3, CPU): local compute is the numpy stand-in, the three collectives of a step are real.  Pooled
outputs and updated tables must equal a single-process evaluation on the full tables."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

ROWS = [50, 7, 1201, 3]
DIM, B, STEPS, LR = 8, 40, 2, 0.05


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _batch(step, rank, L):
    rng = np.random.default_rng(100 * step + rank)
    ids = np.stack([rng.integers(-1 if L > 1 else 0, r, size=(B, L)) for r in ROWS], axis=1).astype(np.int64)
    grad = rng.standard_normal((B, len(ROWS) * DIM)).astype(np.float32)
    return ids, grad


def _full_table():
    return (np.random.default_rng(7).standard_normal((sum(ROWS), DIM)) * 0.1).astype(np.float32)


def _rank_main(rank, world, port, opt, L, out_dir, exchange='all_gather'):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from numpy_bag_backend import NumpyBagBackend
        from recommendation_amd.sharded_bag import ShardedEmbeddingBag
        be = NumpyBagBackend(ROWS, DIM, L, opt, rank, world)
        m = ShardedEmbeddingBag(ROWS, DIM, L, device='cpu', opt=opt, lr=LR, backend=be, capacity_factor=4.0,
                                id_exchange=exchange)
        m.load_weights(_full_table())
        pooled = []
        for step in range(STEPS):
            ids, grad = _batch(step, rank, L)
            pooled.append(m.forward(torch.from_numpy(ids)).numpy().copy())
            m.backward(torch.from_numpy(grad), check=True)
        np.savez(os.path.join(out_dir, "rank%d.npz" % rank), W=be.W.numpy()[:be.local_rows], pooled=np.stack(pooled))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,opt,L,exchange", [(2, "sgd", 1, "all_gather"), (2, "adagrad_rowwise", 3, "all_gather"),
                                                  (3, "adagrad_rowwise", 1, "all_gather"), (2, "adagrad_rowwise", 3, "bucketed"),
                                                  (3, "sgd", 1, "bucketed")])
def test_sharded_bag_over_gloo_equals_single_process(tmp_path, world, opt, L, exchange):
    from oracle import embedding as oe
    mp.spawn(_rank_main, args=(world, _free_port(), opt, L, str(tmp_path), exchange), nprocs=world, join=True)
    toffs = np.concatenate([[0], np.cumsum(ROWS)]).astype(np.int64)
    W = _full_table().astype(np.float64)
    st = np.zeros(W.shape[0])
    got = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % r)) for r in range(world)]
    for step in range(STEPS):
        parts = [_batch(step, r, L) for r in range(world)]
        for r, (ids, _) in enumerate(parts):
            want = oe.embag_fwd(W, toffs, len(ROWS), ids, bag_len=L)
            np.testing.assert_allclose(got[r]["pooled"][step], want, rtol=1e-5, atol=1e-6)
        ids = np.concatenate([p[0] for p in parts])
        grad = np.concatenate([p[1] for p in parts])
        grads = oe.row_grads(toffs, len(ROWS), ids, grad, DIM, bag_len=L)
        if opt == 'sgd':
            oe.apply_sgd(W, grads, LR)
        else:
            oe.apply_adagrad_rowwise(W, st, grads, LR, 1e-8)
    full = np.zeros_like(W)
    for r, g in enumerate(got):
        full[r::world] = g["W"]
    np.testing.assert_allclose(full, W, rtol=1e-5, atol=1e-6)


def test_local_table_offsets_are_the_counts_of_owned_rows_before_each_table():
    from recommendation_amd.sharded_bag import local_table_offsets
    for rows in ([50, 7, 1201, 3], [1, 1, 1], [8, 8]):
        toffs = np.concatenate([[0], np.cumsum(rows)])
        for world in (1, 2, 3, 8):
            for rank in range(world):
                want = [int(((np.arange(t) % world) == rank).sum()) for t in toffs]
                assert local_table_offsets(rows, rank, world) == want
