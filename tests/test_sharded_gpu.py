"""Row-wise sharded FM: N ranks' models driven in lockstep on ONE GPU (each exchange a tensor
copy) must train exactly like the single-GPU FM on the concatenated batch."""
import numpy as np
import pytest
import torch

from oracle import fm as ofm
from tests.errlog import held

ADAM_TOL = 3e-5   # sharded against single-GPU Adam, V after three steps (achieved: 1.5e-5; twice that)

pytestmark = pytest.mark.gpu


def rel_err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


@pytest.mark.parametrize("N", [2, 4, 8])
@pytest.mark.parametrize("opt", ["adagrad_rowwise", "adam_tf"])
def test_sharded_equals_single_gpu(dev, N, opt):
    from recommendation_amd.fm import FactorizationMachine
    from recommendation_amd.sharded import ShardedFactorizationMachine, lockstep_train_step
    rows = [1460, 583, 30001, 305, 24, 3, 10, 4, 18, 15, 105, 27, 2173]
    ref = ofm.State(rows, 16, 13, seed=5, opt=opt)
    single = FactorizationMachine(rows, 16, 13, device=dev, opt=opt, lr=1e-2)
    single.load_state(ref)
    shards = [ShardedFactorizationMachine(rows, 16, 13, device=dev, opt=opt, lr=1e-2, rank=r, world=N)
              for r in range(N)]
    for m in shards:
        m.load_state(ref)
    B = 512
    for step in range(3):
        parts = [ofm.synthetic_batch(B, rows, 13, seed=100 * step + r, dist='zipf') for r in range(N)]
        batches = [tuple(torch.from_numpy(a).to(dev) for a in p) for p in parts]
        whole = tuple(torch.from_numpy(np.concatenate([p[k] for p in parts])).to(dev) for k in range(3))
        l1 = float(single.train_step(*whole).item())
        lN = float(lockstep_train_step(shards, batches).item())
        assert abs(l1 - lN) <= 1e-5 * abs(l1)
        for m in shards:
            m.be.check()
    # reassemble the table from the shards: global row r = rank + N * local row
    V = np.zeros_like(ref.V)
    w1 = np.zeros_like(ref.w1)
    for r, m in enumerate(shards):
        v, w = m.be.rows()
        V[r::N], w1[r::N] = v.cpu().numpy(), w.cpu().numpy()
    tol = ADAM_TOL if opt == "adam_tf" else 1e-5      # Adam: see test_fm_train_step
    held("sharded FM N=%d %s V" % (N, opt), rel_err(V, single.V.cpu().numpy()), tol)
    held("sharded FM N=%d %s w1" % (N, opt), rel_err(w1, single.w1.cpu().numpy()), 1e-5)
    for m in shards:                              # dense side: replicas agree bit for bit
        assert torch.equal(m.dp, shards[0].dp)
    held("sharded FM N=%d %s dense" % (N, opt), rel_err(shards[0].dp.cpu().numpy(), single.dp.cpu().numpy()), 1e-5)


def test_sharded_equals_single_gpu_ragged_batch(dev):
    """A batch that fills neither the last 64-sample chunk nor a 256-slot unit, three ranks, SGD:
    the scalar slot pass and the chunk tails."""
    from recommendation_amd.fm import FactorizationMachine
    from recommendation_amd.sharded import ShardedFactorizationMachine, lockstep_train_step
    rows, N, B, opt = [1460, 5, 30001, 24, 3], 3, 300, "sgd"
    ref = ofm.State(rows, 16, 13, seed=2, opt=opt)
    single = FactorizationMachine(rows, 16, 13, device=dev, opt=opt, lr=1e-2)
    single.load_state(ref)
    shards = [ShardedFactorizationMachine(rows, 16, 13, device=dev, opt=opt, lr=1e-2, rank=r, world=N)
              for r in range(N)]
    for m in shards:
        m.load_state(ref)
    for step in range(3):
        parts = [ofm.synthetic_batch(B, rows, 13, seed=50 * step + r, dist='zipf') for r in range(N)]
        batches = [tuple(torch.from_numpy(a).to(dev) for a in p) for p in parts]
        whole = tuple(torch.from_numpy(np.concatenate([p[k] for p in parts])).to(dev) for k in range(3))
        l1 = float(single.train_step(*whole).item())
        lN = float(lockstep_train_step(shards, batches).item())
        assert abs(l1 - lN) <= 1e-5 * abs(l1)
        for m in shards:
            m.be.check()
    V, w1 = np.zeros_like(ref.V), np.zeros_like(ref.w1)
    for r, m in enumerate(shards):
        v, w = m.be.rows()
        V[r::N], w1[r::N] = v.cpu().numpy(), w.cpu().numpy()
    held("sharded FM N=3 sgd ragged V", rel_err(V, single.V.cpu().numpy()), 1e-5)
    held("sharded FM N=3 sgd ragged w1", rel_err(w1, single.w1.cpu().numpy()), 1e-5)
    for m in shards:
        assert torch.equal(m.dp, shards[0].dp)
    held("sharded FM N=3 sgd ragged dense", rel_err(shards[0].dp.cpu().numpy(), single.dp.cpu().numpy()), 1e-4)


def test_sharded_equals_single_gpu_full_batch_hot_rows(dev):
    """Batch 8192 per rank with one-row and three-row tables: an owner receives far more entries of
    one table than one LDS sort holds, and single keys with more than 8192 entries."""
    from recommendation_amd.fm import FactorizationMachine
    from recommendation_amd.sharded import ShardedFactorizationMachine, lockstep_train_step
    rows, N, B, opt = [1, 3, 50000, 300, 7], 2, 8192, "adagrad_rowwise"
    ref = ofm.State(rows, 16, 13, seed=9, opt=opt)
    single = FactorizationMachine(rows, 16, 13, device=dev, opt=opt, lr=1e-2)
    single.load_state(ref)
    shards = [ShardedFactorizationMachine(rows, 16, 13, device=dev, opt=opt, lr=1e-2, rank=r, world=N)
              for r in range(N)]
    for m in shards:
        m.load_state(ref)
    for step in range(2):
        parts = [ofm.synthetic_batch(B, rows, 13, seed=100 * step + r, dist='zipf') for r in range(N)]
        batches = [tuple(torch.from_numpy(a).to(dev) for a in p) for p in parts]
        whole = tuple(torch.from_numpy(np.concatenate([p[k] for p in parts])).to(dev) for k in range(3))
        l1 = float(single.train_step(*whole).item())
        lN = float(lockstep_train_step(shards, batches).item())
        assert abs(l1 - lN) <= 1e-5 * abs(l1)
        for m in shards:
            m.be.check()
    V = np.zeros_like(ref.V)
    for r, m in enumerate(shards):
        V[r::N] = m.be.rows()[0].cpu().numpy()
    # the hot rows sum 16384 terms in different (fixed) orders on the two paths
    held("sharded FM N=2 B=8192 hot rows V", rel_err(V, single.V.cpu().numpy()), 1e-4)


def test_routing_artefacts_bit_exact(dev):
    """Block o = chunk counts [ceil(B/64)][T] then slot[t][b] = global row // N when global row % N
    == o, else -1; ids outside their table route nowhere and raise the error word."""
    from recommendation_amd import ops
    rows = [7, 1000, 3, 50]
    toffs = np.concatenate([[0], np.cumsum(rows)]).astype(np.int64)
    B, T, N = 300, 4, 3
    rng = np.random.default_rng(0)
    ids = np.stack([rng.integers(0, r, size=B) for r in rows], axis=1).astype(np.int64)
    nc = -(-B // 64)
    assert ops.shard_block_ints(B, T) == nc * T + B * T
    for dtype in (np.int64, np.int32):
        send = torch.full((N, nc * T + T * B), -7, dtype=torch.int32, device=dev)
        err = ops.new_error_word(dev)
        ops.shard_route(torch.from_numpy(toffs).to(dev), torch.from_numpy(ids.astype(dtype)).to(dev), N, send, err)
        g = ids + toffs[:-1][None, :]
        got = send.cpu().numpy()
        for d in range(N):
            mine = g % N == d
            want = np.where(mine, g // N, -1).T.astype(np.int32)
            assert np.array_equal(got[d, nc * T:].reshape(T, B), want)
            cnt = np.stack([mine[64 * c:64 * c + 64].sum(0) for c in range(nc)])
            assert np.array_equal(got[d, :nc * T].reshape(nc, T), cnt)
        ops.check_error_word(err, "ids")
    bad = ids.copy()
    bad[5, 1] = 1000
    bad[6, 0] = -1
    send = torch.full((N, nc * T + T * B), -7, dtype=torch.int32, device=dev)
    err = ops.new_error_word(dev)
    ops.shard_route(torch.from_numpy(toffs).to(dev), torch.from_numpy(bad).to(dev), N, send, err)
    got = send.cpu().numpy()[:, nc * T:].reshape(N, T, B)
    assert (got[:, 1, 5] == -1).all() and (got[:, 0, 6] == -1).all()
    with pytest.raises(ops.InvalidArgumentError):
        ops.check_error_word(err, "ids")


def _owner_case(rows, N, rank, B, fill, seed):
    """recv[N][T][B] for one owner: table t's slots are valid with probability fill[t][0], drawn
    from its local range by fill[t][1] ('uniform', 'hot': 60 % on one key, 'block': half of them in
    16 neighbouring keys)."""
    rng = np.random.default_rng(seed)
    toffs = np.concatenate([[0], np.cumsum(rows)]).astype(np.int64)
    lo = [(int(toffs[t]) - rank + N - 1) // N for t in range(len(rows) + 1)]
    recv = np.full((N, len(rows), B), -1, np.int32)
    for t, (p, kind) in enumerate(fill):
        n_local = lo[t + 1] - lo[t]
        if n_local == 0:
            continue
        k = rng.integers(0, n_local, size=(N, B))
        if kind == 'hot':
            k = np.where(rng.random((N, B)) < 0.6, min(5, n_local - 1), k)
        elif kind == 'block':
            k = np.where(rng.random((N, B)) < 0.5, min(1000, n_local - 16) + rng.integers(0, 16, size=(N, B)), k)
        recv[:, t, :] = np.where(rng.random((N, B)) < p, lo[t] + k, -1)
    return toffs, lo, recv


def _blocks(recv):
    """slots [N][T][B] -> exchange blocks [N][blk] with their chunk counts in front"""
    N, T, B = recv.shape
    nc = -(-B // 64)
    cnt = np.stack([[(recv[s, :, 64 * c:64 * c + 64] >= 0).sum(1) for c in range(nc)] for s in range(N)])
    return np.concatenate([cnt.reshape(N, -1), recv.reshape(N, -1)], axis=1).astype(np.int32)


@pytest.mark.parametrize("case", ["uniform", "skewed", "ragged"])
def test_owner_forward_and_sort_bit_exact(dev, case):
    """The owner launch against numpy: per (source, sample) partial pools, and per table the stable
    sort of the valid slots with payload s*B + b, padding key behind them - including tables whose
    entries exceed one LDS sort (8192) in a key range, in an 8-bit bin, and in a single key."""
    from recommendation_amd import ops
    if case == "uniform":
        rows, N, rank, B = [1460, 583, 1000000, 305, 24, 3, 40000], 4, 2, 2048
        fill = [(0.25, 'uniform')] * len(rows)
    elif case == "ragged":                       # B not a multiple of 256: the scalar slot pass
        rows, N, rank, B = [1460, 3, 100000, 24, 40000], 3, 1, 300
        fill = [(0.4, 'uniform'), (1.0, 'uniform'), (0.3, 'block'), (0.5, 'hot'), (0.33, 'uniform')]
    else:
        rows, N, rank, B = [3, 70000, 2000000, 300, 1, 5, 8, 9], 4, 1, 8192
        fill = [(1.0, 'uniform'), (0.9, 'hot'), (1.0, 'block'), (0.5, 'uniform'), (1.0, 'uniform'), (0.0, 'uniform'),
                (0.45, 'hot'), (1.0, 'uniform')]
    T, D = len(rows), 16
    toffs, lo, recv = _owner_case(rows, N, rank, B, fill, seed=3)
    local_rows = (int(toffs[-1]) - rank + N - 1) // N
    assert lo[-1] == local_rows
    table = ops.FmRecords(local_rows, D, 1, dev)
    rng = np.random.default_rng(1)
    V = (rng.standard_normal((local_rows, D)) * 0.1).astype(np.float32)
    w1 = (rng.standard_normal(local_rows) * 0.1).astype(np.float32)
    table.rec[:, :D] = torch.from_numpy(V).to(dev)
    table.rec[:, table.off_w1] = torch.from_numpy(w1).to(dev)
    partial = torch.zeros((N, B, D + 2), device=dev)
    so = torch.full((T * N * B,), -3, dtype=torch.int32, device=dev)
    pe = torch.full((T * N * B,), -3, dtype=torch.int32, device=dev)
    n_total = torch.zeros(1, dtype=torch.int32, device=dev)
    sched = torch.tensor([0.0, 1.0, 1.0, 0.0], device=dev)
    err = ops.new_error_word(dev)
    ops.shard_owner_fwd_sort(table, local_rows, torch.from_numpy(toffs).to(dev), rank, N,
                             torch.from_numpy(_blocks(recv)).to(dev), B, partial, so, pe, n_total, sched, 1e-3, 0.9,
                             0.999, err)
    ops.check_error_word(err, "slots")
    so, pe = so.cpu().numpy(), pe.cpu().numpy()
    at = 0
    for t in range(T):
        slots = recv[:, t, :].reshape(-1)
        e = np.nonzero(slots >= 0)[0]
        order = np.argsort(slots[e], kind='stable')
        n = e.size
        assert np.array_equal(so[at:at + n], slots[e][order]), "table %d rows" % t
        assert np.array_equal(pe[at:at + n], e[order]), "table %d payload" % t
        at += n
    assert int(n_total.item()) == at
    assert (so[at:] == -3).all() and (pe[at:] == -3).all()           # nothing written past the entries
    valid = recv >= 0                                            # [N][T][B]
    idx = np.where(valid, recv, 0)
    Vg = np.where(valid[..., None], V[idx], 0).astype(np.float64)   # [N][T][B][D]
    want = np.concatenate([Vg.sum(1), (Vg * Vg).sum((1, 3))[..., None],
                           np.where(valid, w1[idx], 0).astype(np.float64).sum(1)[..., None]], axis=2)
    assert rel_err(partial.cpu().numpy(), want) <= 1e-5
    b1p, b2p = np.float32(0.9), np.float32(0.999)
    want_alpha = np.float32(1e-3) * np.sqrt(np.float32(1) - b2p) / (np.float32(1) - b1p)
    assert abs(float(sched[0]) - float(want_alpha)) <= 1e-9
    # a slot outside its table's local range is reported
    recv2 = recv.copy()
    recv2[0, 0, 0] = lo[1] + 1 if lo[2] > lo[1] + 1 else local_rows + 5
    ops.shard_owner_fwd_sort(table, local_rows, torch.from_numpy(toffs).to(dev), rank, N,
                             torch.from_numpy(_blocks(recv2)).to(dev), B, partial, None, None, None, None, 1e-3, 0.9,
                             0.999, err)
    with pytest.raises(ops.InvalidArgumentError):
        ops.check_error_word(err, "slots")


def test_train_step_over_rccl_world1_with_and_without_prefetch(dev):
    """The real train_step (torch.distributed, backend nccl = RCCL, one rank): routing the next
    batch on the side stream must not change a bit, and both must match the lockstep driver."""
    import os
    import torch.distributed as dist
    from recommendation_amd.sharded import ShardedFactorizationMachine, lockstep_train_step
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29611")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(dev))
    try:
        rows = [1460, 583, 30001, 305, 24, 3, 10, 4]
        ref = ofm.State(rows, 16, 13, seed=5, opt="adam_tf")
        models = [ShardedFactorizationMachine(rows, 16, 13, device=dev, opt="adam_tf", lr=1e-2) for _ in range(2)]
        lock = ShardedFactorizationMachine(rows, 16, 13, device=dev, opt="adam_tf", lr=1e-2, rank=0, world=1)
        for m in models + [lock]:
            m.load_state(ref)
        batches = [tuple(torch.from_numpy(a).to(dev) for a in ofm.synthetic_batch(1024, rows, 13, seed=s, dist='zipf'))
                   for s in range(4)]
        for i, b in enumerate(batches):
            nxt = batches[i + 1][0] if i + 1 < len(batches) else None
            l0 = models[0].train_step(*b)
            l1 = models[1].train_step(*b, next_ids=nxt)
            l2 = lockstep_train_step([lock], [b])
            assert torch.equal(l0, l1) and torch.equal(l0, l2)
        torch.cuda.synchronize()
        for m in models[1:] + [lock]:
            assert torch.equal(m.be.table.rec, models[0].be.table.rec)
            assert torch.equal(m.dp, models[0].dp)
            m.be.check()
    finally:
        dist.destroy_process_group()


def test_train_step_over_the_librarys_own_rccl_entry_points_world1(dev):
    """The same step with every collective going through the C ABI's RCCL entry points (mi_comm_init,
    mi_a2a_indices, mi_a2a_pooled, mi_allgather_bytes; RcclComm) instead of torch.distributed: one rank,
    bit-equal to the lockstep driver.  mi_allreduce_dense over one rank leaves the buffer as it is."""
    from recommendation_amd.sharded import RcclComm, ShardedFactorizationMachine, lockstep_train_step
    comm = RcclComm(rank=0, world=1)
    try:
        rows = [1460, 583, 30001, 305, 24, 3, 10, 4]
        ref = ofm.State(rows, 16, 13, seed=6, opt="adam_tf")
        m = ShardedFactorizationMachine(rows, 16, 13, device=dev, opt="adam_tf", lr=1e-2, comm=comm)
        lock = ShardedFactorizationMachine(rows, 16, 13, device=dev, opt="adam_tf", lr=1e-2, rank=0, world=1)
        for x in (m, lock):
            x.load_state(ref)
        for s in range(3):
            b = tuple(torch.from_numpy(a).to(dev) for a in ofm.synthetic_batch(1024, rows, 13, seed=10 + s, dist='zipf'))
            assert torch.equal(m.train_step(*b), lockstep_train_step([lock], [b]))
        torch.cuda.synchronize()
        assert torch.equal(m.be.table.rec, lock.be.table.rec) and torch.equal(m.dp, lock.dp)
        m.be.check()
        g = torch.arange(1000, dtype=torch.float32, device=dev) * 0.25
        want = g.clone()
        comm.all_reduce(g)
        torch.cuda.synchronize()
        assert torch.equal(g, want)
        ints = torch.arange(77, dtype=torch.int32, device=dev)
        out = torch.empty_like(ints)
        comm.all_to_all(out, ints)
        torch.cuda.synchronize()
        assert torch.equal(out, ints)
    finally:
        comm.close()
