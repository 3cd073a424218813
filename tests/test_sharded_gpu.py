"""Row-wise sharded FM: N ranks' models driven in lockstep on ONE GPU (each exchange a tensor
copy) must train exactly like the single-GPU FM on the concatenated batch."""
import numpy as np
import pytest
import torch

from oracle import fm as ofm

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
    tol = 1e-4 if opt == "adam_tf" else 1e-5      # Adam: see test_fm_train_step
    assert rel_err(V, single.V.cpu().numpy()) <= tol
    assert rel_err(w1, single.w1.cpu().numpy()) <= tol
    for m in shards:                              # dense side: replicas agree bit for bit
        assert torch.equal(m.dp, shards[0].dp)
    assert rel_err(shards[0].dp.cpu().numpy(), single.dp.cpu().numpy()) <= tol


def test_routing_artefacts_bit_exact(dev):
    """Destination = global row % N, local row = global row // N; one block per destination holding
    the per-sample offsets and then the rows in (sample, field) order."""
    from recommendation_amd import ops
    rows = [7, 1000, 3, 50]
    toffs = np.concatenate([[0], np.cumsum(rows)]).astype(np.int64)
    B, T, N = 300, 4, 3
    rng = np.random.default_rng(0)
    ids = np.stack([rng.integers(0, r, size=B) for r in rows], axis=1).astype(np.int64)
    blk = ops.shard_block_ints(B, T)
    send = torch.full((N, blk), -7, dtype=torch.int32, device=dev)
    ops.shard_route(torch.from_numpy(toffs).to(dev), torch.from_numpy(ids).to(dev), N, send)
    got = send.cpu().numpy()
    g = ids + toffs[:-1][None, :]
    for d in range(N):
        want_rows, want_off = [], np.zeros(B + 1, np.int32)
        for b in range(B):
            mine = [int(x) // N for x in g[b] if x % N == d]
            want_off[b + 1] = want_off[b] + len(mine)
            want_rows.extend(mine)
        assert np.array_equal(got[d, :B + 1], want_off)
        assert np.array_equal(got[d, B + 1:B + 1 + len(want_rows)], np.array(want_rows, np.int32))
        assert (got[d, B + 1 + len(want_rows):] == -7).all()       # nothing written past the rows
