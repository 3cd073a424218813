"""The entry point that allocates nothing (mi_embag_bwd_prepare_noalloc: sort scratch inside the
library's device image) against the workspace-taking mi_embag_bwd_prepare, through the C ABI.
Bit-exact (index work): sorted rows, permutation, distinct rows, run offsets, counts.  Two calls in
a row, the second with a smaller input, so stale scratch contents of the first would show."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _problem(rng, B, rows, L, pad_frac=0.05):
    T = len(rows)
    ids = np.stack([rng.integers(0, r, size=(B, L)) for r in rows], axis=1).astype(np.int64)
    ids[rng.random(ids.shape) < pad_frac] = -1
    toffs = np.concatenate([[0], np.cumsum(rows)]).astype(np.int64)
    return T, ids, toffs


def _both(dev, T, ids, toffs, L, runs):
    from recommendation_amd import ops
    total = int(toffs[-1])
    td, idd = torch.from_numpy(toffs).to(dev), torch.from_numpy(ids).to(dev)
    ref = ops.SortedPlan(td, T, total, idd, bag_len=L, runs=runs)          # mi_embag_bwd_prepare + workspace
    n = ids.size
    i32 = dict(dtype=torch.int32, device=dev)
    sr, pm = torch.full((n,), -7, **i32), torch.full((n,), -7, **i32)
    ur = torch.full((n,), -7, **i32) if runs else None
    so = torch.full((n + 1,), -7, **i32) if runs else None
    ct = torch.full((2,), -7, **i32) if runs else None
    torch.cuda.synchronize()
    before = torch.cuda.memory_stats(dev)["allocation.all.allocated"]
    ops.bwd_prepare_noalloc(td, T, total, idd, sr, pm, ur, so, ct, bag_len=L)
    torch.cuda.synchronize()
    assert torch.cuda.memory_stats(dev)["allocation.all.allocated"] == before     # torch allocated nothing either
    if runs:
        n_unique, n_valid = ref.counts.tolist()
        assert ct.tolist() == [n_unique, n_valid]
        assert torch.equal(sr[:n_valid], ref.sorted_rows[:n_valid]) and torch.equal(pm[:n_valid], ref.perm[:n_valid])
        assert torch.equal(sr, ref.sorted_rows)                                   # sentinels behind the valid part too
        assert torch.equal(ur[:n_unique], ref.unique_rows[:n_unique])
        assert torch.equal(so[:n_unique + 1], ref.seg_offsets[:n_unique + 1])
    else:
        assert torch.equal(sr, ref.sorted_rows) and torch.equal(pm, ref.perm)
    return sr, pm


def test_noalloc_prepare_equals_workspace_prepare_twice_second_smaller(dev):
    from oracle import embedding as oe
    rng = np.random.default_rng(77)
    rows = [3, 1000, 50, 100000, 10131227, 24]
    # first call: 6 x 40000 x 3 = 720000 entries (8 pairs per thread, three 9-bit passes: odd, starts
    # in the scratch's alternate buffers); with run arrays
    T, ids, toffs = _problem(rng, 40000, rows, 3)
    sr, pm = _both(dev, T, ids, toffs, 3, runs=True)
    want_rows, want_perm, _, _ = oe.prepare(toffs, T, ids, bag_len=3)
    nv = len(want_rows)
    assert np.array_equal(sr[:nv].cpu().numpy().astype(np.int64) & 0xFFFFFFFF, want_rows)
    assert np.array_equal(pm[:nv].cpu().numpy(), want_perm)
    # second call right behind it, smaller (6 x 9000 x 2 = 108000 entries, 4 pairs per thread: the
    # scratch is carved differently), run arrays again
    T, ids, toffs = _problem(rng, 9000, rows, 2)
    _both(dev, T, ids, toffs, 2, runs=True)
    # third: smaller still, without run arrays and few enough entries per table for the one-launch sort
    T, ids, toffs = _problem(rng, 700, rows, 1)
    _both(dev, T, ids, toffs, 1, runs=False)
    # and the global sort without run arrays
    T, ids, toffs = _problem(rng, 20000, rows, 1)
    _both(dev, T, ids, toffs, 1, runs=False)


def test_noalloc_prepare_rejects_sizes_beyond_its_scratch(dev):
    from recommendation_amd import _lib, ops
    n = _lib.MI_NOALLOC_MAX_ENTRIES + 1
    ids = torch.zeros((n, 1), dtype=torch.int32, device=dev)
    out = torch.empty((n,), dtype=torch.int32, device=dev)
    with pytest.raises(ops.MiError, match="built-in scratch"):
        ops.bwd_prepare_noalloc(ops.table_offsets([10], dev), 1, 10, ids, out, out.clone())
    # the limit itself is accepted
    ids = torch.randint(0, 10, (n - 1, 1), dtype=torch.int32, device=dev)
    sr, pm = torch.empty((n - 1,), dtype=torch.int32, device=dev), torch.empty((n - 1,), dtype=torch.int32, device=dev)
    ops.bwd_prepare_noalloc(ops.table_offsets([10], dev), 1, 10, ids, sr, pm)
    assert torch.equal(sr, torch.sort(ids.view(-1), stable=True)[0])


def test_noalloc_prepare_replays_from_a_captured_graph(dev):
    """After one call outside capture (here with no entries: it resolves the scratch address and launches nothing) the
    call can be captured in a HIP graph; a replay on new ids gives mi_embag_bwd_prepare's bits."""
    from recommendation_amd import ops
    rng = np.random.default_rng(78)
    rows = [3, 1000, 100000, 24]
    T, ids, toffs = _problem(rng, 5000, rows, 2)
    total = int(toffs[-1])
    td, idd = torch.from_numpy(toffs).to(dev), torch.from_numpy(ids).to(dev)
    n = ids.size
    i32 = dict(dtype=torch.int32, device=dev)
    sr, pm = torch.empty((n,), **i32), torch.empty((n,), **i32)
    ur, so, ct = torch.empty((n,), **i32), torch.empty((n + 1,), **i32), torch.empty((2,), **i32)
    ops.bwd_prepare_noalloc(td, T, total, idd[:0], sr, pm, ur, so, ct, bag_len=2)     # warm-up: no entries
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        ops.bwd_prepare_noalloc(td, T, total, idd, sr, pm, ur, so, ct, bag_len=2)
    for seed in (1, 2):
        _, ids2, _ = _problem(np.random.default_rng(seed), 5000, rows, 2)
        idd.copy_(torch.from_numpy(ids2))
        g.replay()
        torch.cuda.synchronize()
        ref = ops.SortedPlan(td, T, total, idd, bag_len=2, runs=True)
        n_unique, n_valid = ref.counts.tolist()
        assert ct.tolist() == [n_unique, n_valid]
        assert torch.equal(sr, ref.sorted_rows) and torch.equal(pm[:n_valid], ref.perm[:n_valid])
        assert torch.equal(ur[:n_unique], ref.unique_rows[:n_unique])
        assert torch.equal(so[:n_unique + 1], ref.seg_offsets[:n_unique + 1])
