"""Parity of every HIP kernel, called through the C ABI, against the numpy oracle.

Bars: bit-exact for ids, gathered rows, one-hot matrices and the sorted-run artefacts;
<= 1e-5 relative for fp32 sums (tolerance stated in each test)."""
import numpy as np
import pytest
import torch

from oracle import colour_model as cm
from oracle import embedding as oe
from oracle import fm as ofm
from tests.errlog import held

pytestmark = pytest.mark.gpu

RTOL = 1e-5
ADAM_TOL = 3e-5   # Adam-updated V against the fp64 oracle after the steps of test_fm_train_step (achieved: 1.5e-5; twice that)


def rel_err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def make_tables(rng, rows_per_table, dim):
    total = int(sum(rows_per_table))
    w = rng.standard_normal((total, dim)).astype(np.float32)
    toffs = np.concatenate([[0], np.cumsum(rows_per_table)]).astype(np.int64)
    return w, toffs


@pytest.mark.parametrize("dim", [1, 3, 4, 8, 16, 64, 128])
@pytest.mark.parametrize("idx_dtype", [np.int32, np.int64])
def test_embag_fwd_single_id_rows_bit_exact(dev, dim, idx_dtype):
    """One id per bag: the pooled output IS the gathered row, so it must be bit-exact."""
    from recommendation_amd import ops
    rng = np.random.default_rng(1)
    rows = [7, 1, 300, 1000, 3]
    w, toffs = make_tables(rng, rows, dim)
    B, T = 257, len(rows)
    ids = np.stack([rng.integers(0, r, size=B) for r in rows], axis=1).astype(idx_dtype)
    out = ops.embag_fwd(torch.from_numpy(w).to(dev), torch.from_numpy(toffs).to(dev), T,
                        torch.from_numpy(ids).to(dev), bag_len=1)
    expect = w[ids + toffs[:-1][None, :]].reshape(B, T * dim)
    assert np.array_equal(out.cpu().numpy(), expect)


@pytest.mark.parametrize("dim", [16, 128])
@pytest.mark.parametrize("idx_dtype", [np.int32, np.int64])
def test_embag_fwd_many_single_id_bags_bit_exact(dev, dim, idx_dtype):
    """From 65536 bags of one id on, a lane group serves four bags at once: the output must still be
    the gathered rows bit for bit (+ bias, fp32 and bf16), ids outside their table give zero rows and
    are reported, and a bag count that does not fill the last workgroup is handled."""
    from recommendation_amd import ops
    rng = np.random.default_rng(11)
    rows = [7, 1, 3000, 1000, 3]
    w, toffs = make_tables(rng, rows, dim)
    B, T = 13109, len(rows)                      # 65545 bags
    ids = np.stack([rng.integers(0, r, size=B) for r in rows], axis=1).astype(idx_dtype)
    wd, td = torch.from_numpy(w).to(dev), torch.from_numpy(toffs).to(dev)
    out = ops.embag_fwd(wd, td, T, torch.from_numpy(ids).to(dev), bag_len=1)
    expect = w[ids + toffs[:-1][None, :]].reshape(B, T * dim)
    assert np.array_equal(out.cpu().numpy(), expect)
    bias = rng.standard_normal(T * dim).astype(np.float32)
    outb = ops.embag_fwd(wd, td, T, torch.from_numpy(ids).to(dev), bag_len=1, bias=torch.from_numpy(bias).to(dev))
    assert np.array_equal(outb.cpu().numpy(), expect + bias[None, :])
    out16 = ops.embag_fwd(wd, td, T, torch.from_numpy(ids).to(dev), bag_len=1, out_dtype=torch.bfloat16)
    assert torch.equal(out16, out.to(torch.bfloat16))
    bad = ids.copy()
    bad[5, 2], bad[B - 1, 4], bad[77, 0] = rows[2], 1 << 20, -1
    err = ops.new_error_word(dev)
    outx = ops.embag_fwd(wd, td, T, torch.from_numpy(bad).to(dev), bag_len=1, check_flags=ops.MI_CHECK_RANGE, err=err)
    ex = expect.copy().reshape(B, T, dim)
    ex[5, 2], ex[B - 1, 4], ex[77, 0] = 0, 0, 0
    assert np.array_equal(outx.cpu().numpy(), ex.reshape(B, T * dim))
    with pytest.raises(ops.InvalidArgumentError):
        ops.check_error_word(err, "ids")


@pytest.mark.parametrize("mode", ["sum", "mean"])
@pytest.mark.parametrize("dim", [4, 8, 16, 5])
def test_embag_fwd_fixed_bags_with_padding(dev, mode, dim):
    from recommendation_amd import ops
    rng = np.random.default_rng(2)
    rows = [128, 20]
    w, toffs = make_tables(rng, rows, dim)
    B, T, L = 300, 2, 7
    ids = np.stack([rng.integers(-1, rows[t], size=(B, L)) for t in range(T)], axis=1).astype(np.int32)  # [B,T,L]
    bias = rng.standard_normal(T * dim).astype(np.float32)
    out = ops.embag_fwd(torch.from_numpy(w).to(dev), torch.from_numpy(toffs).to(dev), T,
                        torch.from_numpy(ids).to(dev), bag_len=L, bias=torch.from_numpy(bias).to(dev),
                        pool_mode=ops.MI_POOL_MEAN if mode == "mean" else ops.MI_POOL_SUM)
    expect = oe.embag_fwd(w, toffs, T, ids, bag_len=L, mode=mode, bias=bias)
    assert rel_err(out.cpu().numpy(), expect) <= RTOL


def test_embag_fwd_csr_ragged_and_empty_bags(dev):
    from recommendation_amd import ops
    rng = np.random.default_rng(3)
    rows = [50, 1000, 9]
    dim = 16
    w, toffs = make_tables(rng, rows, dim)
    B, T = 64, 3
    lens = rng.integers(0, 12, size=B * T)
    lens[::5] = 0          # empty bags
    lens[7] = 100          # one long bag
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    ids = np.concatenate([rng.integers(0, rows[i % T], size=lens[i]) for i in range(B * T)]).astype(np.int64)
    out = ops.embag_fwd(torch.from_numpy(w).to(dev), torch.from_numpy(toffs).to(dev), T,
                        torch.from_numpy(ids).to(dev), offsets=torch.from_numpy(offsets).to(dev))
    expect = oe.embag_fwd(w, toffs, T, ids, offsets=offsets)
    assert rel_err(out.cpu().numpy(), expect) <= RTOL
    # bf16 output: one rounding of the fp32 pooled sum
    out16 = ops.embag_fwd(torch.from_numpy(w).to(dev), torch.from_numpy(toffs).to(dev), T,
                          torch.from_numpy(ids).to(dev), offsets=torch.from_numpy(offsets).to(dev),
                          out_dtype=torch.bfloat16)
    assert torch.equal(out16, out.to(torch.bfloat16))


def test_empty_batch_is_a_no_op(dev):
    from recommendation_amd import ops
    w = torch.zeros((4, 8), device=dev)
    toffs = ops.table_offsets([4], dev)
    out = ops.embag_fwd(w, toffs, 1, torch.zeros((0, 1), dtype=torch.int32, device=dev), bag_len=1)
    assert out.shape == (0, 8)


@pytest.mark.parametrize("dtype", [torch.int32, torch.int64])
def test_id_columns_checked_and_concatenated_in_one_launch_each(dev, dtype):
    """mi_validate_id_columns == mi_validate_ids column by column (same error bits, one bad column at a time);
    mi_concat_id_columns == the shifted-id concatenation of the one-hot widths, negative ids kept."""
    from recommendation_amd import ops
    rng = np.random.default_rng(11)
    B = 777
    R, NN, RI = ops.MI_CHECK_RANGE, ops.MI_CHECK_NONNEG, ops.MI_CHECK_INCREASING

    def fresh():
        user = rng.integers(0, 100, (B, 1))
        rec = np.sort(rng.permuted(np.tile(np.arange(128), (B, 1)), axis=1)[:, :6], axis=1)
        click = rng.integers(-1, 128, (B, 1))
        return [user, rec, click]

    specs = [(100, R | NN), (128, R | NN | RI), (128, R)]

    def bits(cols, fused):
        err = ops.new_error_word(dev)
        ts = [(torch.from_numpy(c).to(dev).to(dtype), size, fl) for c, (size, fl) in zip(cols, specs)]
        if fused:
            ops.validate_id_columns(ts, err)
        else:
            for t, size, fl in ts:
                ops.validate_ids(ops.table_offsets([size], dev), 1, t, fl, err)
        return int(err.item())

    good = fresh()
    assert bits(good, True) == 0 == bits(good, False)
    cases = []
    c = fresh(); c[0][5, 0] = 100; cases.append(c)          # user out of range
    c = fresh(); c[0][0, 0] = -1; cases.append(c)           # one-hot must be non-negative
    c = fresh(); c[1][B - 1, 3] = c[1][B - 1, 2]; cases.append(c)   # repeated id in an n-hot row
    c = fresh(); c[1][7, :2] = c[1][7, :2][::-1]; cases.append(c)   # disorder
    c = fresh(); c[2][B // 2, 0] = 128; cases.append(c)     # padding column: -1 fine, 128 not
    c = fresh(); c[2][3, 0] = -7; cases.append(c)           # any negative id is "no entry" there
    want = [ops.MI_IDERR_RANGE, ops.MI_IDERR_RANGE, ops.MI_IDERR_ORDER, ops.MI_IDERR_ORDER, ops.MI_IDERR_RANGE, 0]
    for c, w in zip(cases, want):
        assert bits(c, True) == bits(c, False) == w

    parts = [(torch.from_numpy(c).to(dev).to(dtype), size) for c, (size, _) in zip(good, specs)]
    parts[2][0][4, 0] = -1
    got = ops.concat_id_columns(parts)
    base, cols = 0, []
    for t, size in parts:
        cols.append(torch.where(t >= 0, t + base, t))
        base += size
    assert got.dtype == dtype and torch.equal(got, torch.cat(cols, dim=1))
    with pytest.raises(ops.MiError):
        ops.concat_id_columns([(parts[0][0], 100), (parts[1][0].to(torch.int64 if dtype == torch.int32 else torch.int32), 128)])


def test_validation_reports_and_never_dereferences_bad_ids(dev):
    from recommendation_amd import ops
    w = torch.ones((10, 4), device=dev)
    toffs = ops.table_offsets([10], dev)
    err = ops.new_error_word(dev)
    ids = torch.tensor([[3], [10], [2 ** 31 - 1], [-5]], dtype=torch.int32, device=dev)
    out = ops.embag_fwd(w, toffs, 1, ids, bag_len=1, check_flags=ops.MI_CHECK_RANGE, err=err)
    assert out.cpu().numpy().tolist() == [[1.0] * 4, [0.0] * 4, [0.0] * 4, [0.0] * 4]
    with pytest.raises(ops.InvalidArgumentError):
        ops.check_error_word(err)
    # negative id is padding unless NONNEG is asked for
    ops.validate_ids(toffs, 1, torch.tensor([[-1]], dtype=torch.int32, device=dev), ops.MI_CHECK_RANGE, err)
    ops.check_error_word(err)
    ops.validate_ids(toffs, 1, torch.tensor([[-1]], dtype=torch.int32, device=dev),
                     ops.MI_CHECK_RANGE | ops.MI_CHECK_NONNEG, err)
    with pytest.raises(ops.InvalidArgumentError):
        ops.check_error_word(err)
    # strictly increasing rows (tf.sparse_to_dense validate_indices): repeat and disorder both fail
    for bad in ([[1, 1, 2]], [[2, 0, 5]]):
        ops.validate_ids(toffs, 1, torch.tensor(bad, dtype=torch.int32, device=dev),
                         ops.MI_CHECK_RANGE | ops.MI_CHECK_INCREASING, err)
        with pytest.raises(ops.InvalidArgumentError):
            ops.check_error_word(err)
    ops.validate_ids(toffs, 1, torch.tensor([[0, 2, 5]], dtype=torch.int32, device=dev),
                     ops.MI_CHECK_RANGE | ops.MI_CHECK_INCREASING, err)
    ops.check_error_word(err)


def test_onehot_dense_matches_reference_encoders_bit_exact(dev):
    from recommendation_amd import ops
    batch = cm.synthetic_batch(300)
    u = ops.onehot_dense(torch.from_numpy(batch['user']).to(dev), 100)
    assert np.array_equal(u.cpu().numpy(), cm.input_to_one_hot(batch['user'], 100))
    r = ops.onehot_dense(torch.from_numpy(batch['recommend_colours_1']).to(dev), 128)
    assert np.array_equal(r.cpu().numpy(), cm.input_to_n_hot(batch['recommend_colours_1'], 128, 6))
    c = ops.onehot_dense(torch.from_numpy(batch['click_colour_1']).to(dev), 128)
    assert np.array_equal(c.cpu().numpy(), cm.input_to_one_hot_plus(batch['click_colour_1'], 128))


@pytest.mark.parametrize("n,bits", [(1, 32), (2, 32), (63, 8), (2048, 32), (2049, 16), (100000, 32), (300007, 26)])
def test_radix_sort_is_stable_and_exact(dev, n, bits):
    from recommendation_amd import ops
    rng = np.random.default_rng(n)
    hi = (1 << bits) - 1
    keys = rng.integers(0, min(hi, 5000) + 1, size=n, dtype=np.uint64).astype(np.uint32)  # many duplicates
    if n > 10:
        keys[rng.integers(0, n, size=n // 3)] = rng.integers(0, hi + 1, size=n // 3, dtype=np.uint64).astype(np.uint32)
    vals = np.arange(n, dtype=np.uint32)
    k = torch.from_numpy(keys.view(np.int32).copy()).to(dev)
    v = torch.from_numpy(vals.view(np.int32).copy()).to(dev)
    ops.sort_pairs_u32(k, v, bits)
    order = np.argsort(keys, kind='stable')
    assert np.array_equal(k.cpu().numpy().view(np.uint32), keys[order])
    assert np.array_equal(v.cpu().numpy().view(np.uint32), vals[order])


@pytest.mark.parametrize("layout", ["fixed1", "fixed7", "csr"])
def test_bwd_prepare_artefacts_bit_exact(dev, layout):
    from recommendation_amd import ops
    rng = np.random.default_rng(5)
    rows = [3, 1000, 50, 100000]
    T = len(rows)
    toffs = np.concatenate([[0], np.cumsum(rows)]).astype(np.int64)
    B = 1000
    offsets = None
    if layout == "fixed1":
        ids = np.stack([rng.integers(0, r, size=B) for r in rows], axis=1).astype(np.int64)
        L = 1
    elif layout == "fixed7":
        L = 7
        ids = np.stack([rng.integers(-1, r, size=(B, L)) for r in rows], axis=1).astype(np.int32)
        ids[3, 1, 2] = 5000  # out of range: excluded like padding
    else:
        L = None
        lens = rng.integers(0, 9, size=B * T)
        offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        ids = np.concatenate([rng.integers(0, rows[i % T], size=lens[i]) for i in range(B * T)]).astype(np.int32)
    plan = ops.SortedPlan(torch.from_numpy(toffs).to(dev), T, int(toffs[-1]), torch.from_numpy(ids).to(dev),
                          offsets=None if offsets is None else torch.from_numpy(offsets).to(dev), bag_len=L)
    got = plan.host()
    want = oe.prepare(toffs, T, ids, offsets=offsets, bag_len=L)
    for g, w, name in zip(got, want, ("sorted_rows", "perm", "unique_rows", "seg_offsets")):
        assert np.array_equal(g, w), name


def _bag_problem(rng, dim, mode):
    rows = [3, 128, 5000]
    T, B, L = 3, 700, 5
    w, toffs = make_tables(rng, rows, dim)
    ids = np.stack([rng.integers(-1, r, size=(B, L)) for r in rows], axis=1).astype(np.int32)
    grad = rng.standard_normal((B, T * dim)).astype(np.float32)
    return rows, T, B, L, w, toffs, ids, grad


@pytest.mark.parametrize("mode", ["sum", "mean"])
@pytest.mark.parametrize("dim", [4, 16, 6])
def test_dense_grad_equals_onehot_transpose_times_dy(dev, mode, dim):
    """dW must equal X^T dY with X the materialised n-hot matrix (TensorFlow's MatMul gradient)."""
    from recommendation_amd import ops
    rng = np.random.default_rng(6)
    rows, T, B, L, w, toffs, ids, grad = _bag_problem(rng, dim, mode)
    pm = ops.MI_POOL_MEAN if mode == "mean" else ops.MI_POOL_SUM
    plan = ops.SortedPlan(torch.from_numpy(toffs).to(dev), T, int(toffs[-1]), torch.from_numpy(ids).to(dev),
                          bag_len=L, pool_mode=pm)
    dW = ops.bwd_dense_grad(plan, torch.from_numpy(grad).to(dev), dim, int(toffs[-1]))
    expect = np.zeros((int(toffs[-1]), dim))
    for t in range(T):
        X = np.zeros((B, rows[t]))
        for b in range(B):
            for v in ids[b, t]:
                if v >= 0:
                    X[b, v] += 1.0      # duplicates inside a bag count twice (bag semantics)
        if mode == "mean":
            X /= np.maximum(X.sum(axis=1, keepdims=True), 1.0)
        expect[toffs[t]:toffs[t + 1]] = X.T @ grad[:, t * dim:(t + 1) * dim].astype(np.float64)
    assert rel_err(dW.cpu().numpy(), expect) <= RTOL


@pytest.mark.parametrize("mode", ["sum", "mean"])
@pytest.mark.parametrize("dim,L,idt", [(16, 64, np.int32), (64, 37, np.int64), (128, 64, np.int64), (128, 100, np.int32),
                                       (32, 130, np.int64), (8, 150, np.int32), (16, 20, np.int64)])
def test_wave_per_bag_forward_equals_oracle(dev, mode, dim, L, idt):
    """Bags of 16 ids or more in a small batch take the one-wave-per-bag kernel (64 ids per coalesced load, rows staged
    in LDS, shuffle tree over the lane groups): pooled rows against the oracle, padding ids, a bias, bf16 output."""
    from recommendation_amd import ops
    rng = np.random.default_rng(70 + L)
    rows = [500, 7, 3000]
    T, B = len(rows), 41
    toffs = np.concatenate([[0], np.cumsum(rows)]).astype(np.int64)
    w = rng.standard_normal((int(toffs[-1]), dim)).astype(np.float32)
    ids = np.stack([rng.integers(-1, r, size=(B, L)) for r in rows], axis=1).astype(idt)
    ids[3, 1, :] = -1                                       # a bag of padding only
    bias = rng.standard_normal((T, dim)).astype(np.float32)
    pm = ops.MI_POOL_MEAN if mode == "mean" else ops.MI_POOL_SUM
    err = ops.new_error_word(dev)
    got = ops.embag_fwd(torch.from_numpy(w).to(dev), torch.from_numpy(toffs).to(dev), T, torch.from_numpy(ids).to(dev),
                        bag_len=L, pool_mode=pm, check_flags=ops.MI_CHECK_RANGE, bias=torch.from_numpy(bias).to(dev), err=err)
    want = oe.embag_fwd(w, toffs, T, ids, bag_len=L, mode=mode) + bias.reshape(1, -1)
    assert rel_err(got.cpu().numpy(), want) <= RTOL
    ops.check_error_word(err)
    got16 = ops.embag_fwd(torch.from_numpy(w).to(dev), torch.from_numpy(toffs).to(dev), T, torch.from_numpy(ids).to(dev),
                          bag_len=L, pool_mode=pm, out_dtype=torch.bfloat16)
    assert rel_err(got16.float().cpu().numpy(), want - bias.reshape(1, -1)) <= 1e-2


def test_wave_per_bag_forward_reports_bad_ids(dev):
    """Range and strictly-increasing checks of the wave kernel, also across its 64-id chunks (an order violation
    between ids 63 and 64 of a bag, padding ids in between skipped as the reference's validate_indices would)."""
    from recommendation_amd import ops
    toffs = torch.tensor([0, 1000], dtype=torch.int64, device=dev)
    w = torch.zeros((1000, 16), dtype=torch.float32, device=dev)
    L = 100
    good = np.sort(np.random.default_rng(3).choice(1000, size=(5, L), replace=False), axis=1).astype(np.int32)
    flags = ops.MI_CHECK_RANGE | ops.MI_CHECK_INCREASING

    def run(ids):
        err = ops.new_error_word(dev)
        ops.embag_fwd(w, toffs, 1, torch.from_numpy(ids).to(dev), bag_len=L, check_flags=flags, err=err)
        ops.check_error_word(err)

    run(good)
    padded = good.copy()
    padded[2, 10:20] = -1                                   # padding inside a bag does not break the order
    run(padded)
    for pos in (1, 63, 64, 99):
        bad = good.copy()
        bad[4, pos] = bad[4, pos - 1]                       # a repeated id
        with pytest.raises(ops.InvalidArgumentError):
            run(bad)
    across = good.copy()
    across[1, 60:64] = -1                                   # the neighbour across the chunk border is padding:
    across[1, 64] = across[1, 59]                           # the last real id before it still counts
    with pytest.raises(ops.InvalidArgumentError):
        run(across)
    out_of_range = good.copy()
    out_of_range[0, 99] = 1000
    with pytest.raises(ops.InvalidArgumentError):
        run(out_of_range)


@pytest.mark.parametrize("csr", [False, True])
def test_bag_info_gives_the_bits_of_per_entry_recounting(dev, csr):
    """mi_embag_bag_info's per-bag scale and per-position bag (what SortedPlan hands the apply calls) against the
    numpy count, and the mean-pooled dense gradient with them == without them (each entry recounting its bag's
    ids and searching the CSR offsets), bit for bit."""
    from recommendation_amd import ops
    rng = np.random.default_rng(61)
    dim, T, B = 16, 3, 37
    rows = [11, 200, 5]
    toffs = np.concatenate([[0], np.cumsum(rows)]).astype(np.int64)
    if csr:
        lens = rng.integers(0, 150, size=B * T)
        lens[5] = 0
        offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        ids = np.concatenate([rng.integers(-1, rows[i % T], size=l) for i, l in enumerate(lens)]).astype(np.int64)
        kw = dict(offsets=torch.from_numpy(offsets).to(dev))
    else:
        L = 70
        lens = np.full(B * T, L)
        offsets = np.arange(B * T + 1, dtype=np.int64) * L
        ids = np.stack([rng.integers(-1, r, size=(B, L)) for r in rows], axis=1).astype(np.int64)
        kw = dict(bag_len=L)
    grad = rng.standard_normal((B, T * dim)).astype(np.float32)
    plan = ops.SortedPlan(torch.from_numpy(toffs).to(dev), T, int(toffs[-1]), torch.from_numpy(ids).to(dev),
                          pool_mode=ops.MI_POOL_MEAN, **kw)
    flat = ids.reshape(-1)
    cnt = np.array([np.count_nonzero(flat[offsets[i]:offsets[i + 1]] >= 0) for i in range(B * T)])
    want_scale = np.where(cnt > 1, np.float32(1.0) / np.maximum(cnt, 1).astype(np.float32), np.float32(1.0))
    assert np.array_equal(plan.bag_scale.cpu().numpy(), want_scale.astype(np.float32))
    if csr:
        assert np.array_equal(plan.entry_bag.cpu().numpy()[:len(flat)], np.repeat(np.arange(B * T), lens))
    g = torch.from_numpy(grad).to(dev)
    with_info = ops.bwd_dense_grad(plan, g, dim, int(toffs[-1])).clone()
    plan.bag_scale, plan.entry_bag = None, None
    plan._bind()
    without = ops.bwd_dense_grad(plan, g, dim, int(toffs[-1]))
    assert torch.equal(with_info, without)
    assert float(with_info.abs().sum()) > 0


@pytest.mark.parametrize("opt", ["sgd", "adagrad_rowwise", "adam_tf", "adam_dense_tf"])
def test_fused_sparse_updates(dev, opt):
    """The fused kernels = (run sums) + (optimizer formula).  The run sums are held to 1e-5
    against the oracle by the dense-gradient test above; here the oracle's optimizer formulas
    are fed the device's own run sums (same kernel template, same fixed order, so bitwise the
    sums the fused kernel forms) and every updated buffer must agree to 1e-6: Adam divides by
    sqrt(v)+eps and v is quadratic in the sum, so comparing through an independently rounded
    fp64 sum would test fp32 conditioning, not the kernel.  That independent comparison is made as
    well - the oracle's OWN fp64 run sums through the same formulas, every buffer to 1e-5 (achieved
    errors are printed) - so that no part of the fused kernel is only compared with itself."""
    from recommendation_amd import ops
    rng = np.random.default_rng(7)
    dim = 16
    rows, T, B, L, w, toffs, ids, grad = _bag_problem(rng, dim, "sum")
    ids[:, 0, :] = rng.integers(0, 3, size=(B, L))   # table of 3 rows: runs of ~1100 entries (long path)
    total = int(toffs[-1])
    plan = ops.SortedPlan(torch.from_numpy(toffs).to(dev), T, total, torch.from_numpy(ids).to(dev), bag_len=L)
    g = torch.from_numpy(grad).to(dev)
    wd = torch.from_numpy(w.copy()).to(dev)
    dW = ops.bwd_dense_grad(plan, g, dim, total).cpu().numpy().astype(np.float64)
    touched = np.unique(oe.prepare(toffs, T, ids, bag_len=L)[2])
    grads = {int(r): dW[int(r)] for r in touched}
    want = oe.row_grads(toffs, T, ids, grad, dim, bag_len=L)
    assert sorted(want) == sorted(grads)
    wr = w.astype(np.float64)
    wi = w.astype(np.float64)                      # the independent evaluation: the oracle's own sums
    lr, b1, b2, eps = 0.05, 0.9, 0.999, 1e-8
    TOL = 2e-6
    if opt == "sgd":
        ops.bwd_apply_sgd(plan, g, wd, lr)
        oe.apply_sgd(wr, grads, lr)
        oe.apply_sgd(wi, want, lr)
    elif opt == "adagrad_rowwise":
        st = rng.random(total).astype(np.float32)
        std = torch.from_numpy(st.copy()).to(dev)
        ops.bwd_apply_adagrad_rowwise(plan, g, wd, std, lr, eps)
        sr, si = st.astype(np.float64), st.astype(np.float64)
        oe.apply_adagrad_rowwise(wr, sr, grads, lr, eps)
        oe.apply_adagrad_rowwise(wi, si, want, lr, eps)
        assert rel_err(std.cpu().numpy(), sr) <= TOL
        held("fused %s state vs the oracle's own sums" % opt, rel_err(std.cpu().numpy(), si), 1e-5)
    else:
        m0 = (rng.standard_normal((total, dim)) * 0.1).astype(np.float32)
        v0 = (rng.random((total, dim)) * 0.1).astype(np.float32)
        md, vd = torch.from_numpy(m0.copy()).to(dev), torch.from_numpy(v0.copy()).to(dev)
        lr_t = float(oe.adam_lr_t(lr, b1, b2, 3))
        mr, vr = m0.astype(np.float64), v0.astype(np.float64)
        mi, vi = m0.astype(np.float64), v0.astype(np.float64)
        if opt == "adam_tf":
            ops.bwd_apply_adam_tf(plan, g, wd, md, vd, lr_t, b1, b2, eps)
            oe.apply_adam_tf(wr, mr, vr, grads, lr_t, b1, b2, eps)
            oe.apply_adam_tf(wi, mi, vi, want, lr_t, b1, b2, eps)
        else:
            ops.bwd_apply_adam_dense_tf(plan, g, wd, md, vd, lr_t, b1, b2, eps)
            oe.apply_adam_tf(wr, mr, vr, grads, lr_t, b1, b2, eps, dense=True)
            oe.apply_adam_tf(wi, mi, vi, want, lr_t, b1, b2, eps, dense=True)
        assert rel_err(md.cpu().numpy(), mr) <= TOL
        assert rel_err(vd.cpu().numpy(), vr) <= TOL
        held("fused %s m vs the oracle's own sums" % opt, rel_err(md.cpu().numpy(), mi), 1e-5)
        held("fused %s v vs the oracle's own sums" % opt, rel_err(vd.cpu().numpy(), vi), 1e-5)
    assert rel_err(wd.cpu().numpy(), wr) <= TOL
    held("fused %s weights vs the oracle's own sums" % opt, rel_err(wd.cpu().numpy(), wi), 1e-5)


@pytest.mark.parametrize("dim", [16, 128])
def test_sparse_update_waves_of_distinct_rows(dev, dim):
    """Tables much larger than the batch: most waves of the run-finding kernel hold 64 different rows and
    take four of them per lane group at once.  Table 0's ids are all distinct (every wave of it is such a
    wave), table 1's repeat (the general path), and the boundary waves mix both.  A one-entry run's
    gradient is the entry's own gradient row, so the dense gradient is exact; SGD and row-wise Adagrad
    are held to the oracle's formulas."""
    from recommendation_amd import ops
    rng = np.random.default_rng(31)
    rows = [300000, 200000]
    T, B = 2, 4096 + 37
    toffs = np.concatenate([[0], np.cumsum(rows)]).astype(np.int64)
    total = int(toffs[-1])
    ids = np.stack([rng.permutation(rows[0])[:B], rng.integers(0, 900, size=B)], axis=1).astype(np.int64)[:, :, None]
    grad = rng.standard_normal((B, T * dim)).astype(np.float32)
    w = (rng.standard_normal((total, dim)) * 0.1).astype(np.float32)
    plan = ops.SortedPlan(torch.from_numpy(toffs).to(dev), T, total, torch.from_numpy(ids).to(dev), bag_len=1,
                          runs=False)
    g = torch.from_numpy(grad).to(dev)
    dW = ops.bwd_dense_grad(plan, g, dim, total).cpu().numpy()
    assert np.array_equal(dW[ids[:, 0, 0]], grad[:, :dim] + 0.0)          # table 0: exact
    want = oe.row_grads(toffs, T, ids, grad, dim, bag_len=1)
    dense = np.zeros((total, dim), np.float64)
    for r, v in want.items():
        dense[r] = v
    assert rel_err(dW, dense) <= RTOL
    grads = {int(r): dW[int(r)].astype(np.float64) for r in want}
    lr, eps, TOL = 0.05, 1e-8, 2e-6
    wd = torch.from_numpy(w.copy()).to(dev)
    ops.bwd_apply_sgd(plan, g, wd, lr)
    wr = w.astype(np.float64)
    oe.apply_sgd(wr, grads, lr)
    assert rel_err(wd.cpu().numpy(), wr) <= TOL
    st = rng.random(total).astype(np.float32)
    wd, std = torch.from_numpy(w.copy()).to(dev), torch.from_numpy(st.copy()).to(dev)
    ops.bwd_apply_adagrad_rowwise(plan, g, wd, std, lr, eps)
    wr, sr = w.astype(np.float64), st.astype(np.float64)
    oe.apply_adagrad_rowwise(wr, sr, grads, lr, eps)
    assert rel_err(std.cpu().numpy(), sr) <= TOL and rel_err(wd.cpu().numpy(), wr) <= TOL
    # untouched rows keep their bits
    mask = np.ones(total, bool)
    mask[list(want)] = False
    assert np.array_equal(wd.cpu().numpy()[mask], w[mask])


def test_sparse_update_is_bitwise_reproducible(dev):
    from recommendation_amd import ops
    rng = np.random.default_rng(8)
    dim = 16
    rows, T, B, L, w, toffs, ids, grad = _bag_problem(rng, dim, "sum")
    ids[:, 0, :] = rng.integers(0, 3, size=(B, L))
    outs = []
    for _ in range(3):
        plan = ops.SortedPlan(torch.from_numpy(toffs).to(dev), T, int(toffs[-1]), torch.from_numpy(ids).to(dev),
                              bag_len=L)
        outs.append(ops.bwd_dense_grad(plan, torch.from_numpy(grad).to(dev), dim, int(toffs[-1])).cpu().numpy())
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])


@pytest.mark.parametrize("B,K,N", [(300, 8, 3), (300, 18, 8), (300, 4, 128), (1, 3, 3), (1000, 256, 8)])
def test_linear_fwd_bwd(dev, B, K, N):
    from recommendation_amd import ops
    rng = np.random.default_rng(9)
    x = rng.standard_normal((B, K)).astype(np.float32)
    W = rng.standard_normal((K, N)).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32)
    dy = rng.standard_normal((B, N)).astype(np.float32)
    xd, Wd, bd, dyd = (torch.from_numpy(a).to(dev) for a in (x, W, b, dy))
    y = ops.linear_fwd(xd, Wd, bd)
    x64, W64, dy64 = x.astype(np.float64), W.astype(np.float64), dy.astype(np.float64)
    assert rel_err(y.cpu().numpy(), x64 @ W64 + b) <= RTOL
    dx, dW, db = ops.linear_bwd(xd, Wd, dyd)
    assert rel_err(dx.cpu().numpy(), dy64 @ W64.T) <= RTOL
    assert rel_err(dW.cpu().numpy(), x64.T @ dy64) <= RTOL
    assert rel_err(db.cpu().numpy(), dy64.sum(0)) <= RTOL
    assert rel_err(ops.colsum(dyd).cpu().numpy(), dy64.sum(0)) <= RTOL


def test_softmax_heads_against_reference_heads(dev):
    from recommendation_amd import ops
    rng = np.random.default_rng(10)
    B, N, k = 300, 128, 6
    z = (rng.standard_normal((B, N)) * 3).astype(np.float32)
    z[0, :] = 0.0                      # full tie: top-k must be 0..5
    z[1, 5] = z[1, 77] = 9.0           # two-way tie at the top: lower index first
    lab = rng.integers(-1, N, size=B).astype(np.int32)
    ydense = cm.input_to_one_hot_plus(lab, N)
    r = ops.softmax_heads(torch.from_numpy(z).to(dev), k=k, labels=torch.from_numpy(ydense).to(dev),
                          label_ids=torch.from_numpy(lab).to(dev), grad_scale=1.0 / B, want_probs=True,
                          want_top=True, want_loss=True, want_dlogits=True, want_hits=True)
    z64 = z.astype(np.float64)
    assert rel_err(r['probs'].cpu().numpy(), cm.softmax(z64)) <= RTOL
    idx = r['top_idx'].cpu().numpy()
    assert idx[0].tolist() == [0, 1, 2, 3, 4, 5]
    assert idx[1, :2].tolist() == [5, 77]
    # the device's own probabilities, ranked by the reference rule, must give the device's indices
    _, want_idx = cm.top_k(r['probs'].cpu().numpy(), k)
    assert np.array_equal(idx, want_idx)
    per_row = -(ydense * cm.log_softmax(z64)).sum(axis=1)
    assert rel_err(r['loss_rows'].cpu().numpy(), per_row) <= RTOL
    assert rel_err(r['dlogits'].cpu().numpy(), cm.loss_grad_logits(z64, ydense.astype(np.float64))) <= RTOL
    acc, hits = cm.accuracy(lab, idx, k)
    assert np.array_equal(r['hit_rows'].cpu().numpy(), hits[:, 0])
    got_acc = ops.reduce_sum(ops.hit_rows(torch.from_numpy(lab).to(dev), r['top_idx']), 1.0 / B)
    assert abs(float(got_acc.item()) - float(acc)) <= 1e-6
    got_loss = ops.reduce_sum(r['loss_rows'], 1.0 / B)
    assert abs(float(got_loss.item()) - per_row.mean()) <= RTOL * per_row.mean()


def test_accuracy_known_answer_from_reference_scratch_file(dev):
    """recommendation/test.py:19-28 -> per-row hits [1,0,1,0], accuracy 0.5."""
    from recommendation_amd import ops
    lab = torch.tensor([5, 2, 9, -1], dtype=torch.int32, device=dev)
    pred = torch.tensor([[9, 3, 5, 2, 1], [8, 9, 0, 6, 5], [1, 9, 3, 4, 5], [1, 2, 3, 4, 5]], dtype=torch.int32,
                        device=dev)
    h = ops.hit_rows(lab, pred)
    assert h.cpu().numpy().tolist() == [1.0, 0.0, 1.0, 0.0]
    assert float(ops.reduce_sum(h, 0.25).item()) == 0.5


def test_adam_tf_dense(dev):
    from recommendation_amd import ops
    rng = np.random.default_rng(11)
    n = 5910
    p, m, v, g = (rng.standard_normal(n).astype(np.float32) for _ in range(4))
    v = np.abs(v)
    pd, md, vd, gd = (torch.from_numpy(a.copy()).to(dev) for a in (p, m, v, g))
    lr_t = float(oe.adam_lr_t(1e-4, 0.9, 0.999, 1))
    ops.adam_tf_dense(pd, md, vd, gd, lr_t, 0.9, 0.999, 1e-8)
    _, omb1, omb2 = oe.tf_adam_scalars(1e-4, 0.9, 0.999, 1)
    p64, m64, v64 = oe.adam_tf_update(p.astype(np.float64), m.astype(np.float64), v.astype(np.float64),
                                      g.astype(np.float64), lr_t, omb1, omb2, 1e-8)
    assert rel_err(md.cpu().numpy(), m64) <= 1e-6 and rel_err(vd.cpu().numpy(), v64) <= 1e-6
    assert rel_err(pd.cpu().numpy(), p64) <= 1e-6


@pytest.mark.parametrize("dim", [4, 16, 64, 128])
@pytest.mark.parametrize("idx_dtype", [np.int32, np.int64])
def test_fm_forward(dev, dim, idx_dtype):
    from recommendation_amd import ops
    rows = [1460, 583, 3000, 305, 24, 3, 10, 4, 18, 15, 105, 27, 2173]
    st = ofm.State(rows, dim, 13, seed=3)
    st.bias = np.float32(0.3)
    B = 515
    ids, dense, labels = ofm.synthetic_batch(B, rows, 13, seed=4, idx_dtype=idx_dtype)
    err = ops.new_error_word(dev)
    table = ops.FmRecords(int(st.toffs[-1]), dim, ops.MI_OPT_ADAM_TF, dev)
    table.rec[:, :dim] = torch.from_numpy(st.V).to(dev)
    table.rec[:, table.off_w1] = torch.from_numpy(st.w1).to(dev)
    r = ops.fm_fwd(table, torch.from_numpy(st.toffs).to(dev), torch.from_numpy(ids).to(dev),
                   torch.from_numpy(dense).to(dev), torch.from_numpy(st.wd).to(dev),
                   torch.tensor([st.bias], device=dev), torch.from_numpy(labels).to(dev), 1.0 / B, err)
    ops.check_error_word(err)
    z, S, _ = ofm.forward(st.V, st.w1, st.toffs, ids, dense, st.wd, st.bias)
    per, sig = ofm.bce_with_logits(z, labels.astype(np.float64))
    assert rel_err(r['S'].cpu().numpy(), S) <= RTOL
    assert rel_err(r['logit'].cpu().numpy(), z) <= RTOL
    assert rel_err(r['loss_rows'].cpu().numpy(), per) <= RTOL
    assert rel_err(r['g'].cpu().numpy(), (sig - labels) / B) <= 2e-5


@pytest.mark.parametrize("opt", ["sgd", "adagrad_rowwise", "adam_tf"])
@pytest.mark.parametrize("dist", ["uniform", "zipf"])
def test_fm_train_step(dev, opt, dist):
    """Forward + sorted sparse backward + fused update for 3 steps vs the numpy FM."""
    from recommendation_amd.fm import FactorizationMachine
    rows = [1460, 583, 30000, 305, 24, 3, 10, 4, 18, 15, 105, 27, 2173]
    ref = ofm.State(rows, 16, 13, seed=5, opt=opt)
    fm = FactorizationMachine(rows, 16, 13, device=dev, opt=opt, lr=1e-2)
    fm.load_state(ref)
    B = 2048
    for step in range(3):
        ids, dense, labels = ofm.synthetic_batch(B, rows, 13, seed=10 + step, dist=dist)
        loss = fm.train_step(torch.from_numpy(ids).to(dev), torch.from_numpy(dense).to(dev),
                             torch.from_numpy(labels).to(dev))
        want_loss, _ = ofm.train_step(ref, ids, dense, labels, lr=1e-2)
        assert abs(float(loss.item()) - want_loss) <= 1e-5 * abs(want_loss)
    # Parameters after 3 steps.  SGD / Adagrad are smooth in the gradient: 1e-5.  Adam's step is
    # lr_t*m/(sqrt(v)+eps) with eps=1e-8: for the tiny gradients of a mean loss (|g| ~ 1e-4..1e-9)
    # it amplifies the fp32 rounding of the run sums by up to 0.1/eps, so fp32 (device, and the
    # TensorFlow reference itself) and the fp64 oracle legitimately differ by ~1e-4*lr per step.
    held("fm_train_step %s V" % opt, rel_err(fm.V.cpu().numpy(), ref.V), ADAM_TOL if opt == "adam_tf" else 1e-5)
    held("fm_train_step %s w1" % opt, rel_err(fm.w1.cpu().numpy(), ref.w1), 1e-5)
    held("fm_train_step %s wd" % opt, rel_err(fm.wd.cpu().numpy(), ref.wd), 1e-5)
    assert abs(float(fm.bias.item()) - float(ref.bias)) <= 1e-6


@pytest.mark.parametrize("B", [700, 3000])
def test_runless_plan_matches_the_run_arrays_path(dev, B):
    """No unique_rows / seg_offsets in the plan: the apply kernels find the runs in sorted_rows.
    B=700: every table sorted in LDS (3500 entries per table, padding ends each table's block);
    B=3000: the global multi-pass sort (15000 per table).  Both must give the compacted-run
    path's sums (<= 1e-6: the summation trees differ where a run crosses a wave) and be bitwise
    reproducible from call to call, long runs included."""
    from recommendation_amd import ops
    rng = np.random.default_rng(12)
    dim = 16
    rows, T, _, L, w, toffs, _, _ = _bag_problem(rng, dim, "sum")
    ids = np.stack([rng.integers(-1, r, size=(B, L)) for r in rows], axis=1).astype(np.int32)
    ids[:, 0, :] = rng.integers(0, 3, size=(B, L))
    grad = rng.standard_normal((B, T * dim)).astype(np.float32)
    total = int(toffs[-1])
    g = torch.from_numpy(grad).to(dev)
    outs = []
    for runs in (True, False, False):
        plan = ops.SortedPlan(torch.from_numpy(toffs).to(dev), T, total, torch.from_numpy(ids).to(dev), bag_len=L,
                              runs=runs)
        outs.append(ops.bwd_dense_grad(plan, g, dim, total).cpu().numpy())
        assert plan.long_list[:2].tolist() == [0, 0]          # the queue is left empty for the next call
    assert np.array_equal(outs[1], outs[2])
    assert rel_err(outs[1], outs[0]) <= 1e-6
    want = oe.dense_grad(total, dim, oe.row_grads(toffs, T, ids, grad, dim, bag_len=L))
    assert rel_err(outs[1], want) <= RTOL


@pytest.mark.parametrize("B,L", [(8192, 1), (1000, 7), (1, 1)])
def test_per_table_lds_sort_is_the_global_stable_sort(dev, B, L):
    """Without padding the one-launch per-table sort must equal numpy's stable argsort by
    (global row, position) bit for bit."""
    from recommendation_amd import ops
    rng = np.random.default_rng(13)
    rows = [3, 1000, 50, 100000, 10131227, 24]
    T = len(rows)
    toffs = np.concatenate([[0], np.cumsum(rows)]).astype(np.int64)
    ids = np.stack([rng.integers(0, r, size=(B, L)) for r in rows], axis=1).astype(np.int64)
    plan = ops.SortedPlan(torch.from_numpy(toffs).to(dev), T, int(toffs[-1]), torch.from_numpy(ids).to(dev),
                          bag_len=L, runs=False)
    want_rows, want_perm, _, _ = oe.prepare(toffs, T, ids, bag_len=L)
    assert np.array_equal(plan.sorted_rows.cpu().numpy().astype(np.int64) & 0xFFFFFFFF, want_rows)
    assert np.array_equal(plan.perm.cpu().numpy(), want_perm)


@pytest.mark.parametrize("B", [4096, 6000, 8192])
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_per_table_lds_sort_with_hot_neighbouring_ids_in_a_table_of_2_pow_27_rows(dev, B, seed):
    """A table of more than 2^26 rows shared by several workgroups, half of whose ids are a few ADJACENT rows (a
    frequency-ordered vocabulary): the sampled cuts that divide the table between the workgroups are taken in the
    full (row, position) order (ranked on keys cut to 26 bits they would not ascend: empty and overlapping parts).
    Sorted rows and permutation bit for bit against numpy's stable argsort; no table here is materialised (only
    the ids are sorted)."""
    from recommendation_amd import ops
    rng = np.random.default_rng(700 + seed)
    rows = [1 << 27, 24, (1 << 28) + 12345, 1 << 27]
    T = len(rows)
    toffs = np.concatenate([[0], np.cumsum(rows)]).astype(np.int64)
    cols = []
    for t, r in enumerate(rows):
        col = rng.integers(0, r, size=B)
        hot = rng.random(B) < 0.5
        first = 0 if t != 3 else r - 4              # the hot rows also at the very top of the key range
        col[hot] = first + rng.integers(0, 4, size=int(hot.sum()))
        cols.append(col)
    ids = np.stack(cols, axis=1).astype(np.int64)[:, :, None]
    plan = ops.SortedPlan(torch.from_numpy(toffs).to(dev), T, int(toffs[-1]), torch.from_numpy(ids).to(dev),
                          bag_len=1, runs=False)
    want_rows, want_perm, _, _ = oe.prepare(toffs, T, ids, bag_len=1)
    got_rows = plan.sorted_rows.cpu().numpy().astype(np.int64) & 0xFFFFFFFF
    assert np.array_equal(got_rows, want_rows)
    assert np.array_equal(plan.perm.cpu().numpy(), want_perm)


def _sample_positions(n):
    """Where the large-table sort takes its 1024 samples (csrc/table_sort.h: evenly spaced, each shifted by a fixed
    pseudo-random amount below the spacing)."""
    stride = n // 1024
    i = np.arange(1024, dtype=np.uint64)
    h = (i * np.uint64(2654435761)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(15)
    return (i * np.uint64(stride) + (((h >> np.uint64(8)) * np.uint64(stride)) >> np.uint64(24))).astype(np.int64)


def _check_sorted_plan(plan, toffs, T, ids, L):
    total = int(toffs[-1])
    want_rows, want_perm, _, _ = oe.prepare(toffs, T, ids, bag_len=L)
    got_rows = plan.sorted_rows.cpu().numpy().astype(np.int64) & 0xFFFFFFFF
    got_perm = plan.perm.cpu().numpy()
    per_table = got_rows.size // T
    blocks = got_rows.reshape(T, per_table)
    for t in range(T):                     # padding closes each table's block
        nv = int((blocks[t] != total).sum())
        assert (blocks[t][:nv] != total).all() and (blocks[t][nv:] == total).all()
    keep = got_rows != total
    assert np.array_equal(got_rows[keep], want_rows)
    assert np.array_equal(got_perm[keep], want_perm)
    # the padding entries' positions are the remaining ones, each once
    assert np.array_equal(np.sort(got_perm), np.arange(got_perm.size))


@pytest.mark.parametrize("B,L,idt", [(65536, 1, np.int64), (20000, 3, np.int32), (131072, 1, np.int32), (8193, 1, np.int64),
                                     (30011, 1, np.int32)])
@pytest.mark.parametrize("kind", ["uniform", "padded", "one_row", "period_of_the_sample_spacing"])
def test_large_table_lds_sort_in_parts_is_the_global_stable_sort(dev, B, L, idt, kind):
    """More than 8192 entries per table: ceil(n / 4096) workgroups share a table, each sorting the entries between
    two cuts taken from 1024 samples.  Bit for bit numpy's stable argsort by (row, position), from sample-major ids
    (mi_embag_bwd_prepare: a launch lays them out table by table first) - with out-of-range ids, with every id equal
    (the cuts then divide by position alone) and with ids that repeat with the period of the sample spacing."""
    from recommendation_amd import ops
    rng = np.random.default_rng(17)
    rows = [3, 1000, 10131227, 24, 1 << 27]
    T = len(rows)
    toffs = np.concatenate([[0], np.cumsum(rows)]).astype(np.int64)
    n = B * L
    cols = []
    for r in rows:
        if kind == "one_row":
            col = np.full((B, L), r - 1)
        elif kind == "period_of_the_sample_spacing":
            col = (np.arange(n) % (n // 1024)).reshape(B, L) % r
        else:
            col = rng.integers(0, r, size=(B, L))
        if kind == "padded":
            bad = rng.random((B, L)) < 0.2
            col[bad] = np.where(rng.random(int(bad.sum())) < 0.5, -1, r + 5)
        cols.append(col)
    ids = np.stack(cols, axis=1).astype(idt)
    plan = ops.SortedPlan(torch.from_numpy(toffs).to(dev), T, int(toffs[-1]), torch.from_numpy(ids).to(dev),
                          bag_len=L, runs=False)
    _check_sorted_plan(plan, toffs, T, ids, L)


@pytest.mark.parametrize("idt", [np.int32, np.int64])
def test_large_table_sort_narrows_a_part_the_samples_misjudged(dev, idt):
    """Ids chosen against the sampling: the sampled positions hold keys spread over the whole table, every other
    position holds row 0.  The first part then receives nearly the whole table - more than a workgroup's LDS holds -
    and must halve its range of the (row, position) order until the count fits, part after part.  Field-major ids
    through mi_fm_sort_ids; same bits as the stable argsort."""
    from recommendation_amd import ops
    B, rows = 65536, [1 << 20, 50]
    T = len(rows)
    toffs = np.concatenate([[0], np.cumsum(rows)]).astype(np.int64)
    pos = _sample_positions(B)
    assert np.unique(pos).size == 1024 and pos.max() < B
    col0 = np.zeros(B, dtype=np.int64)
    col0[pos] = 1 + np.arange(1024) * 1000
    col1 = np.zeros(B, dtype=np.int64)
    col1[pos] = np.arange(1024) % 50
    ids = np.stack([col0, col1], axis=1).astype(idt)
    d_ids = torch.from_numpy(ids).to(dev)
    plan = ops.SortedPlan(torch.from_numpy(toffs).to(dev), T, int(toffs[-1]), d_ids, bag_len=1, runs=False)
    plan.sorted_rows.fill_(0)
    plan.perm.fill_(0)
    ops.fm_sort_ids(torch.from_numpy(toffs).to(dev), d_ids, plan, ids_tm=d_ids.t().contiguous())
    _check_sorted_plan(plan, toffs, T, ids[:, :, None], 1)
    with pytest.raises(ops.MiError):
        ops.fm_sort_ids(torch.from_numpy(toffs).to(dev), d_ids, plan)      # sample-major ids: only up to 8192


@pytest.mark.parametrize("opt", ["adagrad_rowwise", "adam_tf"])
def test_fm_two_launch_step_at_batch_65536_equals_the_five_launch_step_bitwise(dev, opt):
    """Field-major ids at a batch beyond one workgroup's LDS: {schedule, forward, sort in parts} + {update, dense side}
    leave the bits of the five separate launches (whose sort goes through mi_embag_bwd_prepare), eager and replayed."""
    from recommendation_amd.fm import FactorizationMachine
    rows = [1460, 583, 300000, 305, 24, 3, 10, 4, 18, 15, 105, 27, 2173]
    B = 65536
    ref = ofm.State(rows, 16, 13, seed=7, opt=opt)
    models = [FactorizationMachine(rows, 16, 13, device=dev, opt=opt, lr=1e-2) for _ in range(3)]
    for m in models:
        m.load_state(ref)
    bufs = tm = None
    for step in range(2):
        arrs = ofm.synthetic_batch(B, rows, 13, seed=90 + step, dist='zipf')
        if bufs is None:
            bufs = tuple(torch.from_numpy(x).to(dev) for x in arrs)
            tm = bufs[0].t().contiguous()
        else:
            for dst, src in zip(bufs, arrs):
                dst.copy_(torch.from_numpy(src))
            tm.copy_(bufs[0].t())
        l0 = models[0].train_step(*bufs).clone()                          # no field-major ids: five launches
        l1 = models[1].train_step(*bufs, ids_tm=tm).clone()
        l2 = models[2].train_step(*bufs, ids_tm=tm, graph=True).clone()
        assert torch.equal(l0, l1) and torch.equal(l0, l2)
        assert torch.equal(models[0].plan.sorted_rows, models[1].plan.sorted_rows)
        assert torch.equal(models[0].plan.perm, models[1].plan.perm)
    for m in models[1:]:
        assert torch.equal(models[0].table.rec, m.table.rec)
        assert torch.equal(models[0].dp, m.dp) and torch.equal(models[0].sched, m.sched)


def test_fm_record_layout_is_checked(dev):
    from recommendation_amd import ops
    t = ops.FmRecords(10, 16, ops.MI_OPT_ADAM_TF, dev)
    assert t.stride % 32 == 0 and t.off_vs0 % 4 == 0 and t.off_vs1 == t.off_vs0 + 16
    t.c.off_vs1 = t.off_vs0 + 4                       # m and v overlap
    plan = ops.SortedPlan(ops.table_offsets([10], dev), 1, 10, torch.zeros((4, 1), dtype=torch.int32, device=dev),
                          bag_len=1, runs=False)
    z = torch.zeros((4, 16), device=dev)
    with pytest.raises(ops.MiError):
        ops.fm_bwd_apply(plan, t, z[:, 0].contiguous(), z, ops.MI_OPT_ADAM_TF, 0.1)


@pytest.mark.parametrize("opt", ["adagrad_rowwise", "adam_tf"])
def test_fm_graph_replay_equals_eager_bitwise(dev, opt):
    """The captured step (sort beside the forward, dense side beside the sparse update, step size
    read from the device-resident schedule) must leave exactly the bits the eager step leaves."""
    from recommendation_amd.fm import FactorizationMachine
    rows = [1460, 583, 30000, 305, 24, 3, 10, 4, 18, 15, 105, 27, 2173]
    ref = ofm.State(rows, 16, 13, seed=5, opt=opt)
    a = FactorizationMachine(rows, 16, 13, device=dev, opt=opt, lr=1e-2)
    b = FactorizationMachine(rows, 16, 13, device=dev, opt=opt, lr=1e-2)
    a.load_state(ref)
    b.load_state(ref)
    ids, dense, labels = ofm.synthetic_batch(4096, rows, 13, seed=21, dist='zipf')
    bufs = (torch.from_numpy(ids).to(dev), torch.from_numpy(dense).to(dev), torch.from_numpy(labels).to(dev))
    for step in range(4):
        ids2, dense2, labels2 = ofm.synthetic_batch(4096, rows, 13, seed=30 + step, dist='zipf')
        for dst, src in zip(bufs, (ids2, dense2, labels2)):
            dst.copy_(torch.from_numpy(src))          # same buffers, new contents
        la = a.train_step(*bufs).clone()
        lb = b.train_step(*bufs, graph=True).clone()
        assert torch.equal(la, lb)
    assert len(b._graphs) == 1
    assert torch.equal(a.table.rec, b.table.rec) and torch.equal(a.dp, b.dp) and torch.equal(a.sched, b.sched)
    # and the device-resident schedule is TensorFlow's fp32 schedule
    alpha, _, _ = oe.tf_adam_scalars(1e-2, 0.9, 0.999, 4)
    assert float(a.sched[0].item()) == np.float32(alpha)


@pytest.mark.parametrize("opt", ["sgd", "adagrad_rowwise", "adam_tf"])
@pytest.mark.parametrize("B", [700, 4096, 8192])
def test_fm_two_launch_step_equals_five_launch_step_bitwise(dev, opt, B):
    """{schedule, forward, sort} in one launch and {sparse update, dense side} in another must leave
    exactly the bits of the five separate launches, eager and under graph replay."""
    from recommendation_amd.fm import FactorizationMachine
    rows = [1460, 583, 30000, 305, 24, 3, 10, 4, 18, 15, 105, 27, 2173]
    ref = ofm.State(rows, 16, 13, seed=7, opt=opt)
    models = [FactorizationMachine(rows, 16, 13, device=dev, opt=opt, lr=1e-2) for _ in range(3)]
    for m in models:
        m.load_state(ref)
    models[0].fused = False
    bufs = None
    for step in range(3):
        arrs = ofm.synthetic_batch(B, rows, 13, seed=50 + step, dist='zipf')
        if bufs is None:
            bufs = tuple(torch.from_numpy(x).to(dev) for x in arrs)
        else:
            for dst, src in zip(bufs, arrs):
                dst.copy_(torch.from_numpy(src))
        l0 = models[0].train_step(*bufs).clone()
        l1 = models[1].train_step(*bufs).clone()
        l2 = models[2].train_step(*bufs, graph=True).clone()
        assert torch.equal(l0, l1) and torch.equal(l0, l2)
        assert torch.equal(models[0].plan.sorted_rows, models[1].plan.sorted_rows)
        assert torch.equal(models[0].plan.perm, models[1].plan.perm)
    for m in models[1:]:
        assert torch.equal(models[0].table.rec, m.table.rec)
        assert torch.equal(models[0].dp, m.dp) and torch.equal(models[0].sched, m.sched)


@pytest.mark.parametrize("idt", ["int64", "int32"])
@pytest.mark.parametrize("B", [700, 8192])
def test_fm_field_major_ids_give_the_same_bits(dev, B, idt):
    """The two-launch step fed the ids as one row per field ([T][B], the reference's one tensor per
    feature, recommendation/code/r_model.py:42-51) must leave the bits of the [B][T] step: same
    sorted rows, same permutation, same records, dense parameters and loss; eager and replayed."""
    from recommendation_amd.fm import FactorizationMachine
    rows = [1460, 583, 30000, 305, 24, 3, 10, 4, 18, 15, 105, 27, 2173]
    ref = ofm.State(rows, 16, 13, seed=9, opt='adam_tf')
    models = [FactorizationMachine(rows, 16, 13, device=dev, opt='adam_tf', lr=1e-2) for _ in range(3)]
    for m in models:
        m.load_state(ref)
    bufs = tm = None
    for step in range(3):
        arrs = list(ofm.synthetic_batch(B, rows, 13, seed=70 + step, dist='zipf'))
        arrs[0] = arrs[0].astype(idt)
        if bufs is None:
            bufs = tuple(torch.from_numpy(x).to(dev) for x in arrs)
            tm = bufs[0].t().contiguous()
        else:
            for dst, src in zip(bufs, arrs):
                dst.copy_(torch.from_numpy(src))
            tm.copy_(bufs[0].t())
        l0 = models[0].train_step(*bufs).clone()
        l1 = models[1].train_step(*bufs, ids_tm=tm).clone()
        l2 = models[2].train_step(*bufs, ids_tm=tm, graph=True).clone()
        assert torch.equal(l0, l1) and torch.equal(l0, l2)
        assert torch.equal(models[0].plan.sorted_rows, models[1].plan.sorted_rows)
        assert torch.equal(models[0].plan.perm, models[1].plan.perm)
    for m in models[1:]:
        assert torch.equal(models[0].table.rec, m.table.rec)
        assert torch.equal(models[0].dp, m.dp) and torch.equal(models[0].sched, m.sched)
    with pytest.raises(RuntimeError):
        models[1].train_step(*bufs, ids_tm=tm[:, :-1])


def test_piecewise_long_runs_exact_and_reproducible(dev):
    """Runs far beyond 4096 entries (one-, two- and three-row tables at batch 32768) are summed in
    pieces by several workgroups and combined by the last one to arrive: integer gradients make
    every order give the same fp32 sum, so the result must equal the exact column sums, and it
    must not change from call to call."""
    from recommendation_amd import ops
    rng = np.random.default_rng(21)
    dim, B = 16, 32768
    rows = [1, 2, 3, 50000]
    T = len(rows)
    toffs = np.concatenate([[0], np.cumsum(rows)]).astype(np.int64)
    ids = np.stack([rng.integers(0, r, size=B) for r in rows], axis=1).astype(np.int64)     # [B][T]
    grad = rng.integers(-3, 4, size=(B, T * dim)).astype(np.float32)
    total = int(toffs[-1])
    want = np.zeros((total, dim), np.float64)
    for t in range(T):
        np.add.at(want, toffs[t] + ids[:, t], grad[:, t * dim:(t + 1) * dim])
    outs = []
    for _ in range(3):
        plan = ops.SortedPlan(torch.from_numpy(toffs).to(dev), T, total, torch.from_numpy(ids).to(dev), bag_len=1,
                              runs=False)
        w = torch.zeros((total, dim), device=dev)
        ops.bwd_apply_sgd(plan, torch.from_numpy(grad).to(dev), w, -1.0)      # w = 0 + 1.0 * summed gradient
        outs.append(w.cpu().numpy())
    assert np.array_equal(outs[0], want.astype(np.float32))
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])


@pytest.mark.parametrize("dim", [16, 128])
def test_lone_padding_id_inside_a_wave_of_distinct_rows(dev, dim):
    """The per-table sort (plan without run arrays, <= 8192 entries per table) leaves a table's padding
    entries at the end of THAT table's block, i.e. in the middle of the array.  With all other ids
    distinct, a single -1 id in table 0 is a lone entry that differs from both neighbours; the wave that
    holds it must not treat it as a row (it would read and write one row past the tables).  Buffers
    carry a guard row behind total_rows, which must keep its bits; results equal the oracle's."""
    from recommendation_amd import ops
    rng = np.random.default_rng(41)
    rows = [5000, 7000, 3000]
    T, B = 3, 1024
    toffs = np.concatenate([[0], np.cumsum(rows)]).astype(np.int64)
    total = int(toffs[-1])
    ids = np.stack([rng.permutation(r)[:B] for r in rows], axis=1).astype(np.int64)[:, :, None]
    ids[100, 0, 0] = -1                                          # exactly one padding id, in table 0
    grad = rng.standard_normal((B, T * dim)).astype(np.float32)
    w = (rng.standard_normal((total + 1, dim)) * 0.1).astype(np.float32)     # row `total` is the guard
    td = torch.from_numpy(toffs).to(dev)
    plan = ops.SortedPlan(td, T, total, torch.from_numpy(ids).to(dev), bag_len=1, runs=False)
    sr = plan.sorted_rows.cpu().numpy().astype(np.int64) & 0xFFFFFFFF
    assert sr[B - 1] == total and (sr[:B - 1] < total).all() and (sr[B:] < total).all()   # sentinel mid-array
    g = torch.from_numpy(grad).to(dev)
    want = oe.row_grads(toffs, T, ids, grad, dim, bag_len=1)
    dense = oe.dense_grad(total, dim, want)
    # dense gradient into a buffer with a guard row
    dW = torch.full((total + 1, dim), 7.0, device=dev)
    from recommendation_amd import _lib
    _lib.call('mi_embag_bwd_dense_grad', plan.ref(), ops._p(g), T * dim, dim, total, ops._p(dW), ops._stream())
    dWh = dW.cpu().numpy()
    assert np.array_equal(dWh[total], np.full(dim, 7.0, np.float32))
    assert rel_err(dWh[:total], dense) <= RTOL
    grads = {int(r): dWh[int(r)].astype(np.float64) for r in want}
    lr, eps, TOL = 0.05, 1e-8, 2e-6
    wd = torch.from_numpy(w.copy()).to(dev)
    ops.bwd_apply_sgd(plan, g, wd, lr)
    wr = w.astype(np.float64)
    oe.apply_sgd(wr[:total], grads, lr)
    assert np.array_equal(wd.cpu().numpy()[total], w[total])
    assert rel_err(wd.cpu().numpy()[:total], wr[:total]) <= TOL
    st = rng.random(total + 1).astype(np.float32)
    wd, std = torch.from_numpy(w.copy()).to(dev), torch.from_numpy(st.copy()).to(dev)
    ops.bwd_apply_adagrad_rowwise(plan, g, wd, std, lr, eps)
    wr, srf = w.astype(np.float64), st.astype(np.float64)
    oe.apply_adagrad_rowwise(wr[:total], srf[:total], grads, lr, eps)
    assert np.array_equal(wd.cpu().numpy()[total], w[total]) and std.cpu().numpy()[total] == st[total]
    assert rel_err(std.cpu().numpy()[:total], srf[:total]) <= TOL and rel_err(wd.cpu().numpy()[:total], wr[:total]) <= TOL


def test_fm_graph_survives_a_batch_of_another_shape(dev):
    """graph step at B=4096, an eager ragged batch of 1000, then the graph step again: the replayed
    graph must still read and write live buffers (each batch shape keeps its own buffer set) and the
    model must match an eager twin bit for bit."""
    from recommendation_amd.fm import FactorizationMachine
    rows = [1460, 583, 30000, 305, 24, 3, 10, 4, 18, 15, 105, 27, 2173]
    ref = ofm.State(rows, 16, 13, seed=5, opt='adam_tf')
    a = FactorizationMachine(rows, 16, 13, device=dev, opt='adam_tf', lr=1e-2)
    b = FactorizationMachine(rows, 16, 13, device=dev, opt='adam_tf', lr=1e-2)
    a.load_state(ref)
    b.load_state(ref)
    big = tuple(torch.from_numpy(x).to(dev) for x in ofm.synthetic_batch(4096, rows, 13, seed=61, dist='zipf'))
    small = tuple(torch.from_numpy(x).to(dev) for x in ofm.synthetic_batch(1000, rows, 13, seed=62, dist='zipf'))
    for bufs, graph in ((big, True), (small, False), (big, True), (small, False), (big, True)):
        la = a.train_step(*bufs).clone()
        lb = b.train_step(*bufs, graph=graph).clone()
        assert torch.equal(la, lb)
    assert len(b._graphs) == 1
    assert torch.equal(a.table.rec, b.table.rec) and torch.equal(a.dp, b.dp)


@pytest.mark.parametrize("opt", ["adagrad_rowwise", "adam_tf"])
@pytest.mark.parametrize("tm", [False, True])
def test_fm_next_batch_sort_prefetch_leaves_the_same_bits(dev, opt, tm):
    """train_step(..., next_ids=) sorts the NEXT batch's ids on a side stream beside this step and
    skips the sort inside the next step.  Same bits as the plain step: eager and replayed from HIP
    graphs, with a caller that breaks its promise once (passes other ids than announced: that batch
    is then sorted inside its own step) and with a ragged batch in between."""
    from recommendation_amd.fm import FactorizationMachine
    rows = [1460, 583, 30000, 305, 24, 3, 10, 4, 18, 15, 105, 27, 2173]
    ref = ofm.State(rows, 16, 13, seed=7, opt=opt)
    models = [FactorizationMachine(rows, 16, 13, device=dev, opt=opt, lr=1e-2) for _ in range(3)]
    for m in models:
        m.load_state(ref)
    B, nb = 4096, 4
    bat = [tuple(torch.from_numpy(x).to(dev) for x in ofm.synthetic_batch(B, rows, 13, seed=90 + j, dist='zipf'))
           for j in range(nb)]
    tms = [b[0].t().contiguous() if tm else None for b in bat]
    small = tuple(torch.from_numpy(x).to(dev) for x in ofm.synthetic_batch(1000, rows, 13, seed=99, dist='zipf'))
    order = [0, 1, 2, 3, 0, 1, 3, 2, 0, 1]            # step 6 gets batch 3 although step 5 announced batch 2
    for k, j in enumerate(order):
        nxt = (j + 1) % nb
        l0 = models[0].train_step(*bat[j], ids_tm=tms[j]).clone()
        l1 = models[1].train_step(*bat[j], ids_tm=tms[j], next_ids=bat[nxt][0], next_ids_tm=tms[nxt]).clone()
        l2 = models[2].train_step(*bat[j], ids_tm=tms[j], next_ids=bat[nxt][0], next_ids_tm=tms[nxt], graph=True).clone()
        assert torch.equal(l0, l1) and torch.equal(l0, l2), k
        assert torch.equal(models[0].plan.sorted_rows, models[1].plan.sorted_rows)
        assert torch.equal(models[0].plan.perm, models[2].plan.perm)
        if k == 3:
            for m in models:
                m.train_step(*small)
    for m in models[1:]:
        assert torch.equal(models[0].table.rec, m.table.rec)
        assert torch.equal(models[0].dp, m.dp) and torch.equal(models[0].sched, m.sched)


def test_fm_step_that_brings_in_the_next_batch_leaves_the_same_bits(dev):
    """train_step(stage=...): the second launch also copies the next batch's pinned host block (ids field-major int32 |
    dense | labels) into the other device block.  Same records, dense parameters, schedule and losses as the step on
    resident tensors; the device blocks hold the host blocks' bytes; eager and captured in a graph by the caller."""
    from recommendation_amd import ops
    from recommendation_amd.fm import FactorizationMachine
    rows = [1460, 583, 30000, 305, 24, 3, 10, 4, 18, 15, 105, 27, 2173]
    T, nd, B, K = len(rows), 13, 2048, 4
    ref = ofm.State(rows, 16, nd, seed=31, opt='adam_tf')
    a = FactorizationMachine(rows, 16, nd, device=dev, opt='adam_tf', lr=1e-2)
    b = FactorizationMachine(rows, 16, nd, device=dev, opt='adam_tf', lr=1e-2)
    for m in (a, b):
        m.load_state(ref)
    data = [ofm.synthetic_batch(B, rows, nd, seed=400 + j, dist='zipf') for j in range(K)]
    sizes = [T * B * 4, B * nd * 4, B * 4]
    offs = [0, -(-sizes[0] // 256) * 256]
    offs.append(offs[1] + -(-sizes[1] // 256) * 256)
    total = offs[2] + -(-sizes[2] // 256) * 256

    def views(block):
        return (block[offs[0]:offs[0] + sizes[0]].view(torch.int32).view(T, B),
                block[offs[1]:offs[1] + sizes[1]].view(torch.float32).view(B, nd),
                block[offs[2]:offs[2] + sizes[2]].view(torch.float32).view(B))

    host = []
    for ids, dense, labels in data:
        blk = torch.zeros((total,), dtype=torch.uint8).pin_memory()
        v = views(blk)
        v[0].copy_(torch.from_numpy(ids.T.astype(np.int32)))
        v[1].copy_(torch.from_numpy(dense))
        v[2].copy_(torch.from_numpy(labels))
        host.append(blk)
    dblk = [torch.zeros((total,), dtype=torch.uint8, device=dev) for _ in range(2)]
    bufs = [views(d) for d in dblk]
    shape_only = torch.empty((B, T), dtype=torch.int32, device=dev)
    res = [tuple(torch.from_numpy(x).to(dev) for x in (ids.astype(np.int32), dense, labels)) for ids, dense, labels in data]
    res_tm = [r[0].t().contiguous() for r in res]

    def staged_steps(first):
        out = []
        for i in range(first, first + K):
            k = i % 2
            out.append(b.train_step(shape_only, bufs[k][1], bufs[k][2], ids_tm=bufs[k][0],
                                    stage=(dblk[k ^ 1], host[(i + 1) % K])).clone())
        return out

    ops.stage_from_host(dblk[0], host[0])
    got = staged_steps(0)
    want = [a.train_step(*res[i % K], ids_tm=res_tm[i % K]).clone() for i in range(K)]
    assert torch.equal(torch.cat(got), torch.cat(want))
    # after K (even) steps block 0 holds batch 0 again and block 1 batch K - 1
    assert torch.equal(dblk[0].cpu(), host[0]) and torch.equal(dblk[1].cpu(), host[K - 1])
    # the same steps captured by the caller as one graph (one chain of launches, no second stream)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        losses = torch.cat(staged_steps(0))
    b.steps -= K                       # (capture does not run the steps)
    g.replay()
    b.steps += K
    want = torch.cat([a.train_step(*res[i % K], ids_tm=res_tm[i % K]).clone() for i in range(K)])
    assert torch.equal(losses, want)
    assert torch.equal(a.table.rec, b.table.rec) and torch.equal(a.dp, b.dp) and torch.equal(a.sched, b.sched)
    with pytest.raises(ops.MiError):
        b.train_step(shape_only, bufs[0][1], bufs[0][2], ids_tm=bufs[0][0], stage=(dblk[1], host[0].clone()))   # not pinned


def test_fm_several_steps_in_one_graph_leave_the_same_bits(dev):
    """train_steps (one HIP graph of consecutive steps) against train_step per batch: same losses, records, dense
    parameters and schedule, bit for bit, also when the replay is repeated with new contents in the same buffers."""
    from recommendation_amd.fm import FactorizationMachine
    rows = [1460, 583, 30000, 305, 24, 3, 10, 4, 18, 15, 105, 27, 2173]
    ref = ofm.State(rows, 16, 13, seed=21, opt='adam_tf')
    a = FactorizationMachine(rows, 16, 13, device=dev, opt='adam_tf', lr=1e-2)
    b = FactorizationMachine(rows, 16, 13, device=dev, opt='adam_tf', lr=1e-2)
    for m in (a, b):
        m.load_state(ref)
    B, K = 2048, 4
    bufs = [tuple(torch.from_numpy(x).to(dev) for x in ofm.synthetic_batch(B, rows, 13, seed=200 + j, dist='zipf'))
            for j in range(K)]
    tms = [x[0].t().contiguous() for x in bufs]
    for rnd in range(3):
        if rnd:
            for j in range(K):
                for dst, src in zip(bufs[j], ofm.synthetic_batch(B, rows, 13, seed=300 + 10 * rnd + j, dist='uniform')):
                    dst.copy_(torch.from_numpy(src))
                tms[j].copy_(bufs[j][0].t())
        losses = a.train_steps(bufs, ids_tms=tms).clone()
        want = torch.cat([b.train_step(*bufs[j], ids_tm=tms[j]).clone() for j in range(K)])
        assert torch.equal(losses, want)
        assert torch.equal(a.loss, b.loss)
    # a batch in the middle replaced by other tensors: NEW list objects (a list seen before is taken to hold the buffers
    # it held then), another graph
    other = tuple(torch.from_numpy(x).to(dev) for x in ofm.synthetic_batch(B, rows, 13, seed=999, dist='zipf'))
    bufs2 = [bufs[0], other] + bufs[2:]
    tms2 = [tms[0], other[0].t().contiguous()] + tms[2:]
    losses = a.train_steps(bufs2, ids_tms=tms2).clone()
    want = torch.cat([b.train_step(*bufs2[j], ids_tm=tms2[j]).clone() for j in range(K)])
    assert torch.equal(losses, want) and torch.equal(a.loss, b.loss)
    assert torch.equal(a.table.rec, b.table.rec) and torch.equal(a.dp, b.dp) and torch.equal(a.sched, b.sched)
    assert a.steps == b.steps == 4 * K
