"""The benchmark configurations at their FULL shapes, through the C ABI, against the numpy oracle.

  FM       26 Criteo-Kaggle tables (33.76 M rows) x dim 16, batch 8192, TensorFlow-form Adam
  DCN-v2   the same tables, d = 26*16+13 = 429 -> 448 (the 256+192 edge tile), batch 65536
  DIN      1 M items x dim 64, history 100, batch 8192
  bags     26 tables x dim 128, batch 65536: one GPU's shapes of the row-wise sharded configuration

The oracle cannot hold 33.76 M rows of parameters AND optimizer state in seconds, and does not need
to: a step only reads the rows its ids name.  The rows named by the test's batches are copied out
of the device model into COMPACT tables (per table: the distinct ids, ascending, each id replaced
by its rank), the oracle runs on those, and the results are compared row by row.  What a compact
table cannot show - that no OTHER row of the big table moved, that the sort artefacts are exact at
this size - is checked on the full arrays directly.  Tolerances as everywhere: index work bit-exact;
fp32 sums / loss 1e-5; Adam-updated rows and moments after two steps 1e-5 (achieved errors are logged: tests/errlog.py); bf16
matrix-core paths 1e-2 against the storage-emulating oracle, and the error against the EXACT
oracle is printed and bounded by 5e-2 (ReLU gates within bf16 rounding of zero flip; see
test_dcn_gpu's docstring)."""
import numpy as np
import pytest
import torch

from oracle import dcn as odcn
from oracle import din as odin
from oracle import embedding as oe
from oracle import fm as ofm
from oracle.dcn import bf16_round
from tests.errlog import held

ADAM_TOL = 1e-5   # Adam-updated rows and moments against the fp64 oracle after two steps (achieved: <= 2e-7)

pytestmark = pytest.mark.gpu

CRITEO = ofm.CRITEO_KAGGLE_CARDINALITIES


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def compact(rows, id_batches):
    """Per table the distinct ids of all batches (ascending); returns (compact row counts, the
    batches with every id replaced by its rank in its table, global rows of the compact rows)."""
    T = len(rows)
    toffs = np.concatenate([[0], np.cumsum(rows)]).astype(np.int64)
    uniq = [np.unique(np.concatenate([ids[:, t] for ids in id_batches])) for t in range(T)]
    remapped = [np.stack([np.searchsorted(uniq[t], ids[:, t]) for t in range(T)], axis=1).astype(np.int64)
                for ids in id_batches]
    glob = np.concatenate([toffs[t] + uniq[t] for t in range(T)])
    return [len(u) for u in uniq], remapped, glob


def test_fm_full_criteo_b8192_two_steps(dev):
    from recommendation_amd.fm import FactorizationMachine
    B, D, ND = 8192, 16, 13
    fm = FactorizationMachine(CRITEO, D, ND, device=dev, opt='adam_tf', lr=1e-3, seed=0)
    batches = [ofm.synthetic_batch(B, CRITEO, ND, seed=100, dist='uniform'),
               ofm.synthetic_batch(B, CRITEO, ND, seed=101, dist='zipf')]
    crow, cids, glob = compact(CRITEO, [b[0] for b in batches])
    gl = torch.from_numpy(glob).to(dev)
    st = ofm.State.__new__(ofm.State)
    st.toffs = np.concatenate([[0], np.cumsum(crow)]).astype(np.int64)
    st.V = fm.V[gl].cpu().numpy()
    st.w1 = fm.w1[gl].cpu().numpy()
    st.wd, st.bias = fm.wd.cpu().numpy().copy(), np.float32(fm.bias.item())
    st.opt, st.t = 'adam_tf', 0
    st.mV, st.vV = np.zeros_like(st.V), np.zeros_like(st.V)
    st.mw, st.vw = np.zeros_like(st.w1), np.zeros_like(st.w1)
    st.md, st.vd = np.zeros(ND + 1, np.float32), np.zeros(ND + 1, np.float32)
    before = fm.table.rec.clone()
    for step, ((ids, dense, labels), ci) in enumerate(zip(batches, cids)):
        idd = torch.from_numpy(ids).to(dev)
        # the layout the benchmark feeds: ids also as one row per field, the step replayed from a HIP graph
        loss = fm.train_step(idd, torch.from_numpy(dense).to(dev), torch.from_numpy(labels).to(dev), check_ids=True,
                             ids_tm=idd.t().contiguous(), graph=True)
        want, _ = ofm.train_step(st, ci, dense, labels, lr=1e-3)
        assert abs(float(loss.item()) - want) <= 1e-5 * abs(want), (step, float(loss.item()), want)
        # the sort artefacts at this size, bit for bit (== numpy's stable argsort by (row, position))
        want_rows, want_perm, _, _ = oe.prepare(fm.toffs.cpu().numpy(), len(CRITEO), ids, bag_len=1)
        assert np.array_equal(fm.plan.sorted_rows.cpu().numpy().astype(np.int64) & 0xFFFFFFFF, want_rows)
        assert np.array_equal(fm.plan.perm.cpu().numpy(), want_perm)
    t = fm.table
    got = t.rec[gl].cpu().numpy()
    held("fm criteo B=8192 V", rel(got[:, :D], st.V), ADAM_TOL)
    held("fm criteo B=8192 w1", rel(got[:, t.off_w1], st.w1), ADAM_TOL)
    held("fm criteo B=8192 m(V)", rel(got[:, t.off_vs0:t.off_vs0 + D], st.mV), ADAM_TOL)
    held("fm criteo B=8192 v(V)", rel(got[:, t.off_vs1:t.off_vs1 + D], st.vV), ADAM_TOL)
    held("fm criteo B=8192 m(w1)", rel(got[:, t.off_ws0], st.mw), ADAM_TOL)
    held("fm criteo B=8192 v(w1)", rel(got[:, t.off_ws1], st.vw), ADAM_TOL)
    held("fm criteo B=8192 wd", rel(fm.wd.cpu().numpy(), st.wd), ADAM_TOL)
    assert abs(float(fm.bias.item()) - float(st.bias)) <= 1e-6
    # exactly the rows the two batches name moved; the other 33.5 M records keep their bits
    moved = (t.rec != before).any(dim=1).nonzero().view(-1).cpu().numpy()
    assert np.array_equal(moved, glob)


def _dcn_state_from(m, rows_c, V_c):
    """Oracle state holding the device model's weights as the contractions read them (bf16 copies)."""
    st = odcn.State.__new__(odcn.State)
    d = m.d
    st.toffs = np.concatenate([[0], np.cumsum(rows_c)]).astype(np.int64)
    st.T, st.dim, st.n_dense, st.d = m.T, m.dim, m.n_dense, d
    st.V = V_c
    st.Wc = [p.h.float().cpu().numpy()[:d, :d] for p in m.Wc]
    st.bc = [p.w.cpu().numpy()[:d] for p in m.bc]
    st.Wm = [p.h.float().cpu().numpy() for p in m.Wm]
    st.Wm[0] = st.Wm[0][:, :d]
    st.bm = [p.w.cpu().numpy() for p in m.bm]
    K = m.mlp[-1]
    st.w_out, st.b_out = m.out.w.cpu().numpy()[:K], m.out.w.cpu().numpy()[K:K + 1]
    st.t, st.acc, st.adam = 0, np.zeros(V_c.shape[0], np.float32), {}
    return st


def test_dcn_criteo_d448_b65536(dev):
    """Forward logits and loss rows of a 4,096-sample slice against the oracle (a sample's forward does
    not depend on the rest of the batch); over the whole batch: padding columns zero in every stored
    activation and every weight gradient, finite loss = mean of the loss rows, and exactly the
    embedding rows the batch names moved."""
    from recommendation_amd.dcn import DCNv2
    B, NS = 65536, 4096
    m = DCNv2(CRITEO, 16, 13, dev, keep_grads=True)
    assert (m.d, m.dp) == (429, 448)
    ids, dense, labels = ofm.synthetic_batch(B, CRITEO, 13, seed=200, dist='uniform')
    crow, (ci,), glob = compact(CRITEO, [ids[:NS]])
    st = _dcn_state_from(m, crow, m.V[torch.from_numpy(glob).to(dev)].cpu().numpy())
    V_before = m.V.clone()
    idd = torch.from_numpy(ids).to(dev)
    loss = m.train_step(idd, torch.from_numpy(dense).to(dev), torch.from_numpy(labels).to(dev), check_ids=True)
    z_want = odcn.forward(st, ci, dense[:NS])
    per, sig = ofm.bce_with_logits(z_want, labels[:NS].astype(np.float64))
    e_z = rel(m.logit[:NS].cpu().numpy(), z_want)
    print("DCN d=448 B=65536: logits vs exact oracle on %d samples: %.2e" % (NS, e_z))
    assert e_z <= 1e-2
    assert rel(m.loss_rows[:NS].cpu().numpy(), per) <= 1e-2
    assert rel(m.g[:NS].cpu().numpy(), (sig - labels[:NS]) / B) <= 1e-2
    lr = m.loss_rows.double()
    assert torch.isfinite(lr).all() and abs(float(loss.item()) - float(lr.mean().item())) <= 1e-5 * float(lr.mean().item())
    d = m.d
    for x in m.x + m.u + m.dx:
        assert not x[:, d:].any().item()
    assert not m.dx0[:, d:].any().item()
    for p in m.Wc:
        assert not p.grad[d:, :].any().item() and not p.grad[:, d:].any().item()
        assert not p.w[d:, :].any().item() and not p.w[:, d:].any().item() and not p.h[d:, :].any().item()
        assert torch.isfinite(p.grad).all().item()
    for p in m.bc:
        assert not p.grad[d:].any().item() and not p.w[d:].any().item()
    assert not m.Wm[0].grad[:, d:].any().item() and not m.Wm[0].w[:, d:].any().item()
    for p in m.Wm + m.bm:
        assert torch.isfinite(p.grad).all().item() and p.grad.abs().max().item() > 0
    # embeddings: exactly the distinct rows of the batch moved (sorted-run index work at 1.7 M entries)
    toffs = m.toffs.cpu().numpy()
    uniq = np.unique(ids + toffs[:-1][None, :])
    moved = (m.V != V_before).any(dim=1).nonzero().view(-1).cpu().numpy()
    assert np.array_equal(moved, uniq)
    assert np.array_equal((m.V_acc != 0).nonzero().view(-1).cpu().numpy(), uniq)


def test_dcn_26_tables_d448_gradients_against_both_oracles(dev):
    """All 26 tables (cut to 64 rows each so the oracle's dense Adam stays small), d = 429 -> 448,
    the benchmark's MLP, batch 2048: every gradient against the storage-emulating oracle (1e-2) and
    against EXACT arithmetic (printed, <= 5e-2)."""
    from recommendation_amd.dcn import DCNv2
    rows, B = [64] * 26, 2048
    m = DCNv2(rows, 16, 13, dev, keep_grads=True)
    st = _dcn_state_from(m, rows, m.V.cpu().numpy())
    ids, dense, labels = ofm.synthetic_batch(B, rows, 13, seed=201, dist='zipf')
    _, gx, _, dVx, kx = odcn.grads(st, ids, dense, labels)
    loss_want, gr, uniq, dV, k = odcn.grads(st, ids, dense, labels, storage='bf16')
    loss = m.train_step(*[torch.from_numpy(a).to(dev) for a in (ids, dense, labels)], check_ids=True)
    assert abs(float(loss.item()) - loss_want) <= 1e-2 * abs(loss_want)
    d, got, worst = m.d, m.dense_params(), 0.0
    for name in gr:
        if name.startswith('out/'):
            continue
        G = got[name].grad.cpu().numpy()
        G = G[:gr[name].shape[0], :gr[name].shape[1]] if gr[name].ndim == 2 else G[:gr[name].shape[0]]
        e_s, e_x = rel(G, gr[name]), rel(G, gx[name])
        worst = max(worst, e_x)
        print("DCN d=448 %-9s vs storage-emulating oracle %.2e, vs exact oracle %.2e" % (name, e_s, e_x))
        assert e_s <= 1e-2 and e_x <= 5e-2, name
    # dx0 is a PER-SAMPLE tensor (nothing is summed over the batch): one ReLU gate that opens on one side
    # only moves a whole row of it, so the largest element error is a statement about single gates; the
    # bound is put on the error of the tensor as a whole (Frobenius norm), the largest element is printed
    got_dx0 = m.dx0.cpu().numpy()[:, :d].astype(np.float64)
    l2 = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))
    e_s, e_x = l2(got_dx0, k['dx0']), l2(got_dx0, kx['dx0'])
    # what bf16 STORAGE alone costs this tensor: the oracle with bf16 storage against the oracle without
    floor = l2(k['dx0'], kx['dx0'])
    print("DCN d=448 dx0       vs storage-emulating oracle %.2e, vs exact oracle %.2e (Frobenius; bf16 storage alone "
          "%.2e); largest element %.2e / %.2e" % (e_s, e_x, floor, rel(got_dx0, k['dx0']), rel(got_dx0, kx['dx0'])))
    # batch-summed gradients above average the gate flips out (<= 5e-2 asserted); a per-sample gradient cannot,
    # so it is held to what bf16 storage itself costs (+10 %), and to 1e-2 against the storage-emulating oracle
    assert e_s <= 1e-2 and e_x <= 1.1 * floor + 1e-3
    for p in m.Wc:
        assert not p.grad[d:, :].any().item() and not p.grad[:, d:].any().item()


def test_din_b8192_l100_1m_items(dev):
    """Index list of the sparse backward bit-exact over all 8192 x 101 positions, forward of a 256-sample
    slice against the exact oracle (1e-2), exactly the rows in the index list moved."""
    from recommendation_amd.din import DIN
    B, L, ROWS, NS = 8192, 100, 1000000, 256
    st = odin.State(ROWS, seed=9)
    st.W1, st.W2 = bf16_round(st.W1), bf16_round(st.W2)
    m = DIN(ROWS, dev, max_len=L)
    m.load_state(st)
    hist, qid, lens, labels = odin.synthetic_batch(B, L, ROWS, seed=300)
    hist[5, 0] = -1
    E_before = m.E.clone()
    t = [torch.from_numpy(a).to(dev) for a in (hist, qid, lens, labels)]
    loss = m.train_step(*t, check_ids=True)
    mask = (np.arange(L)[None, :] < lens[:, None]) & (hist >= 0)
    ids_want = np.concatenate([np.where(mask, hist, -1), qid[:, None]], axis=1).astype(np.int32)
    assert np.array_equal(m.ids_all.cpu().numpy(), ids_want)
    z_want, k = odin.forward(st, hist[:NS], qid[:NS], lens[:NS], keep=True)
    e_z = rel(m.logit[:NS].cpu().numpy(), z_want)
    print("DIN B=8192 L=100: logits vs exact oracle on %d samples: %.2e" % (NS, e_z))
    assert e_z <= 1e-2
    assert rel(m.feat[:NS, :64].cpu().numpy(), k['pooled']) <= 1e-2
    per, _ = ofm.bce_with_logits(z_want, labels[:NS].astype(np.float64))
    assert rel(m.loss_rows[:NS].cpu().numpy(), per) <= 1e-2
    lr = m.loss_rows.double()
    assert torch.isfinite(lr).all() and abs(float(loss.item()) - float(lr.mean().item())) <= 1e-5 * float(lr.mean().item())
    uniq = np.unique(ids_want[ids_want >= 0])
    moved = (m.E != E_before).any(dim=1).nonzero().view(-1).cpu().numpy()
    assert np.array_equal(moved, uniq)


def test_din_gradients_against_both_oracles_l100(dev):
    """History 100, batch 128: the unit's gradients against the storage-emulating oracle (1e-2) and
    against exact arithmetic (printed, <= 5e-2), factored and full-row forms."""
    from recommendation_amd.din import DIN
    B, L, ROWS = 128, 100, 5000
    for factored in (True, False):
        st = odin.State(ROWS, seed=3)
        st.W1, st.W2 = bf16_round(st.W1), bf16_round(st.W2)
        m = DIN(ROWS, dev, max_len=L, keep_grads=True, factored=factored)
        m.load_state(st)
        hist, qid, lens, labels = odin.synthetic_batch(B, L, ROWS, seed=301)
        _, gx, _, dEx, _ = odin.grads(st, hist, qid, lens, labels)
        _, gr, uniq, dE, _ = odin.grads(st, hist, qid, lens, labels, storage='bf16', factored=factored)
        m.train_step(*[torch.from_numpy(a).to(dev) for a in (hist, qid, lens, labels)], check_ids=True)
        got = m.dense_params()
        pairs = [('lau/W1', got['lau/W1'].grad.cpu().numpy()[:80]), ('lau/b1', got['lau/b1'].grad.cpu().numpy()[:80]),
                 ('lau/W2', got['lau/W2'].grad.cpu().numpy()[:40, :80]), ('lau/b2', got['lau/b2'].grad.cpu().numpy()[:40]),
                 ('lau/w3', got['lau/w3b3'].grad.cpu().numpy()[:40]), ('out/W', got['out/Wb'].grad.cpu().numpy()[:128])]
        for name, G in pairs:
            e_s, e_x = rel(G, gr[name]), rel(G, gx[name])
            print("DIN L=100 %s %-7s vs storage-emulating oracle %.2e, vs exact oracle %.2e"
                  % ('factored ' if factored else 'full rows', name, e_s, e_x))
            assert e_s <= 1e-2 and e_x <= 5e-2, (name, factored)
        rows = m.ids_all.cpu().numpy().reshape(-1)
        flat = m.dE.cpu().numpy().reshape(-1, 64)
        summed = np.zeros((ROWS, 64))
        np.add.at(summed, rows[rows >= 0], flat[rows >= 0])
        # per-row gradients sum a handful of positions, not the batch, so single ReLU gates show: the
        # tensor as a whole (Frobenius norm) is held to 1e-2 against the storage-emulating oracle and, against
        # exact arithmetic, to what bf16 storage itself costs (the two oracles against each other, +10 %);
        # the largest single element is printed
        l2 = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))
        e_s, e_x, floor = l2(summed[uniq], dE), l2(summed[uniq], dEx), l2(dE, dEx)
        print("DIN L=100 %s dE      vs storage-emulating oracle %.2e, vs exact oracle %.2e (Frobenius; bf16 storage "
              "alone %.2e); largest element %.2e / %.2e"
              % ('factored ' if factored else 'full rows', e_s, e_x, floor, rel(summed[uniq], dE), rel(summed[uniq], dEx)))
        assert e_s <= 1e-2 and e_x <= 1.1 * floor + 1e-3


@pytest.mark.parametrize("opt", ["sgd", "adagrad_rowwise"])
def test_embedding_bags_dim128_b65536_one_gpu_share(dev, opt):
    """The embedding kernels of the sharded-table configuration at one GPU's shapes: 26 tables x dim 128, batch
    65536 (1.7 M entries: the global radix sort and the wide-row update paths), tables cut to 26 x 400 k rows
    (5.3 GB) so that the test allocates in seconds - the kernels' paths depend on dim, entries per table and
    duplicates, not on the table's height.  Gather: bit-exact against indexing the table.  Sort artefacts: exact
    against numpy's stable argsort.  Update: the oracle on the compacted rows, and no other row moves."""
    from recommendation_amd import ops
    T, D, B = 26, 128, 65536
    rows = [400000] * T
    total = sum(rows)
    toffs_h = np.concatenate([[0], np.cumsum(rows)]).astype(np.int64)
    toffs = torch.from_numpy(toffs_h).to(dev)
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    W = torch.randn((total, D), generator=g, dtype=torch.float32, device=dev) * 0.05
    W0 = W.clone()
    ids = torch.randint(0, rows[0], (B, T), generator=g, dtype=torch.int64, device=dev)
    ids[:, 3] = torch.randint(0, 7, (B,), generator=g, dtype=torch.int64, device=dev)      # a hot table: runs of ~9 k
    ids[5, 7] = -1                                                                         # one padding id
    grad = torch.randn((B, T * D), generator=g, dtype=torch.float32, device=dev) * 1e-2
    err = ops.new_error_word(dev)
    out = ops.embag_fwd(W, toffs, T, ids, bag_len=1, check_flags=ops.MI_CHECK_RANGE, err=err)
    ops.check_error_word(err)
    glob = (ids + toffs[:T].view(1, T)).clamp_(min=0)
    want = W[glob.view(-1)].view(B, T, D)
    want[5, 7] = 0
    assert torch.equal(out.view(B, T, D), want)
    plan = ops.SortedPlan(toffs, T, total, ids, bag_len=1, runs=True)
    keys_h = (ids.cpu().numpy() + toffs_h[:T][None, :]).reshape(-1)
    keys_h[5 * T + 7] = total                                                              # padding sorts last
    order = np.argsort(keys_h, kind='stable')
    srows, perm, uniq, seg = plan.host()
    n_valid = B * T - 1
    assert np.array_equal(srows, keys_h[order][:n_valid]) and np.array_equal(perm, order[:n_valid])
    assert np.array_equal(uniq, np.unique(keys_h[order][:n_valid]))
    plan2 = ops.SortedPlan(toffs, T, total, ids, bag_len=1, runs=False)
    state = torch.full((total,), 0.1, dtype=torch.float32, device=dev)
    lr, eps = 0.05, 1e-8
    if opt == 'sgd':
        ops.bwd_apply_sgd(plan2, grad, W, lr)
    else:
        ops.bwd_apply_adagrad_rowwise(plan2, grad, W, state, lr, eps)
    # oracle on the touched rows: per-row gradient sums in fp64
    touched = torch.from_numpy(uniq.astype(np.int64)).to(dev)
    gsum = torch.zeros((total, D), dtype=torch.float64, device=dev)
    flat_rows = torch.from_numpy(keys_h).to(dev)
    valid = flat_rows < total
    gsum.index_add_(0, flat_rows[valid], grad.view(B * T, D).double()[valid])
    gs = gsum[touched]
    w_old = W0[touched].double()
    if opt == 'sgd':
        w_want = w_old - lr * gs
    else:
        st = 0.1 + (gs * gs).mean(dim=1, keepdim=True)
        w_want = w_old - lr * gs / (st.sqrt() + eps)
        assert rel(state[touched].cpu().numpy(), st.view(-1).cpu().numpy()) <= 1e-5
    assert rel(W[touched].cpu().numpy(), w_want.cpu().numpy()) <= 1e-5
    moved = (W != W0).any(dim=1)
    untouched = torch.ones(total, dtype=torch.bool, device=dev)
    untouched[touched] = False
    assert not bool((moved & untouched).any())
