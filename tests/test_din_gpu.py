"""DIN local activation unit (bf16 contractions + gather / pool / scatter kernels) against the numpy
restatement.  Forward against exact arithmetic within 1e-2; gradients against the restatement that
rounds stored tensors to bf16 where the device does (see tests/test_dcn_gpu.py for why); the index
list of the sparse backward bit-exact."""
import numpy as np
import pytest
import torch

from oracle import din as odin
from oracle.dcn import bf16_round

pytestmark = pytest.mark.gpu

ROWS = 5000


def _state(seed):
    st = odin.State(ROWS, seed=seed)
    st.W1, st.W2 = bf16_round(st.W1), bf16_round(st.W2)
    return st


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def _to(dev, *arrs):
    return [torch.from_numpy(a).to(dev) for a in arrs]


FACTORED = pytest.mark.parametrize("factored", [False, True], ids=["full_rows", "factored"])


@FACTORED
@pytest.mark.parametrize("B,L", [(64, 100), (32, 20)])
def test_forward(dev, B, L, factored):
    from recommendation_amd.din import DIN
    st = _state(1)
    m = DIN(ROWS, dev, max_len=L, factored=factored)
    m.load_state(st)
    hist, qid, lens, labels = odin.synthetic_batch(B, L, ROWS, seed=2)
    hist[3, 1] = -1                                          # a padding id inside the live range
    lens[0], lens[1] = L, 1
    z = m.forward(*_to(dev, hist, qid, lens, labels))
    want, k = odin.forward(st, hist, qid, lens, keep=True)
    # the index list of the backward: live history ids, -1 elsewhere, then the candidate id
    ids_want = np.concatenate([np.where(k['mask'], hist, -1), qid[:, None]], axis=1).astype(np.int32)
    assert np.array_equal(m.ids_all.cpu().numpy(), ids_want)
    # A rows: gathered rows combined in fp32 and rounded once
    A_want = odin.forward(st, hist, qid, lens, keep=True, storage='bf16', dtype=np.float32, factored=factored)[1]['A']
    assert np.array_equal(m.A.float().cpu().numpy(), A_want)
    assert rel(m.a.cpu().numpy().reshape(B, L) * k['mask'], k['a'] * k['mask']) <= 1e-2
    assert rel(m.feat.cpu().numpy()[:, :64], k['pooled']) <= 1e-2
    assert np.array_equal(m.feat.cpu().numpy()[:, 64:], st.E[qid])
    assert rel(z.cpu().numpy(), want) <= 1e-2


@FACTORED
def test_gradients_and_update(dev, factored):
    from recommendation_amd.din import DIN
    B, L = 64, 100
    st = _state(3)
    m = DIN(ROWS, dev, max_len=L, keep_grads=True, factored=factored)
    m.load_state(st)
    hist, qid, lens, labels = odin.synthetic_batch(B, L, ROWS, seed=4)
    hist[:, :3] = hist[0, :3]                                # rows shared by every sample: long runs
    loss_exact, gx = odin.grads(st, hist, qid, lens, labels)[:2]     # exact arithmetic: printed, bounded by 5e-2
    loss_want, gr, uniq, dE, k = odin.grads(st, hist, qid, lens, labels, storage='bf16', factored=factored)
    assert abs(loss_want - loss_exact) <= 1e-2 * abs(loss_exact)
    E_before = m.E.clone()
    loss = m.train_step(*_to(dev, hist, qid, lens, labels), check_ids=True)
    assert abs(float(loss.item()) - loss_want) <= 1e-2 * abs(loss_want)
    got = m.dense_params()
    assert rel(got['lau/W1'].grad.cpu().numpy()[:80], gr['lau/W1']) <= 1e-2
    assert not got['lau/W1'].grad.cpu().numpy()[80:].any()
    assert rel(got['lau/b1'].grad.cpu().numpy()[:80], gr['lau/b1']) <= 1e-2
    assert rel(got['lau/W2'].grad.cpu().numpy()[:40, :80], gr['lau/W2']) <= 1e-2
    assert rel(got['lau/b2'].grad.cpu().numpy()[:40], gr['lau/b2']) <= 1e-2
    w3 = got['lau/w3b3'].grad.cpu().numpy()
    for name, G in (('lau/W1', got['lau/W1'].grad.cpu().numpy()[:80]), ('lau/b1', got['lau/b1'].grad.cpu().numpy()[:80]),
                    ('lau/W2', got['lau/W2'].grad.cpu().numpy()[:40, :80]), ('lau/b2', got['lau/b2'].grad.cpu().numpy()[:40]),
                    ('lau/w3', w3[:40])):
        ex = rel(G, gx[name])
        print("%s gradient vs the EXACT oracle: %.2e" % (name, ex))
        assert ex <= 5e-2, name
    assert rel(w3[:40], gr['lau/w3']) <= 1e-2
    assert abs(w3[64] - gr['lau/b3'][0]) <= 1e-2 * max(abs(gr['lau/b3'][0]), np.abs(gr['lau/w3']).max())
    head = got['out/Wb'].grad.cpu().numpy()
    assert rel(head[:128], gr['out/W']) <= 1e-2
    assert abs(head[128] - gr['out/b'][0]) <= 1e-5 + 1e-2 * abs(gr['out/b'][0])
    # embedding rows: exactly the rows in the index list moved, by row-wise Adagrad of the summed gradient
    moved = (m.E != E_before).any(dim=1).cpu().numpy()
    assert np.array_equal(np.nonzero(moved)[0], uniq)
    acc = (dE * dE).mean(axis=1)
    want_E = st.E[uniq] - 1e-2 * dE / (np.sqrt(acc) + 1e-8)[:, None]
    assert np.abs(m.E.cpu().numpy()[uniq] - want_E).max() <= 2e-2 * 1e-2 * np.sqrt(64)
    # the per-position gradients themselves
    D = 64
    got_dE = m.dE.cpu().numpy().reshape(B, L + 1, D)
    rows = m.ids_all.cpu().numpy()
    flat_rows, flat_g = rows.reshape(-1), got_dE.reshape(-1, D)
    summed = np.zeros((ROWS, D))
    np.add.at(summed, flat_rows[flat_rows >= 0], flat_g[flat_rows >= 0])
    assert rel(summed[uniq], dE) <= 1e-2


@FACTORED
def test_training_follows_the_oracle(dev, factored):
    from recommendation_amd.din import DIN
    B, L = 32, 20
    st = _state(5)
    m = DIN(ROWS, dev, max_len=L, factored=factored)
    m.load_state(st)
    hist, qid, lens, labels = odin.synthetic_batch(B, L, ROWS, seed=6)
    t = _to(dev, hist, qid, lens, labels)
    got, want = [], []
    for _ in range(6):
        got.append(float(m.train_step(*t).item()))
        want.append(float(odin.train_step(st, hist, qid, lens, labels, storage='bf16', factored=factored)))
    assert want[-1] < want[0]
    assert np.allclose(got, want, rtol=1e-2)


def test_out_of_range_id_is_reported(dev):
    from recommendation_amd.din import DIN
    from recommendation_amd.ops import InvalidArgumentError
    m = DIN(ROWS, dev, max_len=20)
    hist, qid, lens, labels = odin.synthetic_batch(32, 20, ROWS, seed=7)
    lens[5] = 20
    hist[5, 2] = ROWS
    m.forward(*_to(dev, hist, qid, lens, labels))
    with pytest.raises(InvalidArgumentError):
        from recommendation_amd import ops
        ops.check_error_word(m.err, "DIN ids")
