"""DCN-v2 with row-wise sharded tables and a data-parallel interaction network: the ranks' models driven in lockstep
on ONE GPU (every exchange a tensor copy, the all-reduce a sum in rank order, kernels the real ones) against the
single-GPU DCNv2 on the union of the ranks' batches - the same bf16 arithmetic in another summation order: 3e-4 of each tensor's largest magnitude -
and the replicas' dense parameters bit-identical to one another; the real collectives at world size 1 over RCCL."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from errlog import held  # noqa: E402
from oracle import dcn as odcn  # noqa: E402
from oracle import fm as ofm  # noqa: E402

pytestmark = pytest.mark.gpu

ROWS = [1460, 583, 3001, 305, 24, 3, 10, 4, 18, 15, 105, 27, 2173, 12, 50, 7]
DIM, ND, MLP = 16, 13, (128, 64)
TOL = 3e-4            # twice what is observed (bf16 storage on both sides; only the order of the sums differs)


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def _state(seed):
    st = odcn.State(ROWS, DIM, ND, n_cross=2, mlp=MLP, seed=seed)
    r = lambda a: torch.from_numpy(a).to(torch.bfloat16).float().numpy()  # noqa: E731
    st.Wc, st.Wm = [r(w) for w in st.Wc], [r(w) for w in st.Wm]
    return st


def _models(dev, st, N):
    from recommendation_amd.sharded_dcn import ShardedDCN
    ms = []
    for rnk in range(N):
        m = ShardedDCN(ROWS, DIM, ND, dev, rank=rnk, world=N, n_cross=2, mlp=MLP, lr=1e-3, emb_lr=1e-2, capacity_factor=4.0)
        m.engine.net.load_state(st)
        m.bag.load_weights(st.V)
        ms.append(m)
    return ms


def _batches(N, B, step, dev):
    out = []
    for rnk in range(N):
        ids, dense, labels = ofm.synthetic_batch(B, ROWS, ND, seed=1000 * step + rnk, dist='zipf')
        out.append((torch.from_numpy(ids).to(dev), torch.from_numpy(dense).to(dev), torch.from_numpy(labels).to(dev)))
    return out


@pytest.mark.parametrize("N", [2, 4])
def test_lockstep_ranks_equal_the_single_gpu_model_on_the_union_of_their_batches(dev, N):
    from recommendation_amd.dcn import DCNv2
    from recommendation_amd.sharded_dcn import lockstep_dcn_step
    B = 256
    st = _state(3)
    one = DCNv2(ROWS, DIM, ND, dev, n_cross=2, mlp=MLP, lr=1e-3, emb_lr=1e-2)
    one.load_state(st)
    ms = _models(dev, st, N)
    for step in range(2):
        bs = _batches(N, B, step, dev)
        losses = lockstep_dcn_step(ms, bs)
        union = tuple(torch.cat([b[i] for b in bs]) for i in range(3))
        l1 = float(one.train_step(*union).item())
        lN = float(torch.stack(losses).mean().item())            # equal batch sizes: the global mean loss
        held("sharded dcn N=%d step %d loss" % (N, step), abs(lN - l1) / abs(l1), TOL)
    # the replicas never diverge: the same summed gradient, the same update, on every rank
    for m in ms[1:]:
        assert torch.equal(m.engine.net.w_arena, ms[0].engine.net.w_arena)
        assert torch.equal(m.engine.net.m_arena, ms[0].engine.net.m_arena)
    for name, p in one.dense_params().items():
        q = ms[0].engine.net.dense_params()[name]
        held("sharded dcn N=%d %s" % (N, name), rel(q.w.cpu().numpy(), p.w.cpu().numpy()), TOL)
    full = np.empty((sum(ROWS), DIM), np.float32)
    for rnk, m in enumerate(ms):
        full[rnk::N] = m.bag.be.W[:m.bag.be.local_rows].cpu().numpy()
    held("sharded dcn N=%d V" % N, rel(full, one.V.cpu().numpy()), TOL)
    moved = np.abs(full - st.V).max()
    assert moved > 1e-4                                            # the rows did take their updates


def test_one_rank_over_rccl_equals_the_lockstep_driver_bitwise(dev):
    """World size 1 through the library's own communicator (mi_comm_*, all-to-all and all-gather over one rank):
    train_step and the lockstep driver leave the same bits."""
    from recommendation_amd.sharded import RcclComm
    from recommendation_amd.sharded_dcn import ShardedDCN, lockstep_dcn_step
    st = _state(4)
    a = ShardedDCN(ROWS, DIM, ND, dev, comm=RcclComm(0, 1), n_cross=2, mlp=MLP)
    b = ShardedDCN(ROWS, DIM, ND, dev, rank=0, world=1, n_cross=2, mlp=MLP)
    for m in (a, b):
        m.engine.net.load_state(st)
        m.bag.load_weights(st.V)
    for step in range(2):
        bs = _batches(1, 256, step, dev)
        la = a.train_step(*bs[0], check=True).clone()
        lb = lockstep_dcn_step([b], bs)[0]
        assert torch.equal(la, lb)
    assert torch.equal(a.engine.net.w_arena, b.engine.net.w_arena)
    assert torch.equal(a.bag.be.W, b.bag.be.W)


def test_the_network_without_a_table_asks_for_the_pooled_rows(dev):
    from recommendation_amd import ops
    from recommendation_amd.dcn import DCNv2
    m = DCNv2(ROWS, DIM, ND, dev, mlp=MLP, embeddings=False, grads_only=True)
    ids, dense, labels = (torch.from_numpy(x).to(dev) for x in ofm.synthetic_batch(64, ROWS, ND, seed=1))
    with pytest.raises(ops.MiError):
        m.forward(ids, dense, labels)
