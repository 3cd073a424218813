"""The colour model's data-parallel step replayed from HIP graphs: two ranks (two processes sharing the one GPU,
collectives over gloo, which stages device tensors through the host) each run ``GraphedTrainStep`` with a process
group - {forward, backward} graph, all-reduce of the flat gradient, {schedule, Adam} graph - beside a twin model
stepping eagerly through ``train_step(process_group=...)``.  Same kernels, same collective: bit-identical results on
each rank, and the two ranks' replicas bit-identical to one another."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rank_main(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import colour_model as cm
        from recommendation_amd.r_model import RecommendationModel
        from recommendation_amd.trainer import AdamOptimizer, GraphedTrainStep, train_step
        dev = torch.device("cuda", 0)
        params = cm.init_params(seed=21, dtype=np.float32)

        def model():
            m = RecommendationModel(colour_count=128, recommend_num=6, user_count=100, country_count=20, device=dev)
            m.load_variables(params)
            return m

        a, b = model(), model()
        opt_a, opt_b = AdamOptimizer(1e-3), AdamOptimizer(1e-3)
        batches = [tuple(torch.from_numpy(v).to(dev) for v in cm.synthetic_batch(300, seed=100 * rank + i).values())
                   for i in range(4)]
        ragged = tuple(torch.from_numpy(v).to(dev) for v in cm.synthetic_batch(77, seed=900 + rank).values())
        g = GraphedTrainStep(a, opt_a, batches[0], check_every=2, process_group=dist.group.WORLD)
        assert g.world == world
        same = True
        for cols in (batches[0], batches[1], ragged, batches[2], batches[3]):
            ra = g(cols)
            rb = train_step(b, opt_b, cols, process_group=dist.group.WORLD)
            same &= ra[4] == rb[4] and torch.equal(ra[3], rb[3]) and torch.equal(ra[2], rb[2])
            same &= torch.equal(a.parameters().detach(), b.parameters().detach())
        same &= torch.equal(opt_a.m, opt_b.m) and torch.equal(opt_a.v, opt_b.v)
        g.check()
        np.savez(os.path.join(out_dir, "rank%d.npz" % rank), same=np.array(bool(same)),
                 params=a.parameters().detach().cpu().numpy(), start=np.concatenate([v.reshape(-1) for v in params.values()]))
    finally:
        dist.destroy_process_group()


def test_data_parallel_step_from_two_graphs_equals_the_eager_step_on_two_ranks(dev, tmp_path):
    world = 2
    mp.spawn(_rank_main, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    got = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % r)) for r in range(world)]
    assert bool(got[0]["same"]) and bool(got[1]["same"])              # graphs == eager, on both ranks
    assert np.array_equal(got[0]["params"], got[1]["params"])          # replicas: the same averaged gradient everywhere
    assert got[0]["params"].size >= 5910 and np.isfinite(got[0]["params"]).all()   # (the arena pads every variable to 16 bytes)
    assert not np.array_equal(got[0]["params"][:64], got[0]["start"][:64])          # and the parameters did move
