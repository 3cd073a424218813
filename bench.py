#!/usr/bin/env python3
"""Benchmark of the hot path: FM training on synthetic Criteo-shaped data.

  python bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path over one batch already resident in HBM: fused
gather + FM interaction + loss forward, deterministic sort-by-row, fused sparse
backward/optimizer, dense-side update.  N = 1: 26 tables (Criteo-Kaggle cardinalities,
33.76 M rows) x dim 16 fp32, batch 8192, 13 dense features.  N > 1 (launched by
torch.distributed.run, one rank per GPU): tables sharded row-wise over the ranks, batch
8192 PER RANK (weak scaling), ids and pooled partial sums exchanged by RCCL all-to-all.

Rank 0 prints ONE JSON line.  Besides the contract keys it carries
  roofline     the dominant HBM kernel: algorithmic bytes / measured kernel time vs 8 TB/s
  stages       every stage of the step with its time and algorithmic bytes
  cpu_baseline the numpy oracle of the same step timed on this host's cores (rank 0, N=1)
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH: 8.0 TB/s spec, ~6.3 TB/s achievable)

CRITEO_KAGGLE = [1460, 583, 10131227, 2202608, 305, 24, 12517, 633, 3, 93145, 5683, 8351593, 3194, 27, 14992,
                 5461306, 10, 5652, 2173, 4, 7046547, 18, 15, 286181, 105, 142572]


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--batch', type=int, default=8192, help='batch per GPU')
    ap.add_argument('--dim', type=int, default=16)
    ap.add_argument('--opt', default=None, choices=['sgd', 'adagrad_rowwise', 'adam_tf'],
                    help='table optimizer (default adam_tf, the form of the reference\'s tf.train.AdamOptimizer)')
    ap.add_argument('--dist', default='uniform', choices=['uniform', 'zipf'])
    ap.add_argument('--idx-bits', type=int, default=64, choices=[32, 64])
    ap.add_argument('--seed', type=int, default=0, help='fm workload: seed of the synthetic batches (rank r draws from 1000 + seed + r)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-steps', type=int, default=50)
    ap.add_argument('--stage-reps', type=int, default=50)
    ap.add_argument('--no-graph', action='store_true', help='launch the step eagerly instead of replaying a HIP graph')
    ap.add_argument('--prefetch', action='store_true',
                    help='sharded path: route and exchange the ids of the next batch on a side stream beside this step\'s update; '
                         'the default with more than one rank (measured slower at world size 1: the extra stream hops cost more '
                         'than the exchange they hide)')
    ap.add_argument('--no-prefetch', action='store_true', help='sharded path with several ranks: no next-batch id exchange beside the update')
    ap.add_argument('--comm', choices=['torch', 'rccl'], default='torch',
                    help="sharded paths: collectives through torch.distributed (default) or the library's own RCCL entry points")
    ap.add_argument('--din-full-rows', action='store_true',
                    help='DIN: materialise the full [q | k | q-k | q*k] rows instead of the factored first layer')
    ap.add_argument('--prefetch-sort', action='store_true',
                    help='single-GPU step: sort the next batch\'s ids on a side stream beside this step instead of inside '
                         'the next one (bit-identical; measured slower at B = 8192: the stream hops and the third launch '
                         'cost more than the sort they hide)')
    ap.add_argument('--steps-per-graph', type=int, default=32,
                    help='fm workload: consecutive steps captured in one HIP graph (rounded down to whole rounds of the 8 resident batches)')
    ap.add_argument('--single-step-graphs', action='store_true',
                    help='single-GPU step: one graph replay per step (default: one replay per --steps-per-graph consecutive steps)')
    ap.add_argument('--ids-sample-major', action='store_true',
                    help='feed the single-GPU step ids as [B][T] only (default: also one [B] row per field)')
    ap.add_argument('--force-sharded', action='store_true',
                    help='use the row-wise sharded model and its collectives even with one rank (rehearsal of the N > 1 path)')
    ap.add_argument('--bag-len', type=int, default=1, help='embag workload: ids per bag (sum pooling)')
    ap.add_argument('--embag-rows', type=int, default=1000000000 // 8,
                    help='embag workload: total rows over the 26 tables (default: one GPU\'s eighth of 1 G rows); '
                         'sharded_embag: rows PER RANK')
    ap.add_argument('--din-items', type=int, default=8000000, help='din workload: rows of the item table')
    ap.add_argument('--id-exchange', choices=['all_gather', 'bucketed'], default='all_gather',
                    help='sharded_embag workload: all-gather of the ids, or an all-to-all of the ids bucketed by owner')
    ap.add_argument('--no-host-fed', action='store_true',
                    help='fm workload: skip the extra line that feeds every batch from pinned host memory')
    ap.add_argument('--workload', default='fm', choices=['fm', 'dcn', 'din', 'infer', 'embag', 'sharded_embag', 'colour'],
                    help='fm (default, the headline configuration); dcn / din: the bf16 matrix-core configurations, 1 GPU')
    args = ap.parse_args()
    if args.opt is None:
        args.opt = 'adam_tf'
    return args


def gen_batches(n_batches, B, rows, n_dense, dist, idx_bits, dev, seed):
    """Synthetic batches generated on the device (torch RNG; shape and distribution as
    oracle.fm.synthetic_batch): uniform ids or Zipf(1.05) ranks, N(0,1) dense, Bernoulli labels."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    card = torch.tensor(rows, dtype=torch.float64, device=dev)
    out = []
    for _ in range(n_batches):
        u = torch.rand((B, len(rows)), generator=g, dtype=torch.float64, device=dev)
        if dist == 'uniform':
            ids = torch.floor(u * card)
        else:
            a = 1.0 - 1.05
            ids = torch.floor(((u * ((card + 1.0) ** a - 1.0)) + 1.0) ** (1.0 / a)) - 1.0
        ids = torch.minimum(ids.clamp_(min=0), card - 1).to(torch.int64 if idx_bits == 64 else torch.int32)
        dense = torch.randn((B, n_dense), generator=g, dtype=torch.float32, device=dev)
        labels = (torch.rand((B,), generator=g, device=dev) < 0.25).to(torch.float32)
        out.append((ids.contiguous(), dense, labels))
    return out


def time_stage(fn, reps):
    """Average device time of `fn` in ms: HIP events recorded on the stream the kernels are
    launched on (torch's current stream), `reps` back-to-back calls between them."""
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def time_rotating(fn, n, reps):
    """time_stage of fn(j) with j cycling over n prepared inputs (whole cycles only, the first cycle
    untimed): with enough distinct inputs that more than the 256 MiB last-level cache is touched
    between two uses of a line, the rate that comes out is an HBM rate, not a cache rate."""
    for j in range(n):
        fn(j)
    it = [0]

    def call():
        fn(it[0] % n)
        it[0] += 1

    reps = -(-max(reps, n) // n) * n
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        call()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def sharded_stage_report(m, batch, reps, idx_bits):
    """Per-launch device time and ALGORITHMIC bytes of one rank's share of a sharded FM step, timed on
    the buffers the last step left behind (no collectives inside: they are reported by the step time)."""
    ids, dense, labels = batch
    B, T = ids.shape
    N, D, nd = m.world, m.dim, m.n_dense
    ib = idx_bits // 8
    n = int(m.n_total.item())
    U = int(torch.unique(m.sorted_rows[:n]).numel())
    blk = m.send.shape[1]
    n_state = {'sgd': 0, 'adagrad_rowwise': 1.0 / D, 'adam_tf': 2}[m.opt]
    n_state_w1 = {'sgd': 0, 'adagrad_rowwise': 1, 'adam_tf': 2}[m.opt]
    stages = []

    def add(name, fn, nbytes):
        ms = time_stage(fn, reps)
        stages.append(dict(name=name, ms=round(ms, 5), kernels=1, alg_bytes=int(nbytes), gbs=round(nbytes / ms / 1e6, 1)))

    add('launch 1: route (ids -> per-owner slot blocks)', lambda: m.phase_route(ids), B * T * ib + N * blk * 4)
    add('launch 2: owner forward + per-table LDS sort', lambda: m.phase_owner_forward(),
        N * blk * 4 + n * (D * 4 + 4) + N * B * (D + 2) * 4 + n * 8)
    add('launch 3: combine (owners\' partials -> logit, loss, X rows)', lambda: m.phase_combine(dense, labels),
        N * B * (D + 2) * 4 + B * nd * 4 + m.rs * m.ld * 4)
    add('launch 4: owner sparse update (segmented sum + fused %s) + dense side' % m.opt, lambda: m.phase_update(),
        n * (8 + (D + 1) * 4) + U * D * 4 * 2 * (1 + n_state) + U * 4 * 2 * (1 + n_state_w1) + N * (m.rs - B) * m.ld * 4)
    return stages, n, U


def stage_report(fm, batches, reps, idx_bits, tms=None):
    """Per-stage device time and ALGORITHMIC bytes of one FM step.  Every stage is timed cycling over
    ALL the bench's batches, each with its own forward outputs and sorted plan prepared beforehand: a
    batch touches ~94 k records (24 MB) + 14 MB of S / permutation, eight of them ~0.5 GB, so no line
    is still in the 256 MiB last-level cache when its batch comes round again.  The update stages run
    with step size 0 and the model's parameters and optimizer state are restored afterwards."""
    from recommendation_amd import ops
    from recommendation_amd.fm import _OPT
    B, T = batches[0][0].shape
    D, nd = fm.dim, fm.n_dense
    ib = idx_bits // 8
    n_b = len(batches)
    fm._buffers(batches[0][0])                 # the model's own loss / buffer set exist before the first step too
    saved = (fm.table.rec.clone(), fm.dp.clone(), fm.dm.clone(), fm.dv.clone(), fm.sched.clone())
    f32 = dict(dtype=torch.float32, device=fm.device)
    fused = fm._fused_ok(B, tms[0] if tms else None)
    sets, Us = [], []
    for j, (ids, dense, labels) in enumerate(batches):
        buf = dict(logit=torch.empty((B,), **f32), S=torch.empty((B, D), **f32), loss_rows=torch.empty((B,), **f32),
                   g=torch.empty((B,), **f32))
        plan = ops.SortedPlan(fm.toffs, T, fm.total_rows, ids, bag_len=1, runs=False)
        ops.fm_fwd(fm.table, fm.toffs, ids, dense if nd else None, fm.wd if nd else None, fm.bias, labels, 1.0 / B,
                   fm.err, out=buf)
        sets.append((buf, plan))
        Us.append(int(torch.unique(plan.sorted_rows).numel()))
    U = sum(Us) / float(n_b)
    n_state = {'sgd': 0, 'adagrad_rowwise': 1.0 / D, 'adam_tf': 2}[fm.opt]
    n_state_w1 = {'sgd': 0, 'adagrad_rowwise': 1, 'adam_tf': 2}[fm.opt]
    # forward = rows + ids + first-order weights + dense features + logit
    fwd_bytes = B * T * (D * 4 + ib) + B * T * 4 + B * nd * 4 + B * 4
    # backward/update = per entry {S row, perm, g}, then read-modify-write of the U distinct rows and their state
    bwd_bytes = (B * T * (D * 4 + 8) + U * D * 4 * 2 * (1 + n_state) + U * 4 * 2 * (1 + n_state_w1) + B * 4)
    bits = int(fm.total_rows).bit_length()
    passes = -(-bits // 9)
    sort_bytes = B * T * (ib + 8) + passes * B * T * 20
    stages = []

    def add(name, fn, nbytes, kernels):
        ms = time_rotating(fn, n_b, reps)
        stages.append(dict(name=name, ms=round(ms, 5), kernels=kernels, alg_bytes=int(nbytes),
                           gbs=round(nbytes / ms / 1e6, 1)))

    def nd_arg(j):
        return batches[j][1] if nd else None

    def fwd_sort(j):
        ids, dense, labels = batches[j]
        ops.fm_fwd_sort(fm.table, fm.toffs, ids, sets[j][1], nd_arg(j), fm.wd if nd else None, fm.bias, labels,
                        1.0 / B, fm.err, out=sets[j][0], ids_tm=tms[j] if tms else None)

    def upd_dense(j):
        buf, plan = sets[j]
        ops.fm_bwd_apply_dense(plan, fm.table, buf['g'], buf['S'], _OPT[fm.opt], 0.0, fm.beta1, fm.beta2, fm.eps,
                               nd_arg(j), buf['loss_rows'], fm.dp, fm.dm, fm.dv, fm.loss)

    def fwd(j):
        ids, dense, labels = batches[j]
        ops.fm_fwd(fm.table, fm.toffs, ids, nd_arg(j), fm.wd if nd else None, fm.bias, labels, 1.0 / B, fm.err,
                   out=sets[j][0])

    def upd(j):
        buf, plan = sets[j]
        ops.fm_bwd_apply(plan, fm.table, buf['g'], buf['S'], _OPT[fm.opt], 0.0, fm.beta1, fm.beta2, fm.eps)

    def dense_side(j):
        buf = sets[j][0]
        ops.fm_dense_step(buf['g'], nd_arg(j), buf['loss_rows'], fm.dp, fm.dm, fm.dv, 0.0, fm.beta1, fm.beta2, fm.eps,
                          fm.loss)

    if fused:
        # the two launches the step is made of (timed without advancing the schedule / with step size 0)
        # (the sort of launch 1 runs in LDS: the ids it reads are the forward's and its sorted rows + permutation
        # are the only bytes it adds - 8 per entry; the radix-pass scratch counted for the stand-alone sort never
        # reaches memory here)
        add('launch 1: forward + per-table LDS sort', fwd_sort, fwd_bytes + B * T * 8, 1)
        add('launch 2: sparse update (segmented sum + fused %s) + dense side' % fm.opt, upd_dense,
            bwd_bytes + B * (nd + 2) * 4, 1)
    # the members on their own
    add('fm_fwd (gather+pool+interaction+loss)', fwd, fwd_bytes, 1)
    lds_sort = B <= ops.fm_fwd_sort_max_batch()
    one = B <= ops.fm_lds_sort_single_max()
    add('sort_by_row (%s)' % (('per-table LDS sort, 1 launch' if one else 'ids to field-major + per-table LDS sort in parts')
                              if lds_sort else 'keygen + %d radix passes' % passes),
        lambda j: sets[j][1].prepare(batches[j][0]), sort_bytes, (1 if one else 2) if lds_sort else 1 + 3 * passes)
    add('fm_bwd_apply (segmented sum + fused %s)' % fm.opt, upd, bwd_bytes, 1)
    add('dense side (dwd/dbias + adam + mean loss)', dense_side, B * (nd + 2) * 4, 1)
    for dst, src in zip((fm.table.rec, fm.dp, fm.dm, fm.dv, fm.sched), saved):
        dst.copy_(src)
    return stages, U


def host_fed_line(model, batches, steps, dev, resident_ms):
    """The same step with every batch copied in from pinned host memory, as the reference's loop reads its batch
    every step (recommendation/local_train_utils.py:34-41,60-63): int32 ids field-major + dense + labels go over
    PCIe as one block on a side stream into one of two device buffer sets while the step of the previous batch runs on the other
    (events order copy -> step -> next copy into that set).  Returns samples/s of that pipeline, of the copies
    alone, and which of the two bounds it."""
    from recommendation_amd import ops
    B, T = batches[0][0].shape
    nb = len(batches)
    nd = batches[0][1].shape[1]
    # ONE pinned block per batch - ids | dense | labels at 256-byte aligned offsets - so a batch is one copy: three
    # separate copies of 852 KB, 426 KB and 32 KB cost ~70 us of call overhead each and moved 6 GB/s
    sizes = [T * B * 4, B * nd * 4, B * 4]
    offs = [0, -(-sizes[0] // 256) * 256]
    offs.append(offs[1] + -(-sizes[1] // 256) * 256)
    total = offs[2] + -(-sizes[2] // 256) * 256
    nbytes = sum(sizes)

    def views(block):
        return (block[offs[0]:offs[0] + sizes[0]].view(torch.int32).view(T, B),
                block[offs[1]:offs[1] + sizes[1]].view(torch.float32).view(B, nd),
                block[offs[2]:offs[2] + sizes[2]].view(torch.float32).view(B))

    host = []
    for b in batches:
        blk = torch.empty((total,), dtype=torch.uint8).pin_memory()
        v = views(blk)
        v[0].copy_(b[0].t().to(torch.int32))
        v[1].copy_(b[1])
        v[2].copy_(b[2])
        host.append(blk)
    shape_only = torch.empty((B, T), dtype=torch.int32, device=dev)      # the step reads the field-major copy
    dblk = [torch.empty((total,), dtype=torch.uint8, device=dev) for _ in range(2)]
    bufs = [views(d) for d in dblk]
    side = torch.cuda.Stream(dev)
    copied = [torch.cuda.Event() for _ in range(2)]
    used = [torch.cuda.Event() for _ in range(2)]
    model.train_step(shape_only, bufs[0][1], bufs[0][2], ids_tm=bufs[0][0])     # buffers of this shape exist before capture
    torch.cuda.synchronize()
    # One replay = the run's nb batches: copy of batch i+1 (side stream) beside the step on batch i (capturing stream),
    # ordered by events inside the graph.  Issued eagerly the same pipeline is bound by the host (~60 us of stream
    # and event calls per step).
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        cap = torch.cuda.current_stream()
        side.wait_stream(cap)
        with torch.cuda.stream(side):
            ops.stage_from_host(dblk[0], host[0])
            copied[0].record(side)
        for i in range(nb):
            k = i % 2
            if i + 1 < nb:
                with torch.cuda.stream(side):
                    if i >= 1:
                        side.wait_event(used[k ^ 1])       # step i-1 was the last reader of that buffer set
                    ops.stage_from_host(dblk[k ^ 1], host[i + 1])
                    copied[k ^ 1].record(side)
            cap.wait_event(copied[k])
            model.train_step(shape_only, bufs[k][1], bufs[k][2], ids_tm=bufs[k][0])
            used[k].record(cap)
        cap.wait_stream(side)
    reps = max(2, steps // nb)
    for _ in range(2):
        graph.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        graph.replay()
    torch.cuda.synchronize()
    t_pipe = (time.perf_counter() - t0) / (reps * nb)
    # The same pipeline with the copy INSIDE the step: the update launch of step i also brings batch i+1 into the other
    # buffer set (train_step(stage=...), mi_fm_bwd_apply_dense_stage) - one chain of launches, no second stream, no
    # event.  Measured beside the side-stream pipeline and SLOWER, so it is reported, not used: the step takes 49 us,
    # i.e. the update launch (16.5 us alone) lasts ~36 us with the copy workgroups in it - their PCIe reads take longer
    # beside the update than the 26 us of the copy alone, and nothing of them overlaps the step's first launch.
    # (An even number of batches, so that every replay starts at buffer set 0.)
    t_in = None
    if nb % 2 == 0:
        ops.stage_from_host(dblk[0], host[0])
        torch.cuda.synchronize()
        graph2 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph2):
            for i in range(nb):
                k = i % 2
                model.train_step(shape_only, bufs[k][1], bufs[k][2], ids_tm=bufs[k][0],
                                 stage=(dblk[k ^ 1], host[(i + 1) % nb]))
        model.steps -= nb
        for _ in range(2):
            graph2.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            graph2.replay()
        torch.cuda.synchronize()
        t_in = (time.perf_counter() - t0) / (reps * nb)
    for j in range(4):
        ops.stage_from_host(dblk[j % 2], host[j % nb])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for j in range(reps * nb):
        ops.stage_from_host(dblk[j % 2], host[j % nb])
    torch.cuda.synchronize()
    t_copy = (time.perf_counter() - t0) / (reps * nb)
    return {'value': round(B / t_pipe, 1), 'unit': 'samples/s', 'ms_per_step': round(t_pipe * 1e3, 4),
            'bytes_per_batch': nbytes, 'ids': 'int32, field-major [T][B]',
            'copy_only_ms_per_batch': round(t_copy * 1e3, 4), 'copy_only_gbs': round(nbytes / t_copy / 1e9, 1),
            # a perfect overlap would cost max(copy, step) per step; what is lost to the two cross-stream edges a step
            # adds inside the graph and to the copy kernel sharing the chip shows as overlap < 1
            'bound': 'host-to-device copy' if t_copy * 1e3 > resident_ms else 'step',
            'overlap': round(max(t_copy * 1e3, resident_ms) / (t_pipe * 1e3), 3),
            'how': 'pinned host memory, one block per batch copied by a kernel (mi_stage_from_host) on a side stream '
                   'into one of two device buffer sets beside the step on the other; %d steps and their copies '
                   'replayed as one HIP graph' % nb,
            'copy_inside_the_update_launch': None if t_in is None else {
                'value': round(B / t_in, 1), 'ms_per_step': round(t_in * 1e3, 4),
                'how': 'the update launch of a step also copies the next batch\'s block into the other buffer set '
                       '(mi_fm_bwd_apply_dense_stage: copy workgroups behind the dense side\'s); one chain of launches, '
                       'no second stream - measured, not used: slower than the side stream'}}


def host_cores():
    """Threads the CPU baseline may use: the cores this process was given (OMP_NUM_THREADS is set to the
    box's share on a shared machine), never more than the machine has."""
    n = os.cpu_count() or 1
    try:
        share = int(os.environ.get('OMP_NUM_THREADS', '0'))
    except ValueError:
        share = 0
    return max(1, min(n, share) if share > 0 else n)


def cpu_baseline(args, rows, fm, batches):
    """The numpy oracle's FM step (gather formulation; a B x 10 M one-hot matrix is infeasible) on the
    SAME workload - same tables, same batches, same batch size - spread over every host core
    (oracle.fm.train_step_mt: forward over sample shards, backward over tables).  10 warm-up steps,
    then the median of --cpu-steps (default 50) timed steps (fewer only if 30 s run out)."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import fm as ofm
    st = ofm.State.__new__(ofm.State)
    st.toffs = np.concatenate([[0], np.cumsum(rows)]).astype(np.int64)
    st.V = fm.V.contiguous().cpu().numpy()
    st.w1 = fm.w1.contiguous().cpu().numpy()
    st.wd = fm.wd.cpu().numpy().copy()
    st.bias = np.float32(fm.bias.item())
    st.opt, st.t = args.opt, 0
    if args.opt == 'adam_tf':
        st.mV, st.vV = np.zeros_like(st.V), np.zeros_like(st.V)
        st.mw, st.vw = np.zeros_like(st.w1), np.zeros_like(st.w1)
    elif args.opt == 'adagrad_rowwise':
        st.aV, st.aw = np.zeros(st.V.shape[0], np.float32), np.zeros(st.V.shape[0], np.float32)
    st.md, st.vd = np.zeros(len(st.wd) + 1, np.float32), np.zeros(len(st.wd) + 1, np.float32)
    host = [(i.cpu().numpy(), d.cpu().numpy(), l.cpu().numpy()) for i, d, l in batches]
    cores = host_cores()
    times = []
    with ThreadPoolExecutor(cores) as pool:
        for k in range(10):
            ofm.train_step_mt(st, *host[k % len(host)], pool, cores, lr=1e-3, dtype=np.float32)
        t_end = time.perf_counter() + 30.0
        while len(times) < args.cpu_steps and time.perf_counter() < t_end:
            t0 = time.perf_counter()
            ofm.train_step_mt(st, *host[len(times) % len(host)], pool, cores, lr=1e-3, dtype=np.float32)
            times.append(time.perf_counter() - t0)
    med = float(np.median(times))
    return dict(value=round(args.batch / med, 1), unit='samples/s', cores=cores, kind='port',
                sample='median of %d steps after 10 warm-up steps of the same workload (the same %d batches of %d, %d '
                       'tables x dim %d, %s) through the numpy oracle, gather formulation, fp32, %d threads (forward '
                       'over sample shards, backward over tables); machine reports %d cores'
                       % (len(times), len(host), args.batch, len(rows), args.dim, args.opt, cores, os.cpu_count() or 0))


MFMA_PEAK_TFLOPS = 2500.0   # dense bf16 peak of MI355X (MI355X_MICROARCH; the 5 PF headline includes 2:1 sparsity)


def _timed_steps(step, args):
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    return time.perf_counter() - t0


def _pmc_traffic(fname, workload, kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC measurement of this same
    workload (the counters cannot be collected from inside this process); None for any other."""
    path = os.path.join(ROOT, 'profiles', fname)
    if not os.path.exists(path):
        return None
    pmc = json.load(open(path))
    k = pmc.get('kernels', {}).get(kernel)
    return k['hbm_bytes_per_launch'] if k and pmc.get('workload') == workload else None


def _line(args, workload, B, dt, dtype, config):
    return {'metric': 'train_samples_per_sec', 'value': round(B * args.steps / dt, 1), 'unit': 'samples/s', 'n_gpus': 1,
            'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(dt / args.steps * 1e3, 4),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': dtype, 'data': 'synthetic',
            'config': dict(workload=workload, global_batch=B, launch='eager', parallelism='single', **config)}


def bench_dcn(args):
    """DCN-v2 (3 cross layers + MLP 1024-1024-512-1) on the Criteo shape, batch 65536 unless --batch is given."""
    from recommendation_amd import ops
    from recommendation_amd.dcn import DCNv2
    dev = torch.device('cuda', 0)
    B = args.batch if args.batch != 8192 else 65536
    rows = CRITEO_KAGGLE
    m = DCNv2(rows, 16, 13, dev)
    batches = gen_batches(4, B, rows, 13, args.dist, args.idx_bits, dev, seed=1000)
    it = [0]

    def step():
        m.train_step(*batches[it[0] % len(batches)])
        it[0] += 1

    dt = _timed_steps(step, args)
    ops.check_error_word(m.err, "bench ids")
    res = _line(args, 'dcn_v2_criteo_26x16_b%d' % B, B, dt, 'bf16',
                dict(tables=len(rows), total_rows=int(sum(rows)), dim=16, n_dense=13, d_padded=m.dp, cross_layers=3,
                     mlp=[1024, 1024, 512, 1], accumulate='f32', optimizer='adam_tf (dense) + adagrad_rowwise (tables)'))
    res['step_tflops'] = round(m.flops_per_sample() * B * args.steps / dt / 1e12, 1)
    # every contraction launch of the step timed on its own, by kernel class: 'row' = gemm_bf16_kernel with both
    # operands stored reduction-contiguous (forward layers and data gradients), 'wgrad' = both reduction-major
    # (weight gradients).  The roofline line is the class that takes more of the step, ALL its launches summed.
    # Beside each: the vendor library's product of the SAME two operands (torch.matmul -> hipBLASLt / rocBLAS), which
    # stores one bf16 (row class) or fp32 (weight gradients) result and nothing else - no bias, activation, cross
    # term, mask or second output, which this library's launches fuse into the same pass.
    per = []
    for cls, name, fn, flops, (opa, opb) in m.contraction_calls():
        ms = time_stage(fn, max(2, args.stage_reps // 5))
        if cls == 'row':
            out = torch.empty((opa.shape[0], opb.shape[0]), dtype=torch.bfloat16, device=dev)
            vfn = lambda opa=opa, opb=opb, out=out: torch.matmul(opa, opb.t(), out=out)           # noqa: E731
        else:
            out = torch.empty((opa.shape[1], opb.shape[1]), dtype=torch.bfloat16, device=dev)
            vfn = lambda opa=opa, opb=opb, out=out: torch.matmul(opa.t(), opb, out=out)           # noqa: E731
        vms = time_stage(vfn, max(2, args.stage_reps // 5))
        per.append(dict(kernel_class=cls, name=name, ms=round(ms, 5), flops=flops, tflops=round(flops / ms / 1e9, 1),
                        vendor_ms=round(vms, 5), vendor_tflops=round(flops / vms / 1e9, 1)))
    res['contractions'] = per
    tot = {c: (sum(p['flops'] for p in per if p['kernel_class'] == c), sum(p['ms'] for p in per if p['kernel_class'] == c))
           for c in ('row', 'wgrad')}
    vtot = {c: sum(p['vendor_ms'] for p in per if p['kernel_class'] == c) for c in ('row', 'wgrad')}
    res['contraction_classes'] = {c: dict(launches=sum(1 for p in per if p['kernel_class'] == c), ms=round(t, 4),
                                          tflops=round(f / t / 1e9, 1), vendor_ms=round(vtot[c], 4),
                                          vendor_tflops=round(f / vtot[c] / 1e9, 1)) for c, (f, t) in tot.items()}
    res['vendor_comparator'] = 'torch.matmul (bf16, %s) on the same operands, product only' % torch.__version__
    dom = max(tot, key=lambda c: tot[c][1])
    flops, ms = tot[dom]
    n_launch = res['contraction_classes'][dom]['launches']
    res['roofline'] = {'kernel': 'gemm_bf16_kernel, %s operands: the %d launches of one step (%s)' % (
                           'row-stored' if dom == 'row' else 'reduction-major', n_launch,
                           'forward layers + data gradients' if dom == 'row' else 'weight gradients'),
                       'bound': 'mfma', 'achieved': round(flops / ms / 1e9, 1), 'peak': MFMA_PEAK_TFLOPS,
                       'unit': 'TFLOP/s', 'frac': round(flops / ms / 1e9 / MFMA_PEAK_TFLOPS, 4),
                       'traffic': None, 'traffic_source': None,
                       'alg_flops_per_launch': flops / n_launch, 'ms_per_launch': round(ms / n_launch, 5),
                       'share_of_step': round(ms / (dt / args.steps * 1e3), 3)}
    if not args.no_cpu_baseline:
        from oracle import dcn as odcn
        Bc = 4096
        st = odcn.State([8] * len(rows), 16, 13, seed=0)          # small tables, then the model's own weights
        st.toffs = np.concatenate([[0], np.cumsum(rows)]).astype(np.int64)
        st.V = m.V.cpu().numpy()
        st.acc = np.zeros(st.V.shape[0], np.float32)
        ids, dense, labels = (t[:Bc].cpu().numpy() for t in batches[0])
        odcn.train_step(st, ids, dense, labels, dtype=np.float32)
        t0, n = time.perf_counter(), 0
        while n < args.cpu_steps and time.perf_counter() - t0 < 20.0:
            odcn.train_step(st, ids, dense, labels, dtype=np.float32)
            n += 1
        cdt = time.perf_counter() - t0
        res['cpu_baseline'] = dict(value=round(n * Bc / cdt, 1), unit='samples/s',
                                   cores=int(os.environ.get('OMP_NUM_THREADS', os.cpu_count() or 1)), kind='port',
                                   sample='%d steps of batch %d (same model, fp32) through the numpy oracle; its '
                                          'matrix products run on the BLAS threads numpy was given' % (n, Bc))
    return res


def _embag_cpu_baseline(args, T, D, B, L, opt, ids_full, grad):
    """The embedding-bag step of the numpy oracle (oracle.embedding.step_mt: gather + pooling, stable argsort,
    segment sums, optimizer on the distinct rows - the gather formulation; a one-hot matrix of these tables is
    infeasible) on every host core, at the SAME batch, dim, bag length and table count.  The tables are cut to
    400 k rows each (the host cannot build 64 GB of rows in the seconds a bench run has; a row still misses the
    CPU's caches, but the duplicates per step are those of the cut tables)."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import embedding as oe
    rc = 400000
    cores = host_cores()
    rng = np.random.default_rng(0)
    toffs_c = np.arange(T + 1, dtype=np.int64) * rc
    Wc = rng.random((T * rc, D), dtype=np.float32)
    Wc -= np.float32(0.5)
    Wc *= np.float32(0.1)
    stc = np.zeros(T * rc, np.float32) if opt == 'adagrad_rowwise' else None
    mc, vc = (np.zeros_like(Wc), np.zeros_like(Wc)) if opt == 'adam_tf' else (None, None)
    idc = (ids_full.reshape(B, T, L) % rc).astype(np.int64)
    out = np.empty((B, T * D), np.float32)
    times = []
    with ThreadPoolExecutor(cores) as pool:
        def cpu_step():
            oe.step_mt(Wc, stc, mc, vc, toffs_c, T, idc, grad, opt, pool, cores, lr=0.01, eps=1e-8, lr_t=1e-3, out=out)
        for _ in range(3):
            cpu_step()
        t_end = time.perf_counter() + 20.0
        while len(times) < args.cpu_steps and (len(times) < 3 or time.perf_counter() < t_end):
            c0 = time.perf_counter()
            cpu_step()
            times.append(time.perf_counter() - c0)
    return dict(value=round(B / float(np.median(times)), 1), unit='samples/s', cores=cores, kind='port',
                sample='median of %d steps after 3 warm-up steps of the same batch (%d x %d bags of %d ids, dim %d, %s) '
                       'through the numpy oracle (gather formulation, vectorised per table, %d threads over the tables), '
                       'tables cut to %d rows each (%.1f GB)' % (len(times), B, T, L, D, opt, cores, rc, T * rc * D * 4 / 1e9))


def bench_embag(args):
    """Embedding gather + sum pooling forward and the sort-based backward with the fused
    optimizer (TensorFlow-form Adam unless --opt says otherwise), on one GPU's share of the sharded-table
    configuration: 26 tables, 125 M rows x dim 128 fp32 (64 GB, 192 GB with Adam's two moments), batch
    65536.  The upstream gradient is a fixed synthetic buffer."""
    from recommendation_amd import ops
    dev = torch.device('cuda', 0)
    T, L = 26, args.bag_len
    D = args.dim if args.dim != 16 else 128
    B = args.batch if args.batch != 8192 else 65536
    rows = [args.embag_rows // T] * T
    total = sum(rows)
    toffs = ops.table_offsets(rows, dev)
    g = torch.Generator(device=dev)
    g.manual_seed(1000)
    opt = args.opt
    # the row-wise state is allocated BEFORE the 64 GB table: allocated behind it, the same update launch took 0.73-0.75 ms
    # instead of 0.66-0.70 ms in 7 of 16 fresh processes (none of 16 this way round; scripts/probe_adagrad_modes.py)
    state = torch.zeros((total,), dtype=torch.float32, device=dev) if opt == 'adagrad_rowwise' else None
    W = torch.empty((total, D), dtype=torch.float32, device=dev)
    chunk = 1 << 22
    for lo in range(0, total, chunk):
        hi = min(lo + chunk, total)
        W[lo:hi] = torch.randn((hi - lo, D), generator=g, dtype=torch.float32, device=dev).mul_(0.05)
    am, av = (torch.zeros_like(W), torch.zeros_like(W)) if opt == 'adam_tf' else (None, None)
    card = torch.tensor(rows, dtype=torch.float64, device=dev).view(1, T, 1)
    idt = torch.int64 if args.idx_bits == 64 else torch.int32
    batches = []
    for _ in range(4):
        u = torch.rand((B, T, L), generator=g, dtype=torch.float64, device=dev)
        if args.dist == 'zipf':
            a = 1.0 - 1.05
            ids = torch.floor(((u * ((card + 1.0) ** a - 1.0)) + 1.0) ** (1.0 / a)) - 1.0
        else:
            ids = torch.floor(u * card)
        batches.append(torch.minimum(ids.clamp_(min=0), card - 1).to(idt).contiguous())
    grad = torch.randn((B, T * D), generator=g, dtype=torch.float32, device=dev).mul_(1e-3)
    out = torch.empty((B, T * D), dtype=torch.float32, device=dev)
    err = ops.new_error_word(dev)
    plan = ops.SortedPlan(toffs, T, total, batches[0], bag_len=L, runs=False)
    it = [0]

    def fwd(ids):
        ops.embag_fwd(W, toffs, T, ids, bag_len=L, check_flags=ops.MI_CHECK_RANGE, err=err, out=out)

    def upd():
        if opt == 'adagrad_rowwise':
            ops.bwd_apply_adagrad_rowwise(plan, grad, W, state, 0.01, 1e-8)
        elif opt == 'sgd':
            ops.bwd_apply_sgd(plan, grad, W, 0.01)
        else:
            ops.bwd_apply_adam_tf(plan, grad, W, am, av, 1e-3, 0.9, 0.999, 1e-8)

    def step():
        ids = batches[it[0] % len(batches)]
        fwd(ids)
        plan.prepare(ids)
        upd()
        it[0] += 1

    dt = _timed_steps(step, args)
    ops.check_error_word(err, "bench ids")
    n = B * T * L
    ib = args.idx_bits // 8
    res = _line(args, 'embag_26x%d_rows%dM_b%d_l%d' % (D, total // 1000000, B, L), B, dt, 'f32',
                dict(tables=T, total_rows=total, dim=D, bag_len=L, pooling='sum', id_bits=args.idx_bits, ids=args.dist,
                     table_bytes=total * D * 4, optimizer='%s (fused into the sparse backward)' % opt))
    ids = batches[0]
    plan.prepare(ids)
    U = int(torch.unique(plan.sorted_rows[:n]).numel())
    fwd_bytes = n * (D * 4 + ib) + B * T * D * 4
    state_floats = {'sgd': 0, 'adagrad_rowwise': 1, 'adam_tf': 2 * D}[opt]
    upd_bytes = n * 8 + (n if L == 1 else B * T) * D * 4 + U * (D + state_floats) * 4 * 2
    stages = []
    for name, fn, nb in (('embag_fwd (gather + sum pooling)', lambda: fwd(ids), fwd_bytes),
                         ('sort_by_row', lambda: plan.prepare(ids), n * (ib + 8)),
                         ('sparse backward + fused %s' % opt, upd, upd_bytes)):
        ms = time_stage(fn, args.stage_reps)
        stages.append(dict(name=name, ms=round(ms, 5), alg_bytes=int(nb), gbs=round(nb / ms / 1e6, 1)))
    res['stages'] = stages
    res['unique_rows_per_step'] = U
    emb = [stages[0], stages[2]]
    res['embedding_hbm_gbs'] = round(sum(s['alg_bytes'] for s in emb) / sum(s['ms'] for s in emb) / 1e6, 1)
    dom = max(emb, key=lambda s: s['ms'])
    kern = 'embag_fwd_kernel' if dom is stages[0] else 'seg_sorted_kernel'
    res['roofline'] = {'kernel': dom['name'], 'bound': 'hbm', 'achieved': dom['gbs'], 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                       'frac': round(dom['gbs'] / HBM_PEAK_GBS, 4),
                       'traffic': (_pmc_traffic('r03_embag_b65536_%s_pmc.json' % opt, res['config']['workload'], kern) or
                                   _pmc_traffic('r02_embag_b65536_%s_pmc.json' % opt, res['config']['workload'], kern) or
                                   _pmc_traffic('r01_embag_b65536_%s_pmc.json' % opt, res['config']['workload'], kern)),
                       'traffic_source': 'quoted profiles/r0x_embag_b65536_%s_pmc.json (null when absent or another workload)' % opt,
                       'alg_bytes_per_launch': dom['alg_bytes'], 'ms_per_launch': dom['ms']}
    if not args.no_cpu_baseline:
        res['cpu_baseline'] = _embag_cpu_baseline(args, T, D, B, L, opt, batches[0].cpu().numpy(), grad.cpu().numpy())
    return res


def bench_sharded_embag(args):
    """Row-wise sharded embedding bags, the 1 G-row x dim-128 configuration: every rank holds
    --embag-rows rows (default 125 M = 64 GB fp32: an eighth of 1 G rows, so 8 ranks hold the whole
    configuration) of 8 tables dealt round robin over the ranks, batch 8192 per rank (65536 at 8
    ranks), one id per bag, row-wise Adagrad.  A step = all-gather of ids, owner partial pools,
    all-to-all of pooled rows, combine; a fixed synthetic upstream gradient; all-to-all of gradient
    rows, sort by row, fused update.  Weak scaling: rows and batch per rank are fixed."""
    from recommendation_amd import ops
    from recommendation_amd.sharded_bag import ShardedEmbeddingBag
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    sys.stdout.flush()
    saved_stdout = os.dup(1)          # RCCL prints a banner on stdout; the JSON line must be alone there
    os.dup2(2, 1)
    if world == 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29578')
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('WORLD_SIZE', '1')
    torch.distributed.init_process_group('nccl', device_id=dev)
    T, L = 8, args.bag_len
    D = args.dim if args.dim != 16 else 128
    B = args.batch
    opt = args.opt if args.opt in ('sgd', 'adagrad_rowwise') else 'adagrad_rowwise'
    rows = [args.embag_rows * world // T] * T
    comm = None
    if args.comm == 'rccl':     # the library's own RCCL entry points, enqueued on the compute stream itself
        from recommendation_amd.sharded import RcclComm
        comm = RcclComm()
    m = ShardedEmbeddingBag(rows, D, L, dev, opt=opt, lr=0.01, seed=0, comm=comm, id_exchange=args.id_exchange)
    g = torch.Generator(device=dev)
    g.manual_seed(1000 + rank)
    card = torch.tensor(rows, dtype=torch.float64, device=dev).view(1, T, 1)
    idt = torch.int64 if args.idx_bits == 64 else torch.int32
    batches = []
    for _ in range(8):
        u = torch.rand((B, T, L), generator=g, dtype=torch.float64, device=dev)
        if args.dist == 'zipf':
            a = 1.0 - 1.05
            ids = torch.floor(((u * ((card + 1.0) ** a - 1.0)) + 1.0) ** (1.0 / a)) - 1.0
        else:
            ids = torch.floor(u * card)
        batches.append(torch.minimum(ids.clamp_(min=0), card - 1).to(idt).contiguous())
    grad = torch.randn((B, T * D), generator=g, dtype=torch.float32, device=dev).mul_(1e-3)
    it = [0]

    def step():
        m.forward(batches[it[0] % len(batches)])
        m.backward(grad)
        it[0] += 1

    def barrier():
        torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    t_issue = time.perf_counter() - t0          # the host has enqueued every step; the device may still be running
    barrier()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    dt = float(t.item())
    m.be.check()
    res = None
    if rank == 0:
        ib = args.idx_bits // 8
        n_bags = B * T
        res = {'metric': 'train_samples_per_sec', 'value': round(B * world * args.steps / dt, 1), 'unit': 'samples/s',
               'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(dt / args.steps * 1e3, 4),
               'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
               'config': {'workload': 'sharded_embag_%dx%d_rows%dM_per_rank_b%d_l%d' % (T, D, args.embag_rows // 1000000, B, L),
                          'global_batch': B * world, 'tables': T, 'total_rows': int(sum(rows)), 'rows_per_rank': m.be.local_rows,
                          'dim': D, 'bag_len': L, 'pooling': 'sum', 'id_bits': args.idx_bits, 'ids': args.dist,
                          'optimizer': '%s (fused into the sparse backward)' % opt, 'launch': 'eager',
                          'placement': 'global row r on rank r %% %d as local row r / %d' % (world, world),
                          'capacity_rows_per_rank_pair': m.cap,
                          'collectives': ('torch.distributed (nccl backend = RCCL)' if args.comm != 'rccl' else
                                          "the library's RCCL entry points on the compute stream"),
                          'parallelism': 'rowwise-sharded tables x%d, pooled rows exchanged' % world}}
        res['host_issue_ms_per_step'] = round(t_issue / args.steps * 1e3, 4)   # == ms_per_step when the host is the bound
        res['config']['id_exchange'] = args.id_exchange
        res['collectives'] = {('all_gather_ids_bytes_received' if not m.bucketed else 'all_to_all_bucketed_ids_bytes_per_rank'):
                              (world * n_bags * L * ib if not m.bucketed else world * (m.cap_ids * 4 + m.cap * 8)),
                              'all_to_all_pooled_rows_bytes_per_rank': world * m.cap * D * 4,
                              'all_to_all_gradient_rows_bytes_per_rank': world * m.cap * D * 4,
                              'rows_that_carry_data_per_exchange': n_bags if L == 1 else None}
    # per-launch stages on rank 0's buffers of the last step (no collectives inside)
    ids = batches[(it[0] - 1) % len(batches)]
    plan = m.be._plan
    opt_code = ops.MI_OPT_SGD if opt == 'sgd' else ops.MI_OPT_ADAGRAD_ROWWISE
    if m.bucketed:
        n_own = world * m.cap
        n_ent = int((m.recv_ids >= 0).sum().item())
        U = int(torch.unique(plan.sorted_rows[:plan.n_entries]).numel()) - (1 if n_ent < plan.n_entries else 0)

        def upd0():
            g = m.grad_in.view(-1, D)
            if opt == 'sgd':
                ops.bwd_apply_sgd(plan, g, m.be.W, 0.0)
            else:
                ops.bwd_apply_adagrad_rowwise(plan, g, m.be.W, m.be.state, 0.0, 1e-8)
        stage_list = (
            ('requester slots + ids bucketed by owner (4 launches, 1 fill)', lambda: m.phase_bucket(ids.reshape(B, T, L)),
             2 * n_bags * L * ib + world * n_bags * 5 + world * (m.cap_ids * 4 + m.cap * 8)),
            ('owner gather + partial pooling (mi_embag_fwd over the received CSR bags)',
             lambda: m.be.gather_csr(m.recv_ids, m.recv_off, m.rows_out),
             world * (m.cap_ids * 4 + m.cap * 8) + n_ent * D * 4 + n_own * D * 4),
            ('requester combine', lambda: m.be.combine(m.rows_in, m.req_row_of_bag, m.n_bags, m.pooled),
             world * n_bags * 4 + n_bags * D * 4 * 2),
            ('requester gradient pack', lambda: m.be.pack(grad, m.req_bag_of_row, m.req_cnt, m.cap, m.grad_out),
             n_bags * (4 + D * 4 * 2)),
            ('owner sort by row (CSR)', lambda: plan.prepare(m.recv_ids), plan.n_entries * 12),
            ('owner sparse backward + fused %s' % opt, upd0,
             n_ent * (8 + 4 + D * 4) + U * (D + (1 if opt != 'sgd' else 0)) * 4 * 2))
        hbm_idx = (1, 5)
    else:
        n_own = int(m.own_cnt.sum().item())
        n_ent = int((m.local_ids >= 0).sum().item())
        U = int(torch.unique(plan.sorted_rows[:plan.n_entries]).numel()) - (1 if n_ent < plan.n_entries else 0)
        stage_list = (
            ('owner slots (2 launches: ownership flags, local ids, block rows)',
             lambda: m.be.owner_slots(m.ids_all, m.n_bags, m.cap, m.local_ids, m.own_row_of_bag, m.own_bag_of_row, m.own_cnt, m.ws),
             world * n_bags * L * (ib + 4) + world * n_bags * 5),
            ('owner gather + partial pooling', lambda: m.be.gather(m.local_ids, m.n_bags, m.cap, m.own_bag_of_row, m.own_cnt, m.rows_out),
             n_own * (4 + L * 4) + n_ent * D * 4 + n_own * D * 4),
            ('requester slots (2 launches)',
             lambda: m.be.requester_slots(ids, m.n_bags, m.cap, m.req_row_of_bag, m.req_bag_of_row, m.req_cnt, m.ws),
             n_bags * L * ib + world * n_bags * 5),
            ('requester combine', lambda: m.be.combine(m.rows_in, m.req_row_of_bag, m.n_bags, m.pooled),
             world * n_bags * 4 + n_bags * D * 4 * 2),
            ('requester gradient pack', lambda: m.be.pack(grad, m.req_bag_of_row, m.req_cnt, m.cap, m.grad_out),
             n_bags * (4 + D * 4 * 2)),
            ('owner sort by row', lambda: plan.prepare(m.local_ids), plan.n_entries * 12),
            ('owner sparse backward + fused %s' % opt,
             lambda: ops.bwd_apply_rows(plan, m.grad_in, m.own_row_of_bag, m.be.W, opt_code, 0.0, state=m.be.state),
             n_ent * (8 + 4 + D * 4) + U * (D + (1 if opt != 'sgd' else 0)) * 4 * 2))
        hbm_idx = (1, 6)
    stages = []
    for name, fn, nb in stage_list:
        ms = time_stage(fn, args.stage_reps)
        stages.append(dict(name=name, ms=round(ms, 5), alg_bytes=int(nb), gbs=round(nb / ms / 1e6, 1)))
    if rank == 0:
        res['stages'] = stages
        res['entries_at_this_owner'] = n_ent
        res['unique_rows_per_step'] = U
        hbm = [stages[i] for i in hbm_idx]
        dom = max(hbm, key=lambda s: s['ms'])
        res['embedding_hbm_gbs'] = round(sum(s['alg_bytes'] for s in hbm) / sum(s['ms'] for s in hbm) / 1e6, 1)
        res['roofline'] = {'kernel': dom['name'], 'bound': 'hbm', 'achieved': dom['gbs'], 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                           'frac': round(dom['gbs'] / HBM_PEAK_GBS, 4), 'traffic': None, 'traffic_source': None,
                           'alg_bytes_per_launch': dom['alg_bytes'], 'ms_per_launch': dom['ms']}
        if not args.no_cpu_baseline:
            res['cpu_baseline'] = _embag_cpu_baseline(args, T, D, B, L, opt, batches[0].cpu().numpy(), grad.cpu().numpy())
    torch.distributed.barrier()
    sys.stdout.flush()
    os.dup2(saved_stdout, 1)
    os.close(saved_stdout)
    if rank == 0:
        print(json.dumps(res), flush=True)
    os.dup2(2, 1)
    torch.distributed.destroy_process_group()


def bench_din(args):
    """DIN local activation unit (8 M items x dim 64 = 2 GB, far past the 256 MiB last-level cache, so that the
    gather's rate is an HBM rate; history 100, lengths U[1,100], MLP 256-80-40-1), batch 8192."""
    from recommendation_amd import ops
    from recommendation_amd.din import DIN
    dev = torch.device('cuda', 0)
    B, L, rows = args.batch, 100, args.din_items
    m = DIN(rows, dev, max_len=L, factored=False if args.din_full_rows else None)
    g = torch.Generator(device=dev)
    g.manual_seed(1000)
    batches = []
    for _ in range(4):
        hist = torch.randint(0, rows, (B, L), generator=g, device=dev)
        if args.idx_bits == 32:
            hist = hist.to(torch.int32)
        qid = torch.randint(0, rows, (B,), generator=g, device=dev).to(hist.dtype)
        lens = torch.randint(1, L + 1, (B,), generator=g, device=dev).to(torch.int32)
        labels = (torch.rand((B,), generator=g, device=dev) < 0.3).to(torch.float32)
        batches.append((hist, qid, lens, labels))
    it = [0]

    def step():
        m.train_step(*batches[it[0] % len(batches)])
        it[0] += 1

    dt = _timed_steps(step, args)
    ops.check_error_word(m.err, "bench ids")
    res = _line(args, 'din_lau_%dMx64_L%d_b%d' % (rows // 1000000, L, B), B, dt, 'bf16',
                dict(items=rows, dim=64, history=L, lengths='U[1,%d]' % L, lau_mlp=[256, 80, 40, 1],
                     hidden_padded=[m.h1p, m.h2p], accumulate='f32',
                     optimizer='adam_tf (dense) + adagrad_rowwise (table)'))
    res['step_tflops'] = round(m.flops_per_sample() * B * args.steps / dt / 1e12, 1)
    ib = args.idx_bits // 8
    res['config']['first_layer'] = 'factored: rows [k | q*k], q term per sample' if m.factored else 'full rows [q | k | q-k | q*k]'
    D, H, H2 = 64, m.h1p, m.h2p
    if m.factored and ops.din_lau_fwd_fused_ok(D, H, H2, L):
        # The unit runs as one launch per direction; the dominant HBM kernel is the backward's data path.  It reads the
        # k block of A, H1, H2 and the weights a, writes dZ2, dZ1 and the rows' gradient: B*L rows whatever the lengths.
        res['config']['unit'] = 'one launch per direction, a workgroup per sample (layers out of LDS)'
        W2T, W1fT = m.W2.hT, m.W1fT
        dfeat = torch.randn((B, 2 * D), device=dev)
        P = ops.din_lau_bwd_part_floats()
        S_, part, sums = (torch.empty((B, H), device=dev), torch.empty((B, P), device=dev), torch.empty((256, P), device=dev))
        dZ2, dZ1, dE = torch.empty_like(m.H2), torch.empty_like(m.H1), torch.empty((B * (L + 1), D), device=dev)
        m.train_step(*batches[0])
        ms = time_rotating(lambda j: ops.din_lau_bwd_fused(m.A, m.H1, m.H2, m.a, m.ids_all, m.feat, dfeat, m.w3.w, W2T, W1fT,
                                                           dZ2, dZ1, dE, S_, part, sums), 1, args.stage_reps)
        nbytes = B * L * (D * 2 + H * 2 + H2 * 2 + 4 + 4) + B * L * (H2 * 2 + H * 2) + B * (L + 1) * D * 4 + B * (H + P) * 4
        res['roofline'] = {'kernel': 'din_lau_bwd_fused_kernel (+ rows_sum_f32_kernel, 3 us): reads k, H1, H2, a; writes dZ2, dZ1, dE',
                           'bound': 'hbm',
                           'achieved': round(nbytes / ms / 1e6, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                           'frac': round(nbytes / ms / 1e6 / HBM_PEAK_GBS, 4),
                           'traffic': _pmc_traffic('r03_din_b8192_pmc.json', res['config']['workload'], 'din_lau_bwd_fused_kernel'),
                           'traffic_source': 'quoted profiles/r03_din_b8192_pmc.json (null when absent or another workload)',
                           'alg_bytes_per_launch': nbytes, 'ms_per_launch': round(ms, 5)}
    else:
        # dominant HBM kernel: the gather that builds the unit's input rows
        ms = time_rotating(lambda j: ops.din_lau_build(m.E, batches[j][0], batches[j][1], batches[j][2], m.A, m.ids_all,
                                                       q_out=m.feat[:, 64:], err=m.err), len(batches), args.stage_reps)
        live = sum(int(b[2].sum().item()) for b in batches) / float(len(batches))
        nbytes = live * 64 * 4 + B * L * ib + B * L * m.A.shape[1] * 2 + B * (L + 1) * 4   # live rows + ids + A rows + index list
        res['roofline'] = {'kernel': 'din_build_kernel (gather + %s rows)' % ('[k|q*k]' if m.factored else '[q|k|q-k|q*k]'),
                           'bound': 'hbm',
                           'achieved': round(nbytes / ms / 1e6, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                           'frac': round(nbytes / ms / 1e6 / HBM_PEAK_GBS, 4),
                           'traffic': None,
                           'traffic_source': None,
                           'alg_bytes_per_launch': nbytes, 'ms_per_launch': round(ms, 5)}
    if not args.no_cpu_baseline:
        from oracle import din as odin
        Bc = 1024
        st = odin.State(16, seed=0)
        st.rows = rows
        st.E = m.E.cpu().numpy()
        st.acc = np.zeros(rows, np.float32)
        hb, qb, lb, yb = (t[:Bc].cpu().numpy() for t in batches[0])
        odin.train_step(st, hb, qb, lb, yb, dtype=np.float32)
        t0, n = time.perf_counter(), 0
        while n < args.cpu_steps and time.perf_counter() - t0 < 20.0:
            odin.train_step(st, hb, qb, lb, yb, dtype=np.float32)
            n += 1
        cdt = time.perf_counter() - t0
        res['cpu_baseline'] = dict(value=round(n * Bc / cdt, 1), unit='samples/s',
                                   cores=int(os.environ.get('OMP_NUM_THREADS', os.cpu_count() or 1)), kind='port',
                                   sample='%d steps of batch %d (same model, fp32) through the numpy oracle' % (n, Bc))
    return res


def bench_colour(args):
    """The reference's own model and train step (recommendation/local_train_utils.py:30-66:
    RecommendationModel(128, 6, 100, 20), batch 300, tf.train.AdamOptimizer(1e-4)) on the GPU, beside the
    numpy oracle's restatement of the reference's formulation on the host: dense one-hot / n-hot
    matrices materialised and multiplied into the first kernels, dense TensorFlow-form Adam."""
    from oracle import colour_model as cm
    from recommendation_amd.r_model import RecommendationModel
    from recommendation_amd.trainer import AdamOptimizer, GraphedTrainStep, train_step
    dev = torch.device('cuda', 0)
    B = args.batch if args.batch != 8192 else 300
    params = cm.init_params(seed=1, dtype=np.float32)
    m = RecommendationModel(128, 6, 100, 20, device=dev)
    m.load_variables(params)
    m.check_every = 64                 # the id error word is read every 64 steps and once at the end
    opt = AdamOptimizer(learning_rate=1e-4)
    host = [cm.synthetic_batch(B, seed=20211207 + j) for j in range(8)]
    batches = [tuple(torch.from_numpy(b[k]).to(dev) for k in cm._KEYS) for b in host]
    it = [0]
    graphed = None if args.no_graph else GraphedTrainStep(m, opt, batches[0], check_every=64)

    def step():
        if graphed is not None:
            graphed(batches[it[0] % len(batches)])      # copies the batch into the graph's inputs, replays
        else:
            train_step(m, opt, batches[it[0] % len(batches)])
        it[0] += 1

    dt = _timed_steps(step, args)
    m.check_errors()
    res = _line(args, 'colour_model_train_b%d' % B, B, dt, 'f32',
                dict(model='RecommendationModel(128, 6, 100, 20)', parameters=5910, optimizer='adam_tf (dense, lr 1e-4)',
                     includes='encoders, 11 linear layers, softmax / top-6 / accuracy / loss, backward, Adam'))
    res['config']['launch'] = 'eager' if graphed is None else 'hip graph replay (7 input copies + 1 replay per step)'
    # plumbing, not a bandwidth workload: a sample is 16 ids and 16 gathered rows (124 floats); the step is
    # bound by the launches of a 5,910-parameter model
    nbytes = B * (16 * 4 + 124 * 4 + 2 * 128 * 4) + 3 * 2 * 5910 * 4
    ms = dt / args.steps * 1e3
    res['roofline'] = {'kernel': 'whole step (about 70 launches of a 5,910-parameter model: launch-bound)', 'bound': 'hbm',
                       'achieved': round(nbytes / ms / 1e6, 3), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                       'frac': round(nbytes / ms / 1e6 / HBM_PEAK_GBS, 6), 'traffic': None, 'traffic_source': None,
                       'alg_bytes_per_launch': nbytes, 'ms_per_launch': round(ms, 5)}
    if not args.no_cpu_baseline:
        cores = host_cores()
        torch.set_num_threads(cores)
        p32 = {n: p.copy() for n, p in params.items()}
        copt = cm.AdamTF(p32, lr=1e-4)
        for k in range(10):
            cm.train_step(p32, copt, host[k % len(host)])
        times = []
        t_end = time.perf_counter() + 30.0
        while len(times) < args.cpu_steps and time.perf_counter() < t_end:
            c0 = time.perf_counter()
            cm.train_step(p32, copt, host[len(times) % len(host)])
            times.append(time.perf_counter() - c0)
        res['cpu_baseline'] = dict(value=round(B / float(np.median(times)), 1), unit='samples/s', cores=cores, kind='port',
                                   sample='median of %d steps after 10 warm-up steps, same batches of %d, through the numpy '
                                          'oracle in the reference\'s formulation (one-hot / n-hot matrices materialised, '
                                          'matmul, dense Adam; fp32; numpy\'s BLAS on %d threads)' % (len(times), B, cores))
    return res


def bench_infer(args):
    """ModelInference.inference on record strings (tokenizer + colour-model forward + top-6 + string
    formatting): the serving-side consumer of the forward path, batch 4096 unless --batch is given."""
    from oracle import colour_model as cm
    from recommendation_amd.inference import ModelInference
    B = args.batch if args.batch != 8192 else 4096
    batch = cm.synthetic_batch(B, seed=1, with_label=False)
    records = [cm.format_record(batch, i, with_label=False) for i in range(B)]
    mi = ModelInference(device=torch.device('cuda', 0))
    dt = _timed_steps(lambda: mi.inference(records), args)
    res = _line(args, 'colour_model_inference_b%d' % B, B, dt, 'f32',
                dict(model='RecommendationModel(128, 6, 100, 20)', includes='record tokenizer, forward, top-6, result strings'))
    res['metric'] = 'inference_records_per_sec'
    res['unit'] = 'records/s'
    return res


def spawn_ranks(n, script=None, argv=None):
    """`python bench.py --gpus N` without a launcher: start N child processes of this same command,
    one per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment), from a parent that
    makes no GPU call; relay rank 0's JSON line; non-zero exit status if any rank fails.  The
    reference starts its own workers too (recommendation/train_job.py:41-63, 2 workers + 1 ps)."""
    import socket
    import subprocess
    have = torch.cuda.device_count()          # counting devices does not initialise the GPU
    if have < n:
        print("bench.py: --gpus %d but this machine has %d GPU(s)" % (n, have), file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, script or os.path.abspath(__file__)] +
                                      (sys.argv[1:] if argv is None else list(argv)), env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out = b''
    failed = None
    import selectors
    sel = selectors.DefaultSelector()
    sel.register(procs[0].stdout, selectors.EVENT_READ)
    open_out = True
    while True:
        if open_out:
            for key, _ in sel.select(timeout=0.2):
                chunk = os.read(key.fileobj.fileno(), 65536)
                if chunk:
                    out += chunk
                else:
                    sel.unregister(key.fileobj)
                    open_out = False
        else:
            time.sleep(0.2)
        codes = [p.poll() for p in procs]
        bad = [c for c in codes if c not in (None, 0)]
        if bad and failed is None:
            failed = bad[0]
            for p in procs:                       # the others would wait in a collective for ever
                if p.poll() is None:
                    p.kill()
        if all(c is not None for c in codes) and not open_out:
            break
    lines = [l for l in out.decode(errors='replace').splitlines() if l.startswith('{')]
    if lines:
        print(lines[-1], flush=True)
    if failed is not None:
        return failed if failed > 0 else 1
    return 0 if lines else 1


def main():
    args = parse_args()
    if args.workload == 'infer':
        torch.cuda.set_device(0)
        print(json.dumps(bench_infer(args)))
        return
    if args.workload == 'sharded_embag':
        if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
            raise SystemExit(spawn_ranks(args.gpus))
        if int(os.environ.get('WORLD_SIZE', '1')) != args.gpus:
            raise SystemExit("WORLD_SIZE=%s does not match --gpus %d" % (os.environ.get('WORLD_SIZE'), args.gpus))
        bench_sharded_embag(args)
        return
    if args.workload != 'fm':
        if args.gpus != 1 or int(os.environ.get('WORLD_SIZE', '1')) != 1:
            raise SystemExit("--workload %s is a single-GPU configuration" % args.workload)
        torch.cuda.set_device(0)
        fn = {'dcn': bench_dcn, 'din': bench_din, 'embag': bench_embag, 'colour': bench_colour}[args.workload]
        print(json.dumps(fn(args)))
        return
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # no launcher around us: start the ranks ourselves (this process has not touched the GPU yet)
        raise SystemExit(spawn_ranks(args.gpus))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE=%d does not match --gpus %d" % (world, args.gpus))
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    sharded = world > 1 or args.force_sharded
    saved_stdout = None
    if sharded:
        # RCCL prints a version banner on stdout when its communicator is created; stdout must carry
        # the one JSON line only, so everything before it goes to stderr
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        if world == 1:
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            os.environ.setdefault('MASTER_PORT', '29577')
            os.environ.setdefault('RANK', '0')
            os.environ.setdefault('WORLD_SIZE', '1')
        torch.distributed.init_process_group('nccl', device_id=dev)

    rows = CRITEO_KAGGLE
    B = args.batch
    if not sharded:
        from recommendation_amd.fm import FactorizationMachine
        model = FactorizationMachine(rows, args.dim, 13, device=dev, opt=args.opt, lr=1e-3, seed=0)
    else:
        from recommendation_amd.sharded import RcclComm, ShardedFactorizationMachine
        # --comm rccl: the library's own RCCL entry points, enqueued on the compute stream itself (no hop to a
        # collective stream and back per exchange); one communicator on one stream, so no side-stream id exchange
        comm = RcclComm() if args.comm == 'rccl' else None
        model = ShardedFactorizationMachine(rows, args.dim, 13, device=dev, opt=args.opt, lr=1e-3, seed=0, comm=comm)
    batches = gen_batches(8, B, rows, 13, args.dist, args.idx_bits, dev, seed=1000 + args.seed + rank)

    def barrier():
        if sharded:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    use_graph = (not sharded) and not args.no_graph
    kw = {'graph': True} if use_graph else {}
    # one id tensor per field, stored field after field ([T][B]): the layout the two-launch step sorts fastest
    tms = [b[0].t().contiguous() for b in batches] if (not sharded and not args.ids_sample_major) else None
    kws = [dict(kw, ids_tm=tms[j]) if tms else kw for j in range(len(batches))]
    prefetch_sort = not sharded and args.prefetch_sort and B <= model._fused_max
    if prefetch_sort:
        # a training loop knows its next batch: its ids are sorted beside this step (the sort needs nothing else)
        nb = len(batches)
        kws = [dict(k, next_ids=batches[(j + 1) % nb][0], **({'next_ids_tm': tms[(j + 1) % nb]} if tms else {}))
               for j, k in enumerate(kws)]
    for i in range(max(args.warmup, 2 * len(batches) if use_graph else 0)):   # every batch's graph captured before timing
        model.train_step(*batches[i % len(batches)], **kws[i % len(batches)])
    # a training loop that holds its next batches replays ONE graph of that many consecutive steps: the 4-5 us the
    # device idles between two graph launches are then paid once per replay, not once per step
    nb = len(batches)
    multi = use_graph and not prefetch_sort and not args.single_step_graphs and B <= model._fused_max and args.steps >= nb
    # steps per replay: the resident batches in turn, round after round (a whole number of rounds, so that every
    # replay starts at batch 0 - or the whole run when it is that short); a step count that is no multiple of it
    # ends with ONE shorter replay
    spg = (args.steps if args.steps <= args.steps_per_graph else max(nb, args.steps_per_graph // nb * nb)) if multi else 0
    rem = args.steps % spg if multi else 0

    def cycle(n):
        return [batches[j % nb] for j in range(n)], ([tms[j % nb] for j in range(n)] if tms else None)
    if multi:
        full, rest = cycle(spg), cycle(rem)
        for _ in range(2):                         # (captured before timing)
            model.train_steps(full[0], ids_tms=full[1])
            if rem > 1:
                model.train_steps(rest[0], ids_tms=rest[1])
    barrier()
    t0 = time.perf_counter()
    i = 0
    while i < args.steps:
        if multi and i % nb == 0 and i + spg <= args.steps:
            model.train_steps(full[0], ids_tms=full[1])       # steps i .. i + spg - 1
            i += spg
            continue
        if multi and rem > 1 and i % nb == 0 and i + rem == args.steps:
            model.train_steps(rest[0], ids_tms=rest[1])       # the last `rem` steps
            i += rem
            continue
        kw = kws[i % nb]
        if sharded and args.comm != 'rccl' and (args.prefetch or (world > 1 and not args.no_prefetch)):   # next batch's ids routed and exchanged beside this step's update
            kw = {'next_ids': batches[(i + 1) % nb][0]}
        model.train_step(*batches[i % nb], **kw)
        i += 1
    t_issue = time.perf_counter() - t0          # the host has enqueued every step; the device may still be running
    barrier()
    dt = time.perf_counter() - t0
    if sharded:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    from recommendation_amd import ops
    ops.check_error_word(model.err, "bench ids")

    result = None
    if rank == 0:
        ms = dt / args.steps * 1e3
        result = {
            'metric': 'train_samples_per_sec', 'value': round(B * world * args.steps / dt, 1), 'unit': 'samples/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(ms, 4),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': 'fm_criteo_%dx%d_b%d' % (len(rows), args.dim, B), 'global_batch': B * world,
                       'tables': len(rows), 'total_rows': int(sum(rows)), 'dim': args.dim, 'n_dense': 13,
                       'id_bits': args.idx_bits, 'ids': args.dist, 'seed': args.seed,
                       'id_layout': 'field-major [T][B]' if tms else 'sample-major [B][T]', 'optimizer': args.opt,
                       'launch': ('hip graph replay, %d consecutive steps per replay (the %d resident batches in turn)' % (spg, nb) if multi else
                                  'hip graph replay, one step per replay') if use_graph else 'eager',
                       'sort': ('next batch\'s ids sorted on a side stream beside this step' if prefetch_sort
                                else 'inside the step'),
                       'parallelism': 'single' if not sharded else 'rowwise-sharded tables x%d + replicated dense' % world},
        }
        # time the host needed to enqueue a step: when it equals ms_per_step the step is bound by the host's
        # launch / collective calls, not by the device
        result['host_issue_ms_per_step'] = round(t_issue / args.steps * 1e3, 4)
        if sharded:
            result['config']['next_batch_id_exchange'] = ('beside the update (side stream)'
                                                          if args.comm != 'rccl' and (args.prefetch or (world > 1 and not args.no_prefetch)) else 'in front of the step')
            result['config']['collectives'] = ('torch.distributed (nccl backend = RCCL)' if args.comm != 'rccl' else
                                               "the library's RCCL entry points on the compute stream")
    if not sharded:
        stages, U = stage_report(model, batches, args.stage_reps, args.idx_bits, tms)
        hbm = [s for s in stages if s['name'].startswith(('fm_fwd', 'fm_bwd_apply'))]
        upd = [s for s in stages if s['name'].startswith('launch 2')]
        dom = upd[0] if upd else max(hbm, key=lambda s: s['ms'])
        result['stages'] = stages
        result['stage_timing'] = ('HIP events on the launch stream; every stage cycles over the %d batches of the run, '
                                  'each with its own prepared inputs (> 256 MiB of distinct lines between two uses of a '
                                  'line)' % len(batches))
        result['unique_rows_per_step'] = round(U, 1)
        n_state = {'sgd': 0, 'adagrad_rowwise': 1.0 / args.dim, 'adam_tf': 2}[args.opt]
        result['roofline_inputs'] = {'B': B, 'T': len(rows), 'D': args.dim, 'U': round(U, 1), 'id_bytes': args.idx_bits // 8,
                                     'n_state': n_state, 'n_dense': 13}
        # HBM bytes per launch of that kernel from the PMC counters: they cannot be collected from inside
        # this process, so the committed measurement of this same workload is quoted (null for any other)
        traffic, traffic_source = None, None
        for fname in ('r03_fm_b8192_pmc.json', 'r02_fm_b8192_pmc.json', 'r01_fm_b8192_pmc.json'):
            pmc_path = os.path.join(ROOT, 'profiles', fname)
            if upd and os.path.exists(pmc_path):
                pmc = json.load(open(pmc_path))
                k = pmc.get('kernels', {}).get('seg_sorted_kernel +dense')
                if k and pmc.get('workload') == result['config']['workload'] and args.opt == 'adam_tf' \
                        and args.dist == 'uniform':
                    traffic, traffic_source = k['hbm_bytes_per_launch'], 'quoted profiles/%s' % fname
                    break
        result['roofline'] = {'kernel': dom['name'], 'bound': 'hbm', 'achieved': dom['gbs'], 'peak': HBM_PEAK_GBS,
                              'unit': 'GB/s', 'frac': round(dom['gbs'] / HBM_PEAK_GBS, 4), 'traffic': traffic,
                              'traffic_source': traffic_source,
                              'alg_bytes_per_launch': dom['alg_bytes'], 'ms_per_launch': dom['ms']}
        emb = [s for s in hbm]
        result['embedding_hbm_gbs'] = round(sum(s['alg_bytes'] for s in emb) / sum(s['ms'] for s in emb) / 1e6, 1)
        if not args.no_host_fed and use_graph and B <= model._fused_max:
            result['host_fed'] = host_fed_line(model, batches, min(args.steps, 200), dev, ms)
        if not args.no_cpu_baseline:
            result['cpu_baseline'] = cpu_baseline(args, rows, model, batches)
    if sharded and rank == 0:
        stages, n_owner, U = sharded_stage_report(model, batches[0], args.stage_reps, args.idx_bits)
        dom = stages[3]
        result['stages'] = stages
        result['entries_at_this_owner'] = n_owner
        result['unique_rows_per_step'] = U
        coll = ops.shard_block_ints(B, len(rows)) * 4 * world, B * (args.dim + 2) * 4 * world, model.rs * model.ld * 4
        result['collectives'] = {'all_to_all_rows_bytes_per_rank': coll[0], 'all_to_all_partials_bytes_per_rank': coll[1],
                                 'all_gather_rows_bytes_per_rank': coll[2]}
        result['roofline'] = {'kernel': dom['name'], 'bound': 'hbm', 'achieved': dom['gbs'], 'peak': HBM_PEAK_GBS,
                              'unit': 'GB/s', 'frac': round(dom['gbs'] / HBM_PEAK_GBS, 4), 'traffic': None,
                              'alg_bytes_per_launch': dom['alg_bytes'], 'ms_per_launch': dom['ms']}
    if sharded:
        torch.distributed.barrier()       # the other ranks wait for rank 0's stage timings before tearing down
    if saved_stdout is not None:
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        os.close(saved_stdout)
    if rank == 0:
        print(json.dumps(result), flush=True)
    if sharded:
        os.dup2(2, 1)                      # teardown messages, if any, do not follow the JSON line on stdout
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
