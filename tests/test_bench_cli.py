"""The benchmark's command line: defaults, the launcher check, and (on a GPU) that each workload
prints exactly one JSON line with the fields the harness reads."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, 'bench.py')

REQUIRED = ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
            'vs_baseline', 'dtype', 'data', 'config', 'roofline')


def _run(*argv, timeout=300):
    env = dict(os.environ)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK'):
        env.pop(k, None)
    return subprocess.run([sys.executable, BENCH] + list(argv), cwd=ROOT, env=env, capture_output=True, text=True,
                          timeout=timeout)


def test_defaults(monkeypatch):
    sys.path.insert(0, ROOT)
    import bench
    monkeypatch.setattr(sys, 'argv', ['bench.py'])
    a = bench.parse_args()
    assert (a.gpus, a.workload, a.batch, a.dim, a.opt, a.idx_bits) == (1, 'fm', 8192, 16, 'adam_tf', 64)
    assert a.steps > 0 and a.warmup >= 0
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--workload', 'embag', '--opt', 'adagrad_rowwise'])
    assert bench.parse_args().opt == 'adagrad_rowwise'


def test_more_gpus_than_the_machine_has_is_refused():
    """Without a launcher bench.py starts its own ranks, after checking that the GPUs are there."""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("this machine has the GPUs")
    r = _run('--gpus', '2', timeout=120)
    assert r.returncode != 0 and 'GPU' in r.stderr and r.stdout.strip() == ''


def _spawn(tmp_path, monkeypatch, body):
    sys.path.insert(0, ROOT)
    import bench
    import torch
    script = tmp_path / 'rank.py'
    script.write_text(body)
    monkeypatch.setattr(torch.cuda, 'device_count', lambda: 2)
    return bench.spawn_ranks(2, script=str(script), argv=['--x'])


def test_self_launch_starts_one_process_per_rank_and_relays_rank_0(tmp_path, monkeypatch, capfd):
    rc = _spawn(tmp_path, monkeypatch, """
import json, os, sys
assert sys.argv[1:] == ['--x'] and os.environ['WORLD_SIZE'] == '2' and os.environ['MASTER_ADDR'] == '127.0.0.1'
assert os.environ['RANK'] == os.environ['LOCAL_RANK'] and int(os.environ['MASTER_PORT']) > 0
print('banner that is not JSON')
print(json.dumps({'rank': int(os.environ['RANK'])}))
""")
    assert rc == 0
    assert capfd.readouterr().out.strip() == '{"rank": 0}'


def test_self_launch_fails_when_a_rank_fails_and_ends_the_others(tmp_path, monkeypatch, capfd):
    import time
    t0 = time.time()
    rc = _spawn(tmp_path, monkeypatch, """
import os, sys, time
if os.environ['RANK'] == '1':
    sys.exit(3)
time.sleep(600)     # rank 0 would wait for ever in a collective
""")
    assert rc == 3 and time.time() - t0 < 60
    assert capfd.readouterr().out.strip() == ''


def _one_json_line(r):
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [x for x in r.stdout.splitlines() if x.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    for k in REQUIRED:
        assert k in d, k
    assert d['value'] > 0 and d['ms_per_step'] > 0 and 'workload' in d['config']
    rf = d['roofline']
    assert rf['bound'] in ('hbm', 'mfma') and rf['peak'] > 0 and abs(rf['frac'] - rf['achieved'] / rf['peak']) < 1e-3
    return d


@pytest.mark.gpu
def test_fm_line(dev):
    d = _one_json_line(_run('--steps', '5', '--warmup', '2', '--stage-reps', '2', '--cpu-steps', '1'))
    assert d['config']['workload'] == 'fm_criteo_26x16_b8192' and d['n_gpus'] == 1 and d['steps'] == 5
    assert d['cpu_baseline']['kind'] == 'port' and d['cpu_baseline']['value'] > 0
    assert d['config']['id_layout'].startswith('field-major')


@pytest.mark.gpu
@pytest.mark.parametrize("opt", ['adam_tf', 'adagrad_rowwise', 'sgd'])
def test_embag_line_small_tables(dev, opt):
    d = _one_json_line(_run('--workload', 'embag', '--embag-rows', '2600000', '--opt', opt, '--steps', '3', '--warmup',
                            '1', '--stage-reps', '2', '--no-cpu-baseline'))
    assert d['config']['dim'] == 128 and d['config']['optimizer'].startswith(opt)
    assert d['embedding_hbm_gbs'] > 0 and len(d['stages']) == 3


@pytest.mark.gpu
def test_sharded_embag_line_one_rank_over_rccl(dev):
    d = _one_json_line(_run('--workload', 'sharded_embag', '--embag-rows', '800000', '--batch', '1024', '--steps', '3',
                            '--warmup', '1', '--stage-reps', '2', '--cpu-steps', '1'))
    assert d['config']['dim'] == 128 and d['n_gpus'] == 1 and d['config']['global_batch'] == 1024
    assert len(d['stages']) == 7 and d['cpu_baseline']['value'] > 0


@pytest.mark.gpu
def test_colour_line(dev):
    d = _one_json_line(_run('--workload', 'colour', '--steps', '5', '--warmup', '2', '--cpu-steps', '3'))
    assert d['config']['workload'] == 'colour_model_train_b300' and d['cpu_baseline']['kind'] == 'port'
