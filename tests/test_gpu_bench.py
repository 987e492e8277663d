"""bench.py as the driver runs it: every --config produces one JSON line with the contract's fields, and the N > 1 path
(one process per rank, here two ranks sharing the one GPU with gloo standing in for RCCL) gives the 1-rank eval result."""
import json
import os
import socket
import subprocess
import sys

import pytest

from util import ensure_lib

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAST = ['--warmup', '1', '--no-autotune', '--no-cpu-baseline']


def _run(args, world=1):
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for rank in range(world):
        env = dict(os.environ)
        if world > 1:
            env.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                       SAGEN_DIST_BACKEND='gloo')
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', str(world)] + args, env=env, cwd=ROOT,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=1200) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e[-3000:]
    lines = [l for o, _ in outs for l in o.splitlines() if l.startswith('{')]
    assert len(lines) == 1, lines                      # rank 0 prints ONE JSON line
    return json.loads(lines[0])


@pytest.mark.parametrize('config,batch', [('a', 10), ('av', 32), ('avf', 32)])
def test_bench_configs_emit_the_contract_line(config, batch):
    ensure_lib()
    r = _run(['--config', config, '--steps', '4'] + FAST)
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype',
              'data', 'config', 'roofline', 'one_in_flight', 'h2d_inclusive'):
        assert k in r, k
    assert r['n_gpus'] == 1 and r['steps'] == 4 and r['value'] > 0 and r['config']['windows_per_gpu_per_step'] == batch
    assert r['config']['name'] == config and r['config']['workload'].startswith('configs[')
    assert set(('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic')) <= set(r['roofline'])
    assert 0 < r['h2d_inclusive']['value'] and 0 < r['one_in_flight']['value']
    if config == 'av':       # the default run carries the other BASELINE configurations as short legs (fresh processes after the headline)
        for name, batch_l in (('a', 10), ('avf', 32), ('eval', 16), ('train', 32)):
            leg = r['leg_' + name]
            assert 'error' not in leg, leg
            assert leg['value'] > 0 and leg['ms_per_step'] > 0 and leg['steps'] > 0 and leg['windows_per_gpu_per_step'] == batch_l
            assert leg['roofline']['kernel'] and 0 < leg['roofline']['frac'] < 1 and leg['roofline']['peak'] > 0
    else:
        assert not any(k.startswith('leg_') for k in r)


def test_bench_eval_config_two_ranks_match_one_rank():
    """configs[3]: whole batches of ONE global window order dealt to the ranks; 2 ranks on this GPU must report the same
    metric means as 1 rank (the batches, hence the batch-norm statistics, are identical), from one JSON line."""
    ensure_lib()
    one = _run(['--config', 'eval', '--steps', '12', '--in-flight', '1'] + FAST)
    two = _run(['--config', 'eval', '--steps', '6', '--in-flight', '1'] + FAST, world=2)
    assert one['n_gpus'] == 1 and two['n_gpus'] == 2 and two['scaling'] == 'strong'
    # 1 rank timed batches 0..11 of its shard; with 2 ranks, rank 0 timed 0..5 and rank 1 timed 288..293: different windows of the
    # pool, so compare only the bookkeeping here and the values through equal window sets below
    assert two['config']['windows_per_s'] > 0 and two['steps'] == 6
    full1 = _run(['--config', 'eval', '--steps', '0', '--in-flight', '1'] + FAST)
    full2 = _run(['--config', 'eval', '--steps', '0', '--in-flight', '1'] + FAST, world=2)
    assert full1['steps'] == 576 and full2['steps'] == 288
    for k in ('stft', 'lsd', 'mse', 'snr'):
        for a, b in zip(full1['eval_metric_means'][k], full2['eval_metric_means'][k]):
            assert abs(a - b) <= 1e-6 * max(1.0, abs(a)), (k, a, b)


def test_bench_train_config_line_and_two_ranks():
    """configs[4]: one Adam step per `step`; the line carries its own roofline (dominant kernel of forward + backward) and phase
    split; two ranks on this GPU (gloo standing in for RCCL) exchange the gradient buckets and print one line."""
    ensure_lib()
    r = _run(['--config', 'train', '--steps', '3'] + FAST)
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data', 'config', 'roofline'):
        assert k in r, k
    assert r['config']['name'] == 'train' and r['config']['workload'].startswith('configs[4]') and r['value'] > 0
    assert set(('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic')) <= set(r['roofline'])
    ph = r['roofline']['phases_us_per_step']
    assert ph['wgrad'] > 0 and ph['dgrad'] > 0 and ph['forward'] > 0
    assert r['config']['last_loss'] == r['config']['last_loss']            # not NaN
    two = _run(['--config', 'train', '--steps', '2'] + FAST, world=2)
    assert two['n_gpus'] == 2 and two['value'] > 0 and 'gloo' in two['config']['gradient_exchange']


def test_four_ranks_share_one_tuned_plan():
    """--gpus 4 (gloo, four ranks on this GPU) WITH autotune: rank 0 tunes, the plan is broadcast, every rank asserts it runs the
    same kernels (bench.py) - for the headline configuration and for the training step."""
    ensure_lib()
    r = _run(['--config', 'av', '--steps', '2', '--warmup', '1', '--no-cpu-baseline', '--no-extra-legs', '--in-flight', '1'], world=4)
    assert r['n_gpus'] == 4 and r['value'] > 0 and r['config']['launch_plan'].startswith('autotuned')
    t = _run(['--config', 'train', '--steps', '2', '--warmup', '1', '--no-cpu-baseline'], world=4)
    assert t['n_gpus'] == 4 and t['value'] > 0 and t['config']['launch_plan'].startswith('autotuned')
