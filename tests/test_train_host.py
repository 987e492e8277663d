"""Host side of the training-step pieces (spatialaudiogen_amd/train.py): schedule, bucket layout, and the gradient exchange
over 2 gloo ranks (the code path RCCL runs on the GPUs)."""
import os
import socket
import sys

import numpy as np
import pytest

from oracle import np_oracle as O
from spatialaudiogen_amd import train as T
from spatialaudiogen_amd.weights import variable_specs

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_learning_rate_schedule_and_adam_step_size():
    for step in (0, 1, 9999, 10000, 25000, 150000):
        assert T.learning_rate(step, 1e-4, 10000, 0.5) == O.exponential_decay_staircase(1e-4, step, 10000, 0.5)
    assert T.learning_rate(29999, 1e-4, 10000, 0.5) == 1e-4 * 0.25
    for t in (1, 2, 10, 1000):
        assert T.adam_lr_t(t, 3e-4) == pytest.approx(3e-4 * np.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t), rel=1e-15)


def test_bucket_layout_covers_the_trainable_variables_once():
    specs = variable_specs(['audio', 'video'])
    layout, sizes = T.bucket_layout(specs, bucket_bytes=64 << 20)
    trainable = [n for n in specs if '/moving_' not in n]
    assert list(layout) == trainable and len(sizes) == 3         # 30.8 M parameters = 123 MB, no variable straddles: three buckets <= 64 MiB
    assert sum(n for _, _, n, _ in layout.values()) == 30810243 - sum(int(np.prod(specs[n])) for n in specs if '/moving_' in n)
    spans = {}
    for name, (b, off, n, shape) in layout.items():
        assert off % 4 == 0 and n == int(np.prod(shape)) and off + n <= sizes[b] <= (64 << 20) // 4
        spans.setdefault(b, []).append((off, off + n))
    for b, sp in spans.items():
        sp.sort()
        assert all(a[1] <= c[0] for a, c in zip(sp, sp[1:]))                  # no overlap
    small, ssz = T.bucket_layout(specs, bucket_bytes=1 << 20)                 # a variable larger than the bucket gets its own
    assert len(ssz) > 10 and max(ssz) == (12544 * 512 + 3) // 4 * 4


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist
    from spatialaudiogen_amd.dist import init_process_group
    from spatialaudiogen_amd import train as T
    init_process_group('gloo')
    specs = {'a/weights': (3, 5), 'b/biases': (7,), 'c/weights': (2, 2, 2, 2), 'c/bn/moving_mean': (2,)}
    st = T.AdamBuckets(specs, bucket_bytes=64)                               # tiny buckets: several of them
    for name in st.layout:
        st.view('grads', name).copy_(torch.full(st.layout[name][3], float(rank + 1)) * (1 + len(name)))
    for b in range(len(st.grads)):
        st.all_reduce(b)                                                      # per bucket, asynchronous
    st.wait()
    out = {name: st.view('grads', name).clone().numpy() for name in st.layout}
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, len(st.grads), out))


def test_gradient_buckets_all_reduce_two_ranks():
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] >= 2
    for _, _, out in res:
        assert set(out) == {'a/weights', 'b/biases', 'c/weights'}            # the moving average is not a trained variable
        for name, g in out.items():
            assert np.array_equal(g, np.full(g.shape, 3.0 * (1 + len(name)), np.float32))      # (1 + 2) x the per-rank value


# ------------------------------------------------------------------------------------------------------------------------
# the loop of train.py:202-234 around a stand-in trainer (no device): checkpoint retention, collective NaN abort
# ------------------------------------------------------------------------------------------------------------------------
class _FakeOpt(object):
    step = 0


class _FakeTrainer(object):
    """Trainer's loop-facing surface: step() -> (loss tensor, lr), save(), opt.step, batch."""
    batch = 2

    def __init__(self, nan_at=None):
        import torch
        self.torch = torch
        self.opt = _FakeOpt()
        self.nan_at = nan_at

    def step(self, audio, video, flow, target, mask=None):
        v = float('nan') if (self.nan_at is not None and self.opt.step == self.nan_at) else 1.0 / (1 + self.opt.step)
        self.opt.step += 1
        return self.torch.tensor(v), 1e-4

    def save(self, model_dir, global_step=None):
        from spatialaudiogen_amd.checkpoint import save_checkpoint
        prefix = os.path.join(model_dir, 'model.ckpt' + ('-%d' % global_step if global_step is not None else ''))
        save_checkpoint(prefix, {'step': np.asarray(self.opt.step, np.int32), 'w': np.arange(6, dtype=np.float32)})
        return prefix


def _endless():
    while True:
        yield (None, None, None, None, None)


def test_train_loop_keeps_one_periodic_checkpoint(tmp_path):
    """tf.train.Saver(max_to_keep=1) (train.py:176): the previous periodic bundle goes when the next one is complete; the final
    `model.ckpt` is written at exit and `checkpoint` names it."""
    from spatialaudiogen_amd.checkpoint import latest_checkpoint, load_checkpoint
    d = str(tmp_path)
    hist = T.train_loop(_FakeTrainer(), _endless(), d, n_iters=9, log_every=2, ckpt_every=2, log=lambda *_: None)
    assert [h[0] for h in hist] == [0, 2, 4, 6, 8]
    files = sorted(os.listdir(d))
    assert files == ['checkpoint', 'model.ckpt-9.data-00000-of-00001', 'model.ckpt-9.index', 'model.ckpt.data-00000-of-00001',
                     'model.ckpt.index'], files                      # saved after steps 2, 4, 6, 8 (opt.step 3, 5, 7, 9): only the last is left
    assert latest_checkpoint(d) == os.path.join(d, 'model.ckpt')
    assert int(load_checkpoint(latest_checkpoint(d))['step']) == 9


def test_resume_never_deletes_bundles_this_run_did_not_write(tmp_path):
    """tf.train.Saver(max_to_keep=1) only deletes bundles it saved in the current session (saver.restore registers nothing): bundles
    a previous process, another experiment or the user left in model_dir - including the one --resume restored - survive; this
    run's own periodic saves still keep one."""
    from spatialaudiogen_amd.checkpoint import save_checkpoint
    d = str(tmp_path)
    for n in (3, 5):                                                 # left by an earlier run / kept by hand
        save_checkpoint(os.path.join(d, 'model.ckpt-%d' % n), {'step': np.asarray(n, np.int32), 'w': np.zeros(6, np.float32)})
    save_checkpoint(os.path.join(d, 'milestone.ckpt-5'), {'step': np.asarray(5, np.int32), 'w': np.zeros(6, np.float32)})
    tr = _FakeTrainer()
    tr.opt.step = 5
    T.train_loop(tr, _endless(), d, n_iters=11, init_step=5, log_every=2, ckpt_every=2, log=lambda *_: None)
    files = sorted(f for f in os.listdir(d) if f.endswith('.index'))
    # this run saved after steps 6, 8, 10 (opt.step 7, 9, 11): only its last periodic bundle is left of those
    assert files == ['milestone.ckpt-5.index', 'model.ckpt-11.index', 'model.ckpt-3.index', 'model.ckpt-5.index', 'model.ckpt.index'], files


class _FakeCtx(object):
    def __init__(self, jump_at):
        self.reads, self.jump_at = 0, jump_at

    def counter(self, name):
        assert name == 'fp16x2_saturations'
        self.reads += 1
        return 0 if self.reads <= self.jump_at else 17


def test_train_loop_stops_when_the_fp16x2_planes_saturate(tmp_path):
    """The fp16x2 activation / gradient planes clamp (and count) elements beyond their statistical range bound; the loop reads the
    counter at every log step and refuses to go on training on clamped values."""
    tr = _FakeTrainer()
    tr.ctx = _FakeCtx(jump_at=3)                                     # read 1: at loop entry; reads 2, 3: log steps 0, 2; read 4: step 4
    with pytest.raises(FloatingPointError, match='clamped 17 elements'):
        T.train_loop(tr, _endless(), str(tmp_path), n_iters=20, log_every=2, ckpt_every=1000, log=lambda *_: None)
    assert tr.opt.step == 5


def test_checkpoint_write_is_atomic(tmp_path, monkeypatch):
    """A writer that dies between the data file and the state file must leave --resume pointing at the previous complete bundle."""
    from spatialaudiogen_amd import checkpoint as C
    d = str(tmp_path)
    C.save_checkpoint(os.path.join(d, 'model.ckpt-1'), {'w': np.ones(3, np.float32)})
    assert C.latest_checkpoint(d) == os.path.join(d, 'model.ckpt-1')
    real, calls = os.replace, []

    def dying_replace(a, b):
        calls.append(b)
        if len(calls) == 2:                                         # .data renamed, the .index rename "is killed"
            raise KeyboardInterrupt()
        return real(a, b)
    monkeypatch.setattr(os, 'replace', dying_replace)
    with pytest.raises(KeyboardInterrupt):
        C.save_checkpoint(os.path.join(d, 'model.ckpt-2'), {'w': np.zeros(3, np.float32)})
    monkeypatch.setattr(os, 'replace', real)
    assert C.latest_checkpoint(d) == os.path.join(d, 'model.ckpt-1')
    assert np.array_equal(C.load_checkpoint(C.latest_checkpoint(d))['w'], np.ones(3, np.float32))
    C.save_checkpoint(os.path.join(d, 'model.ckpt-3'), {'w': np.full(3, 3, np.float32)})
    assert not [f for f in os.listdir(d) if '.tmp-' in f and 'ckpt-3' in f]
    C.remove_checkpoint(os.path.join(d, 'model.ckpt-1'))
    assert not os.path.exists(os.path.join(d, 'model.ckpt-1.index'))


def _nan_worker(rank, world, port, tmp, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    import torch.distributed as dist
    from spatialaudiogen_amd.dist import init_process_group
    from spatialaudiogen_amd import train as T
    import test_train_host as me
    init_process_group('gloo')
    tr = me._FakeTrainer(nan_at=4 if rank == 1 else None)          # only rank 1 ever sees a NaN
    err, hist = None, []
    try:
        T.train_loop(tr, me._endless(), os.path.join(tmp, 'r%d' % rank), n_iters=20, log_every=2, ckpt_every=1000,
                     log=lambda *_: None)
    except ValueError as e:
        err = str(e)
    dist.barrier()                                                   # both ranks left the loop: nobody hangs in a collective
    dist.destroy_process_group()
    q.put((rank, err, tr.opt.step))


def test_nan_on_one_rank_stops_every_rank_in_the_same_step(tmp_path):
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    for r in range(2):
        os.makedirs(str(tmp_path / ('r%d' % r)))
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_nan_worker, args=(r, 2, port, str(tmp_path), q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, err, steps in res:
        assert err and 'NaN' in err, (rank, err)
        assert steps == 5                                            # the NaN appears in step 4, a logged step: both stop after it


def _fake_resnet18_blob(rng, drop=None, reshape=None):
    from spatialaudiogen_amd.weights import resnet18_specs
    blob = {}
    for name, shape in resnet18_specs('S').items():
        key = name[2:]
        if key == drop:
            continue
        blob[key] = rng.standard_normal(shape if key != reshape else tuple(shape) + (1,)).astype(np.float32)
    return blob


def test_pretrained_resnet18_initialises_both_trunks_by_name(tmp_path):
    """--pretrained (resnet.py:238-249 at train.py:182-184) on a synthetic blob: every model variable of BOTH trunks (weights, shortcuts
    and the four batch-norm vectors per layer) takes `blob[name without scope]`; nothing else moves; the file form is the reference's
    (one pickled dict in a .npy); a missing key is a KeyError and a wrong shape an error, with nothing assigned."""
    from spatialaudiogen_amd.weights import variable_specs, init_weights, load_pretrained_resnet18, resnet18_specs
    enc = ['audio', 'flow', 'video']
    specs = variable_specs(enc)
    P0 = init_weights(specs, seed=3, mode='bench')
    blob = _fake_resnet18_blob(np.random.default_rng(0))
    fn = str(tmp_path / 'resnet18.npy')
    np.save(fn, np.array(blob, dtype=object), allow_pickle=True)
    P = dict((k, np.array(v)) for k, v in P0.items())
    names = load_pretrained_resnet18(P, fn, specs=specs)
    assert len(names) == 2 * len(resnet18_specs('S')) and len(names) == sum(1 for k in specs if k.split('/')[0] in ('video_encoder', 'flow_encoder'))
    for k in specs:
        scope = k.split('/')[0]
        if scope in ('video_encoder', 'flow_encoder'):
            assert np.array_equal(P[k], blob[k[len(scope) + 1:]]), k
        else:
            assert np.array_equal(P[k], P0[k]), k
    assert np.array_equal(P['video_encoder/conv3_1/shortcut/weights'], P['flow_encoder/conv3_1/shortcut/weights'])
    assert np.array_equal(P['video_encoder/conv5_2/conv_2/bn/moving_variance'], blob['conv5_2/conv_2/bn/moving_variance'])
    # audio + video only: the flow trunk is not in the graph, nothing of it is touched or required
    specs_av = variable_specs(['audio', 'video'])
    Pav = init_weights(specs_av, seed=3, mode='bench')
    assert len(load_pretrained_resnet18(Pav, blob, specs=specs_av)) == len(resnet18_specs('S'))
    for bad, err in ((_fake_resnet18_blob(np.random.default_rng(1), drop='conv4_2/conv_1/bn/gamma'), KeyError),
                     (_fake_resnet18_blob(np.random.default_rng(1), reshape='conv2_1/conv_1/weights'), ValueError)):
        Q = dict((k, np.array(v)) for k, v in P0.items())
        with pytest.raises(err):
            load_pretrained_resnet18(Q, bad, specs=specs)
        assert all(np.array_equal(Q[k], P0[k]) for k in specs)


def test_train_cli_accepts_pretrained():
    a = T.parse_arguments(['db', 'model', '--pretrained', '/x/resnet18.npy', '--synthetic'])
    assert a.pretrained == '/x/resnet18.npy'
    assert T.parse_arguments(['db', 'model']).pretrained is None
