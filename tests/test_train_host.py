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
