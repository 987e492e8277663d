"""N > 1 path on CPU: two gloo ranks shard a clip list without overlap and reduce their metric sums
with one all-reduce — the same code path bench.py / an eval driver runs over RCCL."""
import os
import socket
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    import torch.distributed as dist
    from spatialaudiogen_amd.dist import init_process_group, shard_range, MetricReducer
    r, w = init_process_group('gloo')
    assert (r, w) == (rank, world)
    lo, hi = shard_range(1024, r, w)                       # 1024 clips -> contiguous blocks
    red = MetricReducer(['stft/avg', 'lsd/avg', 'mse/avg'])
    for clip in range(lo, hi):                             # per-clip "metrics" = functions of the clip id
        red.add([clip, 2.0 * clip, 1.0], 9)                # 9 windows per clip (skip_rate=10 of a 10 s clip)
    vals, n = red.reduce()
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, lo, hi, vals, n))


@pytest.mark.parametrize('world', [2, 3])
def test_sharded_metric_allreduce(world):
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == 0 and res[-1][2] == 1024 and all(res[i][2] == res[i + 1][1] for i in range(world - 1))
    mean_id = sum(range(1024)) / 1024.0
    for _, _, _, vals, n in res:                           # every rank holds the same global means
        assert n == 1024 * 9
        assert abs(vals['stft/avg'] - mean_id) < 1e-9 and abs(vals['lsd/avg'] - 2 * mean_id) < 1e-9 and abs(vals['mse/avg'] - 1.0) < 1e-12
