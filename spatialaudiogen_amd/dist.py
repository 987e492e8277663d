"""Multi-GPU: one process per GPU, windows/clips sharded over ranks, no data-path collective.

The reference is single-device (deploy.py:156, eval.py:33).  Clips are independent given whole
batches (training-mode batch-norm couples windows of a batch), so each rank owns a
full weight replica and a contiguous block of clips; the only exchange is the eval-time reduction
of metric sums (model.py:122-150 families + count), one all-reduce over RCCL ('nccl' on ROCm) —
or gloo on CPU for tests.
"""
import os


def shard_range(n_items, rank, world_size):
    """Contiguous block [lo, hi) of rank `rank`: sizes differ by at most one, order preserved."""
    base, rem = divmod(n_items, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def init_process_group(backend=None):
    import torch
    import torch.distributed as dist
    if dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    world = int(os.environ.get('WORLD_SIZE', 1))
    if world == 1:
        return 0, 1
    if backend is None:          # RCCL ('nccl') on GPUs; SAGEN_DIST_BACKEND=gloo lets several ranks share ONE GPU (tests)
        backend = os.environ.get('SAGEN_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    dist.init_process_group(backend, rank=int(os.environ['RANK']), world_size=world)
    return dist.get_rank(), world


class MetricReducer(object):
    """Per-rank float64 sums per metric + sample count -> global means with ONE all-reduce (3*18+1 doubles for the on-graph
    metrics of eval.py:125-133).  `reduce()` returns the reference's statistic: np.mean over ALL samples (eval.py:223), so a
    non-finite per-sample value makes that metric's mean non-finite, exactly as in the reference.  The means over the finite
    samples only, and how many there were, are kept beside it (`finite_means`, `finite_counts`) as a diagnostic."""

    def __init__(self, names, device=None):
        import torch
        self.names = list(names)
        self.k = len(self.names)
        self.buf = torch.zeros(3 * self.k + 1, dtype=torch.float64, device=device)
        self.finite_means, self.finite_counts = {}, {}

    def add_rows(self, rows):
        """rows [n, k]: one row of metric values per sample."""
        import numpy as np
        import torch
        rows = np.asarray(rows, np.float64).reshape(-1, self.k)
        ok = np.isfinite(rows)
        with np.errstate(invalid='ignore'):
            upd = np.concatenate([rows.sum(0), np.where(ok, rows, 0.0).sum(0), ok.sum(0).astype(np.float64), [float(rows.shape[0])]])
        self.buf += torch.as_tensor(upd, dtype=torch.float64, device=self.buf.device)

    def add(self, values, count):
        """`count` samples that all have the metric values `values`."""
        import numpy as np
        self.add_rows(np.tile(np.asarray(values, np.float64).reshape(1, -1), (int(count), 1)))

    def reduce(self):
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(self.buf)
        k = self.k
        n = float(self.buf[-1].item())
        sums, fsums, cnts = self.buf[:k], self.buf[k:2 * k], self.buf[2 * k:3 * k]
        vals = (sums / max(n, 1.0)).tolist()
        self.finite_means = dict(zip(self.names, (fsums / cnts.clamp(min=1.0)).tolist()))
        self.finite_counts = dict(zip(self.names, [int(c) for c in cnts.tolist()]))
        return dict(zip(self.names, vals)), int(n)
