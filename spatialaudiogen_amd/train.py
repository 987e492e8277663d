"""Training-step pieces on the HIP path (reference train.py:137-236).

What exists: the loss the reference minimises and its gradient with respect to the prediction (`stft_loss`), the optimiser
(`AdamBuckets`: tf.train.AdamOptimizer over flat parameter buckets, one fused launch per bucket), its learning-rate schedule
(`learning_rate`), and the gradient exchange (`AdamBuckets.all_reduce`: one RCCL sum all-reduce per bucket, launched as soon as
a bucket's gradients are complete).  What does not exist yet: the backward pass of the network itself (conv / conv-transpose /
FC data and weight gradients, training-mode batch-norm backward, the mask / iSTFT adjoint) - so there is no end-to-end
training step; DESIGN.md 7 lists the missing kernels.

Layout (MI355X-first, not the reference's per-variable TF ops): every variable lives at a fixed offset of ONE flat fp32
parameter buffer per bucket, and the gradients, the Adam m and v slots are buffers of the same layout.  Backward kernels write
their weight gradients straight into the gradient bucket, the all-reduce runs over whole buckets (xGMI rings are per-link
bound: few, large messages - at most 64 MiB each by default, i.e. three buckets for the 123 MB of the audio+video model), and Adam is one
elementwise pass per bucket instead of one launch per variable (the audio+video model has 154 variables).
"""
import ctypes as C
from collections import OrderedDict

import numpy as np

ADAM_BETA1, ADAM_BETA2, ADAM_EPSILON = 0.9, 0.999, 1e-8          # tf.train.AdamOptimizer defaults (myutils.py:220 passes only lr)


def learning_rate(step, lr, lr_iters, lr_decay):
    """tf.train.exponential_decay(lr, step, lr_iters, lr_decay, staircase=True) (myutils.py:215-218)."""
    return lr * lr_decay ** (int(step) // int(lr_iters))


def adam_lr_t(step_count, lr, beta1=ADAM_BETA1, beta2=ADAM_BETA2):
    """Bias-corrected step size of TF's Adam at its t-th application (t = 1, 2, ...)."""
    return lr * np.sqrt(1.0 - beta2 ** step_count) / (1.0 - beta1 ** step_count)


def bucket_layout(specs, bucket_bytes=64 << 20, trainable=None):
    """Assign every trainable variable a (bucket, offset) in declaration order; a variable never straddles buckets and
    every offset is a multiple of 4 floats (16 bytes).  specs: OrderedDict name -> shape.  BN moving averages are not
    trained (they are updated by assignment, core.py:6 / train.py:147-148)."""
    if trainable is None:
        trainable = lambda n: '/moving_' not in n
    cap = max(bucket_bytes // 4, 4)
    layout, sizes = OrderedDict(), []
    used = 0
    for name, shape in specs.items():
        if not trainable(name):
            continue
        n = int(np.prod(shape)) if len(shape) else 1
        n4 = (n + 3) // 4 * 4
        if sizes and used + n4 > cap and used > 0:
            sizes[-1] = used
            used = 0
            sizes.append(0)
        if not sizes:
            sizes.append(0)
        layout[name] = (len(sizes) - 1, used, n, tuple(shape))
        used += n4
    if sizes:
        sizes[-1] = used
    return layout, sizes


class AdamBuckets(object):
    """Parameters, gradients and Adam slots as flat buckets (see module docstring)."""

    def __init__(self, specs, variables=None, lr=1e-4, lr_iters=10000, lr_decay=1.0, bucket_bytes=64 << 20, device=None):
        import torch
        self.layout, self.sizes = bucket_layout(specs, bucket_bytes)
        self.device = device
        mk = lambda: [torch.zeros(n, dtype=torch.float32, device=device) for n in self.sizes]
        self.params, self.grads, self.m, self.v = mk(), mk(), mk(), mk()
        self.lr, self.lr_iters, self.lr_decay = lr, lr_iters, lr_decay
        self.step = 0
        self._pending = []
        if variables is not None:
            for name, (b, off, n, shape) in self.layout.items():
                self.params[b][off:off + n] = torch.as_tensor(np.asarray(variables[name], np.float32).reshape(-1))

    def view(self, which, name):
        """Tensor view of variable `name` inside bucket list `which` ('params' | 'grads' | 'm' | 'v')."""
        b, off, n, shape = self.layout[name]
        return getattr(self, which)[b][off:off + n].view(shape)

    def variables(self):
        return OrderedDict((name, self.view('params', name)) for name in self.layout)

    def all_reduce(self, bucket=None, async_op=True):
        """Sum all-reduce of the gradient bucket(s) over the ranks (RCCL over xGMI; gloo in the CPU tests).  Call per bucket
        as soon as the backward pass has produced its last gradient, `wait()` before `apply()`."""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
            return
        for b in ([bucket] if bucket is not None else range(len(self.grads))):
            w = dist.all_reduce(self.grads[b], async_op=async_op)
            if async_op:
                self._pending.append(w)

    def wait(self):
        for w in self._pending:
            w.wait()
        self._pending = []

    def apply(self):
        """One optimiser step on every bucket: lr from the staircase schedule at the current global step, gradients averaged
        over the ranks (the all-reduce summed them)."""
        import torch
        import torch.distributed as dist
        from . import _lib
        from ._lib import check
        self.wait()
        world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        lr = learning_rate(self.step, self.lr, self.lr_iters, self.lr_decay)
        self.step += 1
        lr_t = adam_lr_t(self.step, lr)
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream) if self.params and self.params[0].is_cuda else None
        for p, g, m, v in zip(self.params, self.grads, self.m, self.v):
            if not p.is_cuda:
                raise RuntimeError('AdamBuckets.apply needs device buckets: the optimiser kernel has no CPU implementation')
            check(_lib.lib().sagen_adam_update(C.c_void_p(p.data_ptr()), C.c_void_p(g.data_ptr()), C.c_void_p(m.data_ptr()),
                                               C.c_void_p(v.data_ptr()), p.numel(), lr_t, ADAM_BETA1, ADAM_BETA2, ADAM_EPSILON,
                                               1.0 / world, stream))
        return lr


def stft_loss(pred, target, mask=None, need_grad=True):
    """(loss fp64 scalar tensor, dL/dpred [B,4800,3] or None) of the reference's training loss `stft/avg` (model.py:122-127,
    156-159) on the device.  mask [B,3]: channel mask of the WXY-only clips (feeder.py:312-314)."""
    import torch
    from . import _lib
    from ._lib import check
    pr = pred.to(torch.float32).contiguous()
    gt = target.to(device=pr.device, dtype=torch.float32).contiguous()
    B = pr.shape[0]
    if tuple(pr.shape) != (B, 4800, 3) or tuple(gt.shape) != tuple(pr.shape):
        raise ValueError('predictions / targets must be [B, 4800, 3]')
    mk = None if mask is None else torch.as_tensor(mask).to(device=pr.device, dtype=torch.float32).contiguous()
    grad = torch.empty_like(pr) if need_grad else None
    loss = torch.zeros(1, dtype=torch.float64, device=pr.device)
    ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    check(_lib.lib().sagen_stft_loss_grad(ptr(pr), ptr(gt), ptr(mk), B, ptr(grad), ptr(loss),
                                          C.c_void_p(torch.cuda.current_stream(pr.device).cuda_stream)))
    return loss[0], grad
