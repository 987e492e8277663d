"""The training step on the HIP path (reference train.py:137-236).

`Trainer.step` is one `sess.run(train_op)` (train.py:208): forward with retained activations, the loss the reference minimises
(`stft/avg`, model.py:156-159), the backward pass of the whole network (csrc/train_model.hip: weight gradients by wgrad_kernel,
data gradients on the forward's contraction kernels, training-mode batch-norm backward, the mask / iSTFT adjoint), the gradient
exchange (`AdamBuckets.all_reduce`: one RCCL sum all-reduce per bucket) and the optimiser (`AdamBuckets`: tf.train.AdamOptimizer
over flat parameter buckets, one fused launch per bucket, under the staircase schedule `learning_rate`), plus the contrib
batch_norm moving-average updates (UPDATE_OPS, train.py:147-148).  `train` is the loop of train.py:192-234 (NaN guard, periodic
checkpoints, --resume).

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
                v = variables[name]
                v = v.detach().to(dtype=torch.float32).reshape(-1) if isinstance(v, torch.Tensor) else torch.as_tensor(np.asarray(v, np.float32).reshape(-1))
                self.params[b][off:off + n] = v.to(self.params[b].device)

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
    mk = channel_mask(mask, B, pr.device)
    grad = torch.empty_like(pr) if need_grad else None
    loss = torch.zeros(1, dtype=torch.float64, device=pr.device)
    ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    check(_lib.lib().sagen_stft_loss_grad(ptr(pr), ptr(gt), ptr(mk), B, ptr(grad), ptr(loss),
                                          C.c_void_p(torch.cuda.current_stream(pr.device).cuda_stream)))
    return loss[0], grad


def channel_mask(mask, B, device):
    """[B,3] float32 channel mask of the predicted channels (Y,Z,X) or None.  The feeder's masks are [B,4] over W,Y,Z,X
    (feeder.py:312-314); train.py:127 / eval.py slice off the input channels (`[:, ambi_order**2:]`) - accepted here too."""
    import torch
    if mask is None:
        return None
    mk = torch.as_tensor(mask).to(device=device, dtype=torch.float32)
    if tuple(mk.shape) == (B, 4):
        mk = mk[:, 1:]
    if tuple(mk.shape) != (B, 3):
        raise ValueError('channel mask must be [B,3] (Y,Z,X) or [B,4] (W,Y,Z,X), got %s' % (tuple(mk.shape),))
    return mk.contiguous()


class Trainer(object):
    """One model replica + optimiser state on one GPU (one process per GPU; gradients summed over the ranks with RCCL).

        net = SptAudioGen(1, encoders=['audio', 'video'], separation='unet_mask'); net.load_variables(P)
        tr = Trainer(net, batch=32, lr=1e-4)
        loss = tr.step(audio, video, None, target)          # audio [B,52799,1], video [B,1,224,448,3], target [B,4800,3]

    The parameters live in `self.opt.params` (flat buckets); the native context is bound to views of them, so the fused Adam
    update is visible to the next step without a copy (the step re-packs the filters, as the variables changed)."""

    def __init__(self, net, batch, lr=1e-4, lr_iters=10000, lr_decay=1.0, bucket_bytes=64 << 20, variables=None):
        import torch
        from . import _lib
        from ._lib import check, SagenTensor
        from .model import _Ctx
        from .definitions import FREQ_MASK
        if net.separation != FREQ_MASK:
            raise ValueError("training implements separation 'unet_mask' (what train.py trains)")
        self.net, self.batch, self.device = net, batch, net.device
        specs = net.variable_specs()
        src = variables if variables is not None else net._variables
        if src is None:
            raise RuntimeError('load_variables() first (or pass variables=)')
        self.opt = AdamBuckets(OrderedDict((k, v) for k, v in specs.items()), variables=src, lr=lr, lr_iters=lr_iters, lr_decay=lr_decay,
                               bucket_bytes=bucket_bytes, device=self.device)
        # BN moving averages: not trained, updated by assignment (core.py:210 / train.py:147-148)
        self.moving = OrderedDict()
        for k, shape in specs.items():
            if '/moving_' in k:
                v = src.get(k) if hasattr(src, 'get') else None
                t = torch.as_tensor(np.asarray(v.cpu() if isinstance(v, torch.Tensor) else v, np.float32)) if v is not None else \
                    (torch.zeros(shape) if k.endswith('moving_mean') else torch.ones(shape))
                self.moving[k] = t.to(device=self.device, dtype=torch.float32).contiguous()
        live = OrderedDict(self.opt.variables())
        live.update(self.moving)
        self.ctx = _Ctx(net._config(batch), live, self.device)
        l = _lib.lib()
        nbytes = int(l.sagen_train_workspace_bytes(self.ctx.handle))
        self.train_ws = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=self.device)

        def tensors(d):
            arr = (SagenTensor * len(d))()
            keep = []
            for i, (k, t) in enumerate(d.items()):
                keep.append(k.encode())
                arr[i].name, arr[i].data, arr[i].ndim = keep[-1], t.data_ptr(), t.dim()
                for j, sdim in enumerate(t.shape):
                    arr[i].shape[j] = sdim
            return arr, keep
        grads = OrderedDict((k, self.opt.view('grads', k)) for k in self.opt.layout)
        ga, self._k1 = tensors(grads)
        ma, self._k2 = tensors(self.moving)
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        check(l.sagen_train_bind(self.ctx.handle, ga, len(grads), ma, len(self.moving), C.c_void_p(self.train_ws.data_ptr()),
                                 self.train_ws.numel() * 4, stream))
        self.loss = torch.zeros(1, dtype=torch.float64, device=self.device)
        self.pred = torch.empty(batch, 4800, 3, dtype=torch.float32, device=self.device)

    def _prep(self, t, tail):
        import torch
        if t is None:
            return None
        t = torch.as_tensor(np.asarray(t)) if not isinstance(t, torch.Tensor) else t
        t = t.to(device=self.device, dtype=torch.float32).contiguous()
        if tuple(t.shape) != (self.batch,) + tail:
            raise ValueError('expected shape %s, got %s' % ((self.batch,) + tail, tuple(t.shape)))
        return t

    def forward_backward(self, audio, video, flow, target, mask=None, update_moving=True):
        """Forward + loss + backward; gradients land in self.opt.grads.  Returns the loss as a device fp64 tensor (no sync)."""
        import torch
        from . import _lib
        from ._lib import check
        from .definitions import VIDEO, FLOW
        a = self._prep(audio, (52799, 1))
        v = self._prep(video, (1, 224, 448, 3)) if VIDEO in self.net.encoders else None
        f = self._prep(flow, (1, 224, 448, 3)) if FLOW in self.net.encoders else None
        t = self._prep(target, (4800, 3))
        mk = channel_mask(mask, self.batch, self.device)
        p = lambda x: C.c_void_p(x.data_ptr()) if x is not None else None
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        check(_lib.lib().sagen_train_step(self.ctx.handle, p(a), p(v), p(f), p(t), p(mk), p(self.pred), p(self.loss), int(update_moving), stream))
        return self.loss

    def step(self, audio, video, flow, target, mask=None):
        """One training iteration (train.py:208): returns (loss tensor on device, learning rate used)."""
        loss = self.forward_backward(audio, video, flow, target, mask)
        self.opt.all_reduce()
        lr = self.opt.apply()
        return loss, lr

    def grad(self, name):
        return self.opt.view('grads', name)

    def buffer(self, name):
        """Named buffer of the native train workspace (parity tests)."""
        from . import _lib
        from ._lib import check
        data, n = C.c_void_p(), C.c_size_t()
        check(_lib.lib().sagen_train_get_buffer(self.ctx.handle, name.encode(), C.byref(data), C.byref(n)))
        for ws in (self.train_ws, self.ctx.workspace):
            off = (data.value - ws.data_ptr()) // 4
            if 0 <= off and off + n.value <= ws.numel():
                return ws[off:off + n.value]
        raise RuntimeError('buffer %s lies outside the workspaces' % name)

    def variables(self):
        """Current values of every variable (checkpoint content, deploy.py:79 / train.py:223-225)."""
        out = OrderedDict((k, v.clone()) for k, v in self.opt.variables().items())
        out.update((k, v.clone()) for k, v in self.moving.items())
        return out
