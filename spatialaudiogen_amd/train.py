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

    def all_reduce_after(self, events, comm_stream, timing=None):
        """The same exchange, overlapped with the backward pass: bucket b's all-reduce is enqueued on `comm_stream` behind
        events[b], which the native step records as soon as the last gradient of bucket b has been enqueued
        (sagen_train_set_grad_events).  Buckets follow the declaration order of the variables = the order of the forward, so the
        backward completes them last to first.
        timing (a list, measurement only): per bucket a pair of timing events on the communication stream is appended - (bucket,
        ready = its gradients are complete and the stream is free, done = its all-reduce has finished); `comm_timing` turns them
        into the time the stream spent waiting for gradients vs exchanging them."""
        import torch
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
            return
        for b in reversed(range(len(self.grads))):
            comm_stream.wait_event(events[b])
            with torch.cuda.stream(comm_stream):
                if timing is None:
                    self._pending.append(dist.all_reduce(self.grads[b], async_op=True))
                else:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    w = dist.all_reduce(self.grads[b], async_op=True)
                    w.wait()                                      # (the communication stream waits for the collective's stream)
                    e1.record()
                    timing.append((b, e0, e1))

    def wait(self):
        for w in self._pending:
            w.wait()
        self._pending = []

    def apply(self, events=None, opt_stream=None):
        """One optimiser step on every bucket: lr from the staircase schedule at the current global step, gradients averaged
        over the ranks (the all-reduce summed them).
        events + opt_stream (one rank): bucket b is updated on `opt_stream` behind events[b] - which the native step records once
        the bucket's last gradient AND every reader of its variables in this step's backward have been enqueued (a variable is read
        by the backward of its own layer only, and the milestones are flushed at block ends: train_model.hip ms_flush) - last bucket
        first, i.e. under the rest of the still running backward instead of behind it; the caller's stream joins `opt_stream`.
        The update is elementwise: the result does not depend on where it runs."""
        import torch
        import torch.distributed as dist
        from . import _lib
        from ._lib import check
        self.wait()
        world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        lr = learning_rate(self.step, self.lr, self.lr_iters, self.lr_decay)
        self.step += 1
        lr_t = adam_lr_t(self.step, lr)
        early = events is not None and opt_stream is not None
        on = opt_stream if early else torch.cuda.current_stream()
        stream = C.c_void_p(on.cuda_stream) if self.params and self.params[0].is_cuda else None
        order = reversed(range(len(self.params))) if early else range(len(self.params))
        for b in order:
            p, g, m, v = self.params[b], self.grads[b], self.m[b], self.v[b]
            if not p.is_cuda:
                raise RuntimeError('AdamBuckets.apply needs device buckets: the optimiser kernel has no CPU implementation')
            if early:
                opt_stream.wait_event(events[b])
            check(_lib.lib().sagen_adam_update(C.c_void_p(p.data_ptr()), C.c_void_p(g.data_ptr()), C.c_void_p(m.data_ptr()),
                                               C.c_void_p(v.data_ptr()), p.numel(), lr_t, ADAM_BETA1, ADAM_BETA2, ADAM_EPSILON,
                                               1.0 / world, stream))
        if early:
            torch.cuda.current_stream().wait_stream(opt_stream)
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

    def __init__(self, net, batch, lr=1e-4, lr_iters=10000, lr_decay=1.0, bucket_bytes=None, variables=None, overlap=None, early_adam=None):
        """overlap: start each gradient bucket's all-reduce as soon as the backward has produced it (None: when there is more than
        one rank; SAGEN_NO_OVERLAP=1 forces the exchange after the backward).  early_adam (round 6, one rank; None: only with
        SAGEN_EARLY_ADAM=1): update each bucket on a second stream as soon as the backward is done with it (AdamBuckets.apply)
        instead of three launches (145 us of a 6.1 ms step) behind it.  OPT-IN, a recorded negative result: same box, alternating,
        548 - 550 ambisonic-s/s trained without it, 546 - 549 with it - the updates take from the backward's kernels the bandwidth
        they no longer use at the end (tools/ab_train_env.sh, tools/train_timeline.sh).
        bucket_bytes (None): 64 MiB on one rank (three Adam launches); 16 MiB with several ranks or early_adam - ten buckets, of
        which nine (108 of 123 MB) are complete 2.2 ms before the backward ends (tools/bucket_events.py; the first bucket holds the
        stem and completes last whatever its size)."""
        import torch
        import torch.distributed as dist
        multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        if early_adam is None:
            early_adam = not multi and bool(os.environ.get('SAGEN_EARLY_ADAM'))
        self.early_adam = bool(early_adam) and not multi
        if bucket_bytes is None:
            bucket_bytes = (16 << 20) if (multi or self.early_adam) else (64 << 20)
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
        if overlap is None:
            overlap = multi and not os.environ.get('SAGEN_NO_OVERLAP')
        self.bucket_events, self.comm_stream = None, None
        if overlap or self.early_adam:
            self._enable_overlap()

    def _enable_overlap(self, timing=False):
        """One event per gradient bucket, recorded by the native step when the bucket is complete; the exchange runs on its own stream."""
        import torch
        from . import _lib
        from ._lib import check
        nb = len(self.opt.grads)
        self.bucket_events = [torch.cuda.Event(enable_timing=timing) for _ in range(nb)]
        for e in self.bucket_events:
            e.record(torch.cuda.current_stream(self.device))       # (creates the underlying hipEvent_t)
        self.comm_stream = torch.cuda.Stream(device=self.device)
        names = list(self.opt.layout)
        self._k3 = [k.encode() for k in names]
        arr_n = (C.c_char_p * len(names))(*self._k3)
        arr_b = (C.c_int32 * len(names))(*[self.opt.layout[k][0] for k in names])
        arr_e = (C.c_void_p * nb)(*[e.cuda_event for e in self.bucket_events])
        check(_lib.lib().sagen_train_set_grad_events(self.ctx.handle, arr_n, arr_b, len(names), arr_e, nb))

    def _prep(self, t, tail, keep_u8=False):
        import torch
        if t is None:
            return None
        t = torch.as_tensor(np.asarray(t)) if not isinstance(t, torch.Tensor) else t
        if keep_u8 and t.dtype == torch.uint8:          # frames as decoded: normalised on the device (sagen_train_step_u8)
            t = t.to(device=self.device).contiguous()
        else:
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
        v = self._prep(video, (1, 224, 448, 3), keep_u8=True) if VIDEO in self.net.encoders else None
        f = self._prep(flow, (1, 224, 448, 3)) if FLOW in self.net.encoders else None
        t = self._prep(target, (4800, 3))
        mk = channel_mask(mask, self.batch, self.device)
        p = lambda x: C.c_void_p(x.data_ptr()) if x is not None else None
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        self.last_entry = 'sagen_train_step_u8' if (v is not None and v.dtype == torch.uint8) else 'sagen_train_step'
        step = getattr(_lib.lib(), self.last_entry)
        check(step(self.ctx.handle, p(a), p(v), p(f), p(t), p(mk), p(self.pred), p(self.loss), int(update_moving), stream))
        return self.loss

    def step(self, audio, video, flow, target, mask=None, comm_timing=None):
        """One training iteration (train.py:208): returns (loss tensor on device, learning rate used).
        comm_timing (dict, measurement only, needs the overlapped exchange): filled with the milliseconds the communication stream
        spent waiting for gradient buckets and inside all-reduces during this step (synchronises at the end)."""
        import torch
        timing = None
        if comm_timing is not None and self.bucket_events is not None:
            timing = []
            t_begin = torch.cuda.Event(enable_timing=True)
            t_begin.record(torch.cuda.current_stream(self.device))
            self.comm_stream.wait_event(t_begin)
        loss = self.forward_backward(audio, video, flow, target, mask)
        if self.early_adam and self.bucket_events is not None:       # one rank: nothing to exchange, the optimiser follows the milestones
            return loss, self.opt.apply(self.bucket_events, self.comm_stream)
        if self.bucket_events is not None:
            self.opt.all_reduce_after(self.bucket_events, self.comm_stream, timing)      # under the rest of the (still running) backward
            if timing is not None:                                # the optimiser (current stream) consumes what the exchange produced
                torch.cuda.current_stream(self.device).wait_stream(self.comm_stream)
        else:
            self.opt.all_reduce()
        lr = self.opt.apply()
        if timing:
            t_end = torch.cuda.Event(enable_timing=True)
            t_end.record(torch.cuda.current_stream(self.device))
            torch.cuda.synchronize(self.device)
            wait_ms, xchg_ms, prev, rows = 0.0, 0.0, t_begin, []
            for b, e0, e1 in timing:
                w, x = max(prev.elapsed_time(e0), 0.0), e0.elapsed_time(e1)
                rows.append({'bucket': b, 'mbytes': round(self.opt.grads[b].numel() * 4 / 1e6, 1), 'ready_ms': round(t_begin.elapsed_time(e0), 3),
                             'wait_ms': round(w, 3), 'all_reduce_ms': round(x, 3)})
                wait_ms += w
                xchg_ms += x
                prev = e1
            comm_timing.update(step_ms=round(t_begin.elapsed_time(t_end), 3), waiting_for_gradients_ms=round(wait_ms, 3),
                               in_all_reduce_ms=round(xchg_ms, 3), last_all_reduce_done_ms=round(t_begin.elapsed_time(timing[-1][2]), 3),
                               buckets=rows)
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

    # ---- launch plan / per-launch profile of the training context (same native mechanism as SptAudioGen.autotune) ----
    def autotune(self, audio, video, flow, target, mask=None):
        """Time every (tile, split-K) candidate of every contraction of the step - forward and data gradients - on these inputs
        and keep the fastest.  The gradients this call leaves behind are not meaningful.  Returns the plan."""
        import torch
        from . import _lib
        from ._lib import check
        from .definitions import VIDEO, FLOW
        a = self._prep(audio, (52799, 1))
        if video is not None:
            video = video if isinstance(video, torch.Tensor) else torch.as_tensor(np.asarray(video))
            if video.dtype == torch.uint8:                  # (the tuning step takes float frames: x / 255 - 0.5, whatever the container)
                video = (video.to(torch.float64) / 255.0 - 0.5).to(torch.float32)
        v = self._prep(video, (1, 224, 448, 3)) if VIDEO in self.net.encoders else None
        f = self._prep(flow, (1, 224, 448, 3)) if FLOW in self.net.encoders else None
        t = self._prep(target, (4800, 3))
        mk = channel_mask(mask, self.batch, self.device)
        p = lambda x: C.c_void_p(x.data_ptr()) if x is not None else None
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        check(_lib.lib().sagen_train_autotune(self.ctx.handle, p(a), p(v), p(f), p(t), p(mk), stream))
        return self.plan()

    def plan(self):
        from . import _lib
        from ._lib import check
        buf = C.create_string_buffer(1 << 17)
        n = _lib.lib().sagen_plan_describe(self.ctx.handle, buf, len(buf))
        if n < 0:
            check(n)
        rows = []
        for line in buf.value.decode().splitlines():
            layer, tile, sk, us = line.split('\t')
            rows.append((layer, tile, int(sk), float(us)))
        return rows

    def plan_set(self, layer, tile, splitk):
        from . import _lib
        from ._lib import check
        check(_lib.lib().sagen_plan_set(self.ctx.handle, layer.encode(), max(int(tile), 0), int(splitk)))

    def load_plan_rows(self, rows):
        from .model import SptAudioGen
        names = SptAudioGen.tile_names()
        for layer, tile, sk, _ in rows:
            if layer.endswith('#materialize'):
                self.plan_set(layer, 0, sk)
            else:
                self.plan_set(layer, names.index(tile) if tile in names else 0, max(sk, 1))

    def profile_enable(self, on=True):
        from . import _lib
        from ._lib import check
        check(_lib.lib().sagen_profile_enable(self.ctx.handle, int(on)))

    def profile_report(self):
        """[(kernel, layer, microseconds, flops)] for every launch of the last step (forward + backward)."""
        from . import _lib
        from ._lib import check
        buf = C.create_string_buffer(1 << 19)
        n = _lib.lib().sagen_profile_report(self.ctx.handle, buf, len(buf))
        if n < 0:
            check(n)
        rows = []
        for line in buf.value.decode().splitlines():
            k, layer, us, fl = line.split('\t')
            rows.append((k, layer, float(us), float(fl)))
        return rows

    # ---- checkpoints (tf.train.Saver of train.py:176, 223-225, 234; restored by --resume, train.py:194-200) ----
    def state_dict(self):
        """Everything tf.train.Saver would write: variables, BN moving averages, Adam slots (`<var>/Adam`, `<var>/Adam_1`: the
        `with tf.variable_scope('optimization') and tf.control_dependencies(..)` of train.py:148 enters only the second context, so
        the slots carry no scope prefix), `beta1_power`, `beta2_power` and the global `step`."""
        out = OrderedDict()
        for k in self.opt.layout:
            out[k] = self.opt.view('params', k).cpu().numpy()
            out[k + '/Adam'] = self.opt.view('m', k).cpu().numpy()
            out[k + '/Adam_1'] = self.opt.view('v', k).cpu().numpy()
        for k, v in self.moving.items():
            out[k] = v.cpu().numpy()
        out['step'] = np.asarray(self.opt.step, np.int32)
        out['beta1_power'] = np.asarray(ADAM_BETA1 ** (self.opt.step + 1), np.float32)     # TF stores beta^t for the NEXT application
        out['beta2_power'] = np.asarray(ADAM_BETA2 ** (self.opt.step + 1), np.float32)
        return out

    def load_state_dict(self, state, strict_slots=False, log=None):
        """Restore variables, Adam slots, moving averages and the step.  A checkpoint without the optimiser slots (a deploy /
        weights-only bundle) cannot continue the Adam trajectory: the step is then reset to 0 with zero moments - a FRESH optimiser on
        the restored weights, whose bias corrections are consistent - and a warning says so (strict_slots=True raises instead)."""
        import warnings
        import torch
        missing_slots = []
        for k in self.opt.layout:
            for which, sfx in (('params', ''), ('m', '/Adam'), ('v', '/Adam_1')):
                dst = self.opt.view(which, k)
                if k + sfx in state:
                    src = np.asarray(state[k + sfx], np.float32)
                    if tuple(src.shape) != tuple(dst.shape):
                        raise ValueError('checkpoint entry %s has shape %s, the model expects %s' % (k + sfx, tuple(src.shape), tuple(dst.shape)))
                    dst.copy_(torch.as_tensor(src))
                elif not sfx:
                    raise KeyError('variable %s missing from the checkpoint' % k)
                else:
                    missing_slots.append(k + sfx)
        for k in self.moving:
            if k in state:
                src = np.asarray(state[k], np.float32)
                if tuple(src.shape) != tuple(self.moving[k].shape):
                    raise ValueError('checkpoint entry %s has shape %s, the model expects %s' % (k, tuple(src.shape), tuple(self.moving[k].shape)))
                self.moving[k].copy_(torch.as_tensor(src))
        step = int(np.asarray(state.get('step', 0)))
        if missing_slots:
            msg = ('%d Adam slots are missing from the checkpoint (first: %s): the optimiser restarts at step 0 with zero moments '
                   'on the restored variables' % (len(missing_slots), missing_slots[0]))
            if strict_slots:
                raise KeyError(msg)
            warnings.warn(msg)
            if log:
                log('WARNING: ' + msg)
            for k in self.opt.layout:
                self.opt.view('m', k).zero_()
                self.opt.view('v', k).zero_()
            step = 0
        elif 'beta1_power' in state:
            # TF keeps beta^(t+1) next to the slots; a bundle whose step and powers disagree was assembled from two runs
            b1 = float(np.asarray(state['beta1_power']))
            want = ADAM_BETA1 ** (step + 1)
            if abs(b1 - want) > 1e-3 * max(want, 1e-30) and want > 1e-30:
                warnings.warn('beta1_power %.6g of the checkpoint does not match step %d (expected %.6g)' % (b1, step, want))
        self.opt.step = step

    def save(self, model_dir, global_step=None):
        from .checkpoint import save_checkpoint
        prefix = os.path.join(model_dir, 'model.ckpt' + ('-%d' % global_step if global_step is not None else ''))
        save_checkpoint(prefix, self.state_dict())
        return prefix

    def restore(self, model_dir):
        """tf.train.latest_checkpoint + saver.restore (train.py:196-200); returns the restored global step (0 if none)."""
        from .checkpoint import latest_checkpoint, load_checkpoint
        ckpt = latest_checkpoint(model_dir)
        if not ckpt:
            return 0
        self.load_state_dict(load_checkpoint(ckpt))
        return self.opt.step


import os  # noqa: E402


def save_params(args, model_dir):
    """myutils.save_params (myutils.py:28-31): `key: value` lines, read back by deploy.load_params / myutils.load_params."""
    with open(os.path.join(model_dir, 'train-params.txt'), 'w') as f:
        for k, v in sorted(vars(args).items()):
            f.write('{}: {}\n'.format(k, v))


def synthetic_batches(encoders, batch, seed=0, pool=2):
    """Endless synthetic batches of SURVEY 8(d): a small pool of distinct batches, cycled.  The target is the crop of the mono
    context scaled per channel (a learnable fixed mixing)."""
    from .weights import synth_inputs
    items = []
    for i in range(pool):
        inp = synth_inputs(batch, encoders, seed=seed + i)
        tgt = (inp['audio'][:, 24000:28800, :] * np.array([0.5, 0.25, -0.5], np.float32)).astype(np.float32)
        items.append((inp['audio'], inp.get('video'), inp.get('flow'), tgt, np.ones((batch, 4), np.float32)))
    i = 0
    while True:
        yield items[i % pool]
        i += 1


def silence_threshold(subset_fn, db_dir=None):
    """feeder.py:310: `0.01 if 'REC-Street' in self.subset_fn else 0.2` - the SUBSET FILE NAME decides (README: `--subset_fn
    meta/subsets/REC-Street.train.1.lst`); a match in the dataset folder name is accepted as well."""
    return 0.01 if ('REC-Street' in str(subset_fn or '') or 'REC-Street' in str(db_dir or '')) else 0.2


def folder_batches(db_dir, ids, params, batch, layouts=None, seed=0, samples_per_clip=5, subset_fn=None):
    """The training feeder (feeder.py:372-407 with for_eval=False): clips in shuffled order for ever, per clip a SampleReader with
    shuffled windows, silence skipping and random rotations about z, `samples_per_clip` (NUM_SAMPLING = 5) windows of each; the
    sample stream is cut into batches.  Yields (audio [B,n,1], video, flow, target [B,4800,3], channel mask [B,4])."""
    import random
    from .feeder import SampleReader
    from .definitions import VIDEO, FLOW
    rnd = random.Random(seed)
    thr = silence_threshold(subset_fn if subset_fn is not None else getattr(params, 'subset_fn', None), db_dir)
    ss, t = int(params.audio_rate * params.context) // 2, int(params.audio_rate * 0.1)
    buf = []
    while True:
        order = list(ids)
        rnd.shuffle(order)
        for yid in order:
            reader = SampleReader(os.path.join(db_dir, yid), ambi_order=params.ambi_order, audio_rate=params.audio_rate,
                                  video_rate=params.video_rate, context=params.context, duration=0.1,
                                  return_video=VIDEO in params.encoders, img_prep=None, return_flow=FLOW in params.encoders,     # video frames stay uint8: Trainer dispatches to sagen_train_step_u8
                                  skip_silence_thr=thr, shuffle=True, random_rotations=True)
            for smp in reader.loop_chunks(samples_per_clip):
                smp['mask'] = np.asarray((layouts or {}).get(yid, np.ones(4)), np.float32)
                buf.append(smp)
                if len(buf) == batch:
                    amb = np.stack([s_['ambix'] for s_ in buf], 0).astype(np.float32)
                    def st(k):
                        if k not in buf[0]:
                            return None
                        x = np.stack([s_[k] for s_ in buf], 0)
                        return x if (k == 'video' and x.dtype == np.uint8) else x.astype(np.float32)    # decoded frames as they are
                    yield (amb[:, :, :1], st('video'), st('flow'), np.ascontiguousarray(amb[:, ss:ss + t, 1:]),
                           np.stack([s_['mask'] for s_ in buf], 0))       # train.py:107-111: input W, target Y,Z,X of the centre window
                    buf = []


def train_loop(tr, batches, model_dir, n_iters, init_step=0, log_every=20, ckpt_every=5000, log=print):
    """The loop of train.py:202-234: one optimiser step per iteration; every `log_every` steps the loss is read back and a NaN
    aborts the run (train.py:212-213); a checkpoint every `ckpt_every` steps (train.py:223-225) and one at exit (train.py:234),
    also when the loop dies.  Returns the list of logged (step, loss, lr)."""
    import math
    import time
    import torch
    import torch.distributed as dist
    multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
    history = []
    t_last, n_last = time.time(), init_step
    step = init_step
    # tf.train.Saver(max_to_keep=1), train.py:176: the Saver only ever deletes bundles IT saved in this session (saver.restore does
    # not register the restored one), so nothing this run did not write is touched - hand-kept milestones, another experiment's
    # bundles and the bundle --resume restored all stay (the directory is never globbed).
    last_periodic = None
    flag = None
    sat_seen = tr.ctx.counter('fp16x2_saturations') if hasattr(tr, 'ctx') else 0
    try:
        for step in range(init_step, n_iters):
            audio, video, flow, target, mask = next(batches)
            loss, lr = tr.step(audio, video, flow, target, mask)
            if step % log_every == 0:
                # fp16x2 guard: elements the activation / gradient plane passes had to clamp since the last look (expected 0; a non-finite
                # value counts and also poisons the planes with NaN, h2_planes.h) - read at the log step's synchronisation
                sat_now = tr.ctx.counter('fp16x2_saturations') if hasattr(tr, 'ctx') else 0
                sat_new, sat_seen = sat_now - sat_seen, sat_now
                if multi:
                    # every rank learns about a NaN on ANY rank in the same step (sum of [loss, isnan]) and raises with the others - a
                    # rank that left alone would leave the rest hanging in the next bucket all-reduce; the logged loss is the global mean
                    flag = torch.stack([loss.detach().double().reshape(()), torch.isnan(loss.detach()).double().reshape(()),
                                        torch.tensor(float(sat_new), dtype=torch.float64, device=loss.device)])
                    dist.all_reduce(flag)
                    lv = float(flag[0]) / dist.get_world_size()
                    bad = float(flag[1]) > 0 or math.isnan(lv)
                    sat_new = int(flag[2])
                else:
                    lv = float(loss)                                        # the only host synchronisation of the loop
                    bad = math.isnan(lv)
                if bad:
                    raise ValueError('Training produced a NaN metric or loss.')
                if sat_new > 0:
                    raise FloatingPointError('fp16x2 planes clamped %d elements since step %d (statistical range bound exceeded): '
                                             'restart with SAGEN_TRAIN_NO_H2=1 SAGEN_TRAIN_NO_H2D=1 (bf16 planes)' % (sat_new, max(step - log_every, init_step)))
                now = time.time()
                rate = tr.batch * max(step - n_last, 1) / max(now - t_last, 1e-9)
                t_last, n_last = now, step
                history.append((step, lv, lr))
                if rank == 0:
                    log('TRAIN | step %d | stft/mse %.6g | lr %.3g | %.1f samples/s per GPU' % (step, lv, lr, rate))
            if step % ckpt_every == 0 and step != 0 and rank == 0:
                prefix = tr.save(model_dir, global_step=tr.opt.step)
                from .checkpoint import remove_checkpoint
                if last_periodic and last_periodic != prefix:               # max_to_keep=1 over this run's own periodic saves
                    remove_checkpoint(last_periodic)
                last_periodic = prefix
                log('=' * 60 + '\nCheckpoint saved\n' + '=' * 60)
    finally:
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        if rank == 0:
            log('End of training.\nSaving model.')
            tr.save(model_dir)
    return history


def parse_arguments(argv=None):
    """The CLI of train.py:15-59 (same option names and defaults), plus --synthetic for runs without a dataset."""
    import argparse
    from .definitions import ENCODERS, SEPARATION, FREQ_MASK, NUM_SEP_TRACKS_DEF, SEP_FFT_WINDOW_DEF, CTX_FEATS_FCUNITS_DEF, \
        SEP_FREQ_MASK_FCUNITS_DEF, LOC_FCUNITS_DEF
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument('db_dir', help='Directory containing db ("synthetic" with --synthetic).')
    ap.add_argument('model_dir', help='Directory to store model.')
    ap.add_argument('--subset_fn', default='')
    ap.add_argument('--encoders', nargs='*', type=str.lower, choices=ENCODERS, default=['audio', 'flow', 'video'])
    ap.add_argument('--separation', type=str.lower, default=FREQ_MASK, choices=SEPARATION)
    ap.add_argument('--ambi_order', type=int, default=1)
    ap.add_argument('--audio_rate', type=int, default=48000)
    ap.add_argument('--video_rate', type=int, default=10)
    ap.add_argument('--context', type=float, default=1.0)
    ap.add_argument('--sample_dur', type=float, default=0.1)
    ap.add_argument('--n_iters', type=int, default=1000000)
    ap.add_argument('--lr', type=float, default=1e-4)
    ap.add_argument('--lr_decay', type=float, default=0.5)
    ap.add_argument('--lr_iters', type=int, default=250000)
    ap.add_argument('--batch_size', type=int, default=32)
    ap.add_argument('--resume', action='store_true')
    ap.add_argument('--num_sep_tracks', default=NUM_SEP_TRACKS_DEF, type=int)
    ap.add_argument('--fft_window', default=SEP_FFT_WINDOW_DEF, type=float)
    ap.add_argument('--context_units', default=CTX_FEATS_FCUNITS_DEF, nargs='+', type=int)
    ap.add_argument('--freq_mask_units', default=SEP_FREQ_MASK_FCUNITS_DEF, nargs='*', type=int)
    ap.add_argument('--loc_units', default=LOC_FCUNITS_DEF, nargs='+', type=int)
    ap.add_argument('--gpu', type=int, default=0)
    ap.add_argument('--pretrained', default=None, metavar='resnet18.npy',
                    help="ImageNet initialisation of the ResNet18 trunks: the reference's pyutils/tflib/models/image/resnet18.npy (a pickled "
                         '{variable name without scope: array}; resnet.py:238-249, run at train.py:182-184).  Assigned after the initialisers, '
                         'before a --resume restore (which then overwrites it, as in the reference)')
    ap.add_argument('--synthetic', action='store_true', help='synthetic inputs / targets instead of a dataset folder')
    ap.add_argument('--seed', type=int, default=0)
    args = ap.parse_args(argv)
    if len(args.subset_fn) == 0:
        args.subset_fn = None
    if args.resume and not os.path.isfile(os.path.join(args.model_dir, 'train-params.txt')):
        args.resume = False
    return args


def main(argv=None):
    """train.py:62-236 on the HIP path.  One process per GPU (torchrun): every rank holds a replica, draws its own batches and the
    gradient buckets are summed over the ranks (RCCL) before the fused Adam step."""
    import torch
    from .deploy import load_params
    from .dist import init_process_group
    from .model import SptAudioGen, SptAudioGenParams
    from .weights import init_weights
    from .evaluate import read_layouts
    args = parse_arguments(argv)
    rank, world = init_process_group()
    if torch.cuda.is_available():
        torch.cuda.set_device((int(os.environ.get('LOCAL_RANK', args.gpu))) % torch.cuda.device_count())
    os.makedirs(args.model_dir, exist_ok=True)
    if args.resume:
        prm = load_params(args.model_dir)
        for k in ('encoders', 'separation', 'ambi_order', 'audio_rate', 'video_rate', 'context', 'sample_dur'):
            setattr(args, k, getattr(prm, k))
    elif rank == 0:
        save_params(args, args.model_dir)
    args.encoders = sorted(args.encoders)
    args.video_rate = int(1. / min(args.context, args.sample_dur, 1. / args.video_rate))      # train.py:83-84
    net = SptAudioGen(args.ambi_order, audio_rate=args.audio_rate, video_rate=args.video_rate, context=args.context,
                      sample_duration=args.sample_dur, encoders=list(args.encoders), separation=args.separation,
                      params=SptAudioGenParams(sep_num_tracks=args.num_sep_tracks, ctx_feats_fc_units=args.context_units,
                                               loc_fc_units=args.loc_units, sep_freq_mask_fc_units=args.freq_mask_units,
                                               sep_fft_window=args.fft_window))
    # initialisers of the reference (core.py:13,34; fc3 ~ N(0, 0.001^2), model.py:255); the same replica on every rank
    P = init_weights(net.variable_specs(), seed=args.seed, mode='bench', fc3_std=0.001)
    if args.pretrained:                  # sess.run(rest_ops), train.py:182-184: both trunks start from the ImageNet ResNet18
        from .weights import load_pretrained_resnet18
        names = load_pretrained_resnet18(P, args.pretrained, specs=net.variable_specs())
        if rank == 0:
            print('Initialised %d trunk variables from %s' % (len(names), args.pretrained))
    tr = Trainer(net, batch=args.batch_size, lr=args.lr, lr_iters=args.lr_iters, lr_decay=args.lr_decay, variables=P)
    init_step = tr.restore(args.model_dir) if args.resume else 0
    if args.synthetic:
        batches = synthetic_batches(args.encoders, args.batch_size, seed=1234 + rank)
    else:
        ids = [l.strip() for l in open(args.subset_fn)] if args.subset_fn else sorted(os.listdir(args.db_dir))
        ids = [i for i in ids if i and os.path.isdir(os.path.join(args.db_dir, i))]
        layouts = read_layouts(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'meta', 'audio_layouts.txt'))
        from .feeder import BatchPrefetcher
        batches = iter(BatchPrefetcher(folder_batches(args.db_dir, ids, args, args.batch_size, layouts, seed=args.seed + rank, subset_fn=args.subset_fn), depth=4))
    train_loop(tr, batches, args.model_dir, args.n_iters, init_step)


if __name__ == '__main__':
    main()
