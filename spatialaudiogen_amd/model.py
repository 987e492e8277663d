"""SptAudioGen facade — same constructor and `inference_ops` signature as the reference class
(model.py:24-60, 356-434), executing on libsagen_hip.so instead of building a TF1 graph.

    net = SptAudioGen(ambi_order=1, encoders=['audio', 'video'], separation='unet_mask',
                      params=SptAudioGenParams(sep_num_tracks=32, loc_fc_units=[512, 512]))
    net.load_variables(P)                      # {TF checkpoint key: tensor/ndarray}  (deploy.py:79-87)
    x_ambi = net.inference_ops(audio, video)   # audio [B,52799,1], video [B,1,224,448,3] -> [B,4800,3]
"""
import ctypes as C
from collections import OrderedDict
import numpy as np
import torch

from . import _lib
from ._lib import check, SagenConfig, SagenTensor
from .definitions import *   # noqa: F401,F403  (AUDIO/VIDEO/FLOW, separation modes, defaults)
from .geometry import Geometry


class SptAudioGenParams(object):
    """reference model.py:10-21 (ctx_feats_fc_units / sep_freq_mask_fc_units are carried but, as in
    the reference graph, unused by inference_ops)."""

    def __init__(self, sep_num_tracks=NUM_SEP_TRACKS_DEF, ctx_feats_fc_units=CTX_FEATS_FCUNITS_DEF,
                 loc_fc_units=LOC_FCUNITS_DEF, sep_freq_mask_fc_units=SEP_FREQ_MASK_FCUNITS_DEF,
                 sep_fft_window=SEP_FFT_WINDOW_DEF):
        self.sep_num_tracks = sep_num_tracks
        self.ctx_feats_fc_units = ctx_feats_fc_units
        self.loc_fc_units = loc_fc_units
        self.sep_freq_mask_fc_units = sep_freq_mask_fc_units
        self.sep_fft_window = sep_fft_window


class _Ctx(object):
    """One native context = one batch size (the TF graph is also built for a fixed batch)."""

    def __init__(self, cfg, variables, device, groups=1):
        l = _lib.lib()
        self.handle = C.c_void_p()
        self.groups = int(groups)
        if self.groups > 1:             # `groups` independent batches per forward, one launch per layer (include/sagen.h: grouped launch)
            check(l.sagen_create_grouped(C.byref(self.handle), C.byref(cfg), self.groups))
        else:
            check(l.sagen_create(C.byref(self.handle), C.byref(cfg)))
        self.batch = cfg.batch
        nbytes = int(l.sagen_workspace_bytes(self.handle))
        self.workspace = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=device)
        n = l.sagen_num_variables(self.handle)
        arr = (SagenTensor * n)()
        keep = []
        name, ndim, shape = C.c_char_p(), C.c_int32(), (C.c_int64 * 4)()
        for i in range(n):
            check(l.sagen_variable_spec(self.handle, i, C.byref(name), C.byref(ndim), shape))
            key = name.value.decode()
            if key not in variables:
                if '/moving_' in key:      # never read: BN always runs in training mode (model.py:197)
                    arr[i].name, arr[i].data, arr[i].ndim = name.value, None, ndim.value
                    continue
                raise KeyError('variable %s missing from the checkpoint dict' % key)
            t = variables[key]
            want = tuple(shape[k] for k in range(ndim.value))
            if tuple(t.shape) != want:
                raise ValueError('variable %s has shape %s, expected %s' % (key, tuple(t.shape), want))
            keep.append(name.value)
            arr[i].name = name.value
            arr[i].data = t.data_ptr()
            arr[i].ndim = ndim.value
            for k in range(4):
                arr[i].shape[k] = shape[k]
        valid = [a for a in arr if a.data]
        arr2 = (SagenTensor * len(valid))(*valid)
        stream = C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
        check(l.sagen_bind_weights(self.handle, arr2, len(valid), C.c_void_p(self.workspace.data_ptr()),
                                   self.workspace.numel() * 4, stream))
        self._keep = keep

    def set_option(self, name, value):
        """Per-context switch of the native library (sagen_set_option), e.g. 'materialize_mask'."""
        check(_lib.lib().sagen_set_option(self.handle, name.encode(), int(value)))

    def counter(self, name):
        """Diagnostic counter of the native context (sagen_counter), e.g. 'fp16x2_saturations'; synchronises the current stream."""
        v = C.c_uint64(0)
        check(_lib.lib().sagen_counter(self.handle, name.encode(), C.byref(v), C.c_void_p(torch.cuda.current_stream(self.workspace.device).cuda_stream)))
        return int(v.value)

    def intermediate(self, name):
        l = _lib.lib()
        data, ndim, shape, ps = C.c_void_p(), C.c_int32(), (C.c_int64 * 4)(), C.c_int64()
        check(l.sagen_get_intermediate(self.handle, name.encode(), C.byref(data), C.byref(ndim), shape, C.byref(ps)))
        dims = [shape[k] for k in range(ndim.value)]
        off = (data.value - self.workspace.data_ptr()) // 4
        # strided view into the workspace (concat buffers hold two tensors side by side)
        if ndim.value == 4:
            st = [dims[1] * dims[2] * ps.value, dims[2] * ps.value, ps.value, 1]
        elif ndim.value == 3:
            st = [dims[1] * ps.value, ps.value, 1]
        else:
            raise ValueError(ndim.value)
        return torch.as_strided(self.workspace, dims, st, off).clone()

    def __del__(self):
        try:
            if self.handle:
                _lib.lib().sagen_destroy(self.handle)
        except Exception:
            pass


class SptAudioGen(object):
    def __init__(self, ambi_order, audio_rate=48000, video_rate=10, context=1., sample_duration=0.1,
                 encoders=None, separation='none', params=None, device=None, groups=1):
        if params is None:              # defaults: 32 separated tracks; the mono-only decoder has exactly one (deploy.py:62-63 passes 1)
            params = SptAudioGenParams(sep_num_tracks=1) if separation == NO_SEPARATION else SptAudioGenParams()
        if separation == NO_SEPARATION and params.sep_num_tracks != 1:
            raise ValueError("separation 'none' needs params.sep_num_tracks = 1 (got %r): the reference sizes fc3 with sep_num_tracks + 1 "
                             "(model.py:254) against the single mono track of model.py:274-280" % (params.sep_num_tracks,))
        self.geom = Geometry(audio_rate=audio_rate, video_rate=video_rate, context=context,
                             sample_duration=sample_duration, ambi_order=ambi_order,
                             fft_window=params.sep_fft_window)        # asserts of model.py:33,42
        self.ambi_order = ambi_order
        self.num_ambi_channels = sum([2 * i + 1 for i in range(ambi_order + 1)])
        self.snd_rate, self.vid_rate = audio_rate, video_rate
        self.context, self.duration = context, sample_duration
        self.snd_contx, self.snd_dur, self.snd_size = self.geom.snd_contx, self.geom.snd_dur, self.geom.snd_size
        if encoders is None:
            encoders = [AUDIO, VIDEO, FLOW]
        assert isinstance(encoders, list)
        assert all([e in ENCODERS for e in encoders])
        if separation not in SEPARATION:
            raise ValueError('Unknown separation mode.')                 # model.py:350
        self.encoders = encoders
        self.separation = separation
        self.params = params
        self.wind_size = self.geom.wind_size
        self.device = torch.device(device) if device is not None else torch.device('cuda', torch.cuda.current_device()) \
            if torch.cuda.is_available() else None
        self._variables = None
        self._ctx = {}
        # groups > 1: every forward carries `groups` INDEPENDENT batches (inputs [groups * B, ...], the batches back to back), run as
        # one launch per layer by a grouped native context (include/sagen.h: sagen_create_grouped) - each batch keeps its own
        # batch-norm statistics and its output is bit-identical to a forward of that batch alone; `batch` arguments of the plan /
        # profile / option methods stay the size of ONE batch
        self.groups = int(groups)
        if self.groups < 1:
            raise ValueError('groups must be >= 1')

    # ---- weights (tf.train.Saver.restore, deploy.py:79-87) ---------------------------------
    def variable_specs(self):
        from .weights import variable_specs
        return variable_specs(self.encoders, self.separation, self.params.sep_num_tracks,
                              tuple(self.params.loc_fc_units), self.geom)

    def load_variables(self, variables):
        if self.device is None:
            raise RuntimeError('SptAudioGen needs a ROCm device: there is no CPU implementation of the path')
        specs = self.variable_specs()
        dev = OrderedDict()
        for k, shape in specs.items():
            if k not in variables:
                if '/moving_' in k:
                    continue
                raise KeyError('variable %s missing' % k)
            t = torch.as_tensor(np.asarray(variables[k]) if not isinstance(variables[k], torch.Tensor) else variables[k])
            if tuple(t.shape) != tuple(shape):
                raise ValueError('variable %s has shape %s, expected %s' % (k, tuple(t.shape), tuple(shape)))
            dev[k] = t.to(device=self.device, dtype=torch.float32).contiguous()
        self._variables = dev
        self._ctx = {}

    def _config(self, batch):
        cfg = SagenConfig()
        cfg.batch = batch
        cfg.encoders = sum({AUDIO: 1, VIDEO: 2, FLOW: 4}[e] for e in set(self.encoders))
        cfg.separation = {NO_SEPARATION: 0, FREQ_MASK: 1}[self.separation]
        cfg.num_sep_tracks = self.params.sep_num_tracks
        cfg.n_loc_units = len(self.params.loc_fc_units)
        for i, u in enumerate(self.params.loc_fc_units):
            cfg.loc_units[i] = u
        cfg.ambi_order = self.ambi_order
        cfg.audio_rate, cfg.video_rate = self.snd_rate, self.vid_rate
        cfg.context, cfg.sample_duration = self.context, self.duration
        cfg.fft_window = self.params.sep_fft_window
        return cfg

    def context_for(self, batch):
        if self._variables is None:
            raise RuntimeError('load_variables() first')
        if batch not in self._ctx:
            self._ctx[batch] = _Ctx(self._config(batch), self._variables, self.device, groups=self.groups)
        return self._ctx[batch]

    # ---- the hot path ------------------------------------------------------------------------
    def inference_ops(self, audio, video=None, flow=None, is_training=True, out=None):
        """audio [B,snd_size,1]; video/flow [B,1,224,448,3] -> x_ambi [B,snd_dur,3] (Y,Z,X).
        `is_training` is accepted for signature parity; as in the reference graph it does not change
        the arithmetic (BN uses batch statistics either way, model.py:197)."""
        def prep(t, name, tail):
            if t is None:
                return None
            if not isinstance(t, torch.Tensor):
                t = torch.as_tensor(np.asarray(t))
            t = t.to(device=self.device, dtype=torch.float32).contiguous()
            if tuple(t.shape[1:]) != tail:
                raise ValueError('%s has shape %s, expected [B,%s]' % (name, tuple(t.shape), ','.join(map(str, tail))))
            return t
        audio = prep(audio, 'audio', (self.snd_size, 1))
        if isinstance(video, np.ndarray) and video.dtype == np.uint8:
            video = torch.as_tensor(video)
        video_u8 = isinstance(video, torch.Tensor) and video.dtype == torch.uint8 and VIDEO in self.encoders
        if video_u8:          # decoded frames as they are: the x/255 - 0.5 of the feeder (myutils.py:88-89) runs on the device
            if tuple(video.shape[1:]) != (1, 224, 448, 3):
                raise ValueError('video has shape %s, expected [B,1,224,448,3]' % (tuple(video.shape),))
            video = video.to(device=self.device).contiguous()
        else:
            video = prep(video, 'video', (1, 224, 448, 3)) if VIDEO in self.encoders else None
        flow = prep(flow, 'flow', (1, 224, 448, 3)) if FLOW in self.encoders else None
        if VIDEO in self.encoders and video is None:
            raise ValueError('video encoder enabled but no video given')
        if FLOW in self.encoders and flow is None:
            raise ValueError('flow encoder enabled but no flow given')
        N = audio.shape[0]
        for t, nm in ((video, 'video'), (flow, 'flow')):
            if t is not None and t.shape[0] != N:
                raise ValueError('%s holds %d windows, audio %d' % (nm, t.shape[0], N))
        if N % self.groups:
            raise ValueError('%d windows do not make %d groups of equal batches' % (N, self.groups))
        B = N // self.groups
        ctx = self.context_for(B)
        if out is None:
            out = torch.empty(N, self.snd_dur, self.geom.num_out, dtype=torch.float32, device=self.device)
        elif tuple(out.shape) != (N, self.snd_dur, self.geom.num_out) or not out.is_contiguous():
            raise ValueError('out must be a contiguous [%d, %d, %d] tensor' % (N, self.snd_dur, self.geom.num_out))
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        L = _lib.lib()
        if self.groups > 1:
            fwd = L.sagen_forward_grouped_u8 if video_u8 else L.sagen_forward_grouped
            check(fwd(ctx.handle, self.groups, p(audio), p(video), p(flow), p(out), stream))
        else:
            fwd = L.sagen_forward_u8 if video_u8 else L.sagen_forward
            check(fwd(ctx.handle, p(audio), p(video), p(flow), p(out), stream))
        return out

    def inference_ops_checked(self, audio, video=None, flow=None, on_saturation='rerun', out=None):
        """inference_ops + the guard of the fp16x2 trunk arithmetic (what deploy.py / evaluate.py call).

        The ResNet trunks' activation planes are scaled per layer from batch-norm STATISTICS (|beta_c| + 8 sigma_c, fp16 range 64 x
        beyond that: csrc/conv3h.hip); an element beyond that range is clamped and counted on the device.  With weights nobody has
        seen a heavy-tailed channel could get there, so the hosts look: after the forward the counter is read (4 bytes D2H; the
        callers synchronise per batch anyway) and, if this batch clamped anything, the batch is run again with the trunk on three
        bf16 planes (fp32's exponent range, nothing to saturate; `on_saturation='rerun'`) or the call raises (`'raise'`).
        `self.saturation_events` = [(batch size, clamped elements)] of every batch that was re-run."""
        y = self.inference_ops(audio, video, flow, out=out)
        B = y.shape[0] // self.groups
        ctx = self.context_for(B)
        if not hasattr(self, 'saturation_events'):
            self.saturation_events = []
        # the last-seen value lives ON the context: the native counter is the context's, and load_variables() replaces the contexts
        # (a count kept per batch size on this object would go stale against a fresh context's zero: ADVICE r05)
        now = ctx.counter('fp16x2_saturations')
        clamped = now - getattr(ctx, 'sat_seen', 0)
        ctx.sat_seen = now
        if clamped <= 0:
            return y
        if on_saturation == 'raise' or self.groups > 1:          # (a grouped context has no bf16-plane path to re-run on)
            raise FloatingPointError('fp16x2 activation planes saturated (%d elements clamped in this batch): run with '
                                     "set_option(batch, 'fp16x2', 0) / SAGEN_NO_H2=1" % clamped)
        import warnings
        warnings.warn('fp16x2 activation planes clamped %d elements of this batch: re-running it on three bf16 planes' % clamped)
        self.saturation_events.append((B, clamped))
        ctx.set_option('fp16x2', 0)
        try:
            y = self.inference_ops(audio, video, flow, out=y)
        finally:
            ctx.set_option('fp16x2', 1)
        return y

    # ---- evaluation metrics (model.py:110-154) -----------------------------------------------------
    def evaluation_ps(self, preds_t, targets_t):
        """Device-only part of evaluation_ops: (ps [4, B, 3] = per-sample stft distance, lsd, mse, snr; pw [2] fp64 power sums),
        no host synchronisation (the eval loop keeps batches in flight and reduces once at the end)."""
        pr = torch.as_tensor(preds_t).to(device=self.device, dtype=torch.float32).contiguous()
        gt = torch.as_tensor(targets_t).to(device=self.device, dtype=torch.float32).contiguous()
        B = pr.shape[0]
        if tuple(pr.shape) != (B, self.snd_dur, 3) or tuple(gt.shape) != tuple(pr.shape):
            raise ValueError('predictions / targets must be [B, %d, 3]' % self.snd_dur)
        l = _lib.lib()
        key = ('eval', B)
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        if key not in self._ctx:
            nbytes = int(l.sagen_eval_scratch_bytes(B))
            scratch = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=self.device)
            check(l.sagen_eval_init(C.c_void_p(scratch.data_ptr()), scratch.numel() * 4, B, stream))
            self._ctx[key] = scratch
        scratch = self._ctx[key]
        ps = torch.empty(4, B, 3, dtype=torch.float32, device=self.device)
        pw = torch.zeros(2, dtype=torch.float64, device=self.device)
        check(l.sagen_eval_metrics(C.c_void_p(pr.data_ptr()), C.c_void_p(gt.data_ptr()), B, C.c_void_p(ps.data_ptr()),
                                   C.c_void_p(pw.data_ptr()), C.c_void_p(scratch.data_ptr()), scratch.numel() * 4, stream))
        return ps, pw

    def evaluation_ops(self, preds_t, targets_t, w_t=None, mask_channels=None):
        """preds/targets [B, snd_dur, 3] -> (metrics OrderedDict, stft_dist_ps, lsd_ps, mse_ps, snr_ps), the last four
        [B, 3] device tensors exactly as the reference returns them.  `w_t` is accepted for signature parity (unused
        by the reference too).  Per-sample values are computed by libsagen_hip.so; the masked channel means of
        model.py:119-150 are a handful of scalar operations on [B,3] done here."""
        ps, pw = self.evaluation_ps(preds_t, targets_t)
        B = ps.shape[1]
        mask = torch.ones(B, 3, device=self.device) if mask_channels is None else \
            torch.as_tensor(mask_channels).to(device=self.device, dtype=torch.float32)
        num_masked = torch.clamp(mask.sum(0), min=1.0)
        metrics = OrderedDict()
        for name, idx, scale in (('stft', 0, 100.), ('lsd', 1, 1.), ('mse', 2, 5e3), ('snr', 3, 1.)):
            v = (ps[idx].double() * mask.double()).sum(0) / num_masked.double() * scale
            metrics[name + '/avg'] = float(v.mean())
            for i, ch in enumerate('YZX'):
                metrics[name + '/' + ch] = float(v[i])
        metrics['pow/pred'] = float(pw[0]) / (3.0 * B)
        metrics['pow/gt'] = float(pw[1]) / (3.0 * B)
        return metrics, ps[0], ps[1], ps[2], ps[3]

    def intermediate(self, batch, name):
        return self.context_for(batch).intermediate(name)

    def counter(self, batch, name):
        """'fp16x2_saturations': activation elements the fp16x2 plane passes had to clamp since the weights were bound (expected: 0)."""
        return self.context_for(batch).counter(name)

    def set_option(self, batch, name, value):
        """'materialize_mask' = 1: the next forwards keep the mask logits ('separation/deconv1') instead of folding the mask into the
        last deconvolution's epilogue."""
        self.context_for(batch).set_option(name, value)

    # ---- per-layer launch plan -------------------------------------------------------------------
    def autotune(self, audio, video=None, flow=None):
        """Time every (tile, split-K) candidate of every contraction on these inputs and keep the fastest
        (stored per batch size).  Returns the plan as [(layer, tile, splitk, microseconds)]."""
        out = self.inference_ops(audio, video, flow)            # validates / stages the inputs, creates the ctx
        B = out.shape[0] // self.groups
        ctx = self.context_for(B)
        def to_dev(t, tail):
            if t is None:
                return None
            t = torch.as_tensor(np.asarray(t) if not isinstance(t, torch.Tensor) else t).to(device=self.device)
            if t.dtype == torch.uint8:                          # decoded frames: the tuner times the general kernels on the feeder's float frames
                t = (t.double() / 255.0 - 0.5)
            return t.to(dtype=torch.float32).contiguous()
        a, v, f = to_dev(audio, None), to_dev(video if VIDEO in self.encoders else None, None), to_dev(flow if FLOW in self.encoders else None, None)
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        check(_lib.lib().sagen_autotune(ctx.handle, p(a), p(v), p(f), p(out), stream))
        return self.plan(B)

    def plan_set(self, batch, layer, tile, splitk):
        check(_lib.lib().sagen_plan_set(self.context_for(batch).handle, layer.encode(), max(int(tile), 0), int(splitk)))

    def save_plan(self, batch, fn):
        """Persist the launch plan of this batch size (JSON) so later processes can replay it without tuning."""
        import json
        with open(fn, 'w') as f:
            json.dump({'batch': batch, 'encoders': self.encoders, 'plan': self.plan(batch)}, f, indent=1)

    def load_plan(self, batch, fn):
        import json
        from ._lib import SIGNATURES  # noqa: F401
        with open(fn) as f:
            doc = json.load(f)
        if doc.get('batch') != batch or doc.get('encoders') != self.encoders:
            raise ValueError('%s was tuned for batch %s / encoders %s' % (fn, doc.get('batch'), doc.get('encoders')))
        names = self.tile_names()
        for layer, tile, sk, _ in doc['plan']:
            self.plan_set(batch, layer, names.index(tile) if tile in names else 0, sk)

    @staticmethod
    def tile_names():
        """Names of the contraction-kernel instantiations, index = the `tile` argument of plan_set (sagen_tile_name)."""
        L = _lib.lib()
        return [L.sagen_tile_name(i).decode() for i in range(L.sagen_num_tiles())]

    def plan(self, batch):
        buf = C.create_string_buffer(1 << 16)
        n = _lib.lib().sagen_plan_describe(self.context_for(batch).handle, buf, len(buf))
        if n < 0:
            check(n)
        rows = []
        for line in buf.value.decode().splitlines():
            layer, tile, sk, us = line.split('\t')
            rows.append((layer, tile, int(sk), float(us)))
        return rows

    # ---- measurement aid: per-launch HIP-event timing inside the native runtime ---------------
    def profile_enable(self, batch, on=True):
        check(_lib.lib().sagen_profile_enable(self.context_for(batch).handle, int(on)))

    def profile_report(self, batch):
        """[(kernel, layer, microseconds, flops)] for every launch of the last forward."""
        buf = C.create_string_buffer(1 << 18)
        n = _lib.lib().sagen_profile_report(self.context_for(batch).handle, buf, len(buf))
        if n < 0:
            check(n)
        rows = []
        for line in buf.value.decode().splitlines():
            k, layer, us, fl = line.split('\t')
            rows.append((k, layer, float(us), float(fl)))
        return rows
