"""CPU ORACLE — TEST INFRASTRUCTURE ONLY (not product code).

A numpy restatement of the reference's STFT -> encoders -> U-Net mask decoder -> iSTFT ->
ambisonic-mix inference path, op for op, NHWC, in float64 (ground truth) or float32.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

PARITY UNPINNED: the arithmetic of the reference lives in tensorflow-gpu==1.4.0rc1
(requirements.txt:10), which is absent here and cannot be installed; the reference has no
golden vectors or tests for this path.  This file restates the cited
reference lines plus TF's documented op semantics; it is pinned only by analytic
known-answer tests and by an independent second implementation (oracle/torch_ref.py).

All `file:line` citations are relative to /root/reference.
"""
from collections import OrderedDict
import numpy as np

AUDIO, VIDEO, FLOW = 'audio', 'video', 'flow'
NO_SEPARATION, FREQ_MASK = 'none', 'unet_mask'
BN_EPS = 1e-3      # tf.contrib.layers.batch_norm default epsilon (core.py:6,210 passes none)


# ----------------------------------------------------------------------------------------
# TF op semantics (third-party: tensorflow 1.4; restated from its documented behaviour)
# ----------------------------------------------------------------------------------------
def same_pad(n, k, s):
    """TF 'SAME': out = ceil(n/s); pad_total = max((out-1)*s + k - n, 0); before = total//2."""
    out = -(-n // s)
    tot = max((out - 1) * s + k - n, 0)
    return out, tot // 2, tot - tot // 2


def _windows(x, kh, kw, sh, sw):
    """[B,H,W,C] -> strided view [B,Ho,Wo,kh,kw,C] (VALID)."""
    b, h, w, c = x.shape
    ho, wo = (h - kh) // sh + 1, (w - kw) // sw + 1
    st = x.strides
    return np.lib.stride_tricks.as_strided(
        x, (b, ho, wo, kh, kw, c), (st[0], st[1] * sh, st[2] * sw, st[1], st[2], st[3]),
        writeable=False)


def nn_convolution(x, w, stride=(1, 1), padding='SAME'):
    """tf.nn.convolution, NHWC x HWIO, cross-correlation (core.py:206)."""
    kh, kw, cin, cout = w.shape
    sh, sw = stride
    if padding == 'SAME':
        _, pt, pb = same_pad(x.shape[1], kh, sh)
        _, pl, pr = same_pad(x.shape[2], kw, sw)
        x = np.pad(x, ((0, 0), (pt, pb), (pl, pr), (0, 0)))
    else:
        assert padding == 'VALID'
    x = np.ascontiguousarray(x)
    win = _windows(x, kh, kw, sh, sw)
    return np.tensordot(win, w, axes=([3, 4, 5], [0, 1, 2]))


def nn_conv2d_transpose(x, w, stride, padding='VALID'):
    """tf.nn.conv2d_transpose = gradient of conv2d wrt its input (core.py:140).
    w is [kh,kw,Cout,Cin]; out[b, i*sh+p, j*sw+q, o] += x[b,i,j,c] * w[p,q,o,c]; VALID only
    (the only mode on the path, model.py:304)."""
    assert padding == 'VALID'
    kh, kw, cout, cin = w.shape
    sh, sw = stride
    b, h, wd, _ = x.shape
    out = np.zeros((b, h * sh + kh - sh, wd * sw + kw - sw, cout), dtype=x.dtype)   # core.py:139
    for p in range(kh):
        for q in range(kw):
            out[:, p:p + (h - 1) * sh + 1:sh, q:q + (wd - 1) * sw + 1:sw, :] += x @ w[p, q].T
    return out


def batch_norm_train(x, gamma, beta, eps=BN_EPS):
    """contrib batch_norm with is_training=True: batch mean / biased variance over N,H,W."""
    ax = tuple(range(x.ndim - 1))
    mu = x.mean(axis=ax)
    var = x.var(axis=ax)
    return (x - mu) / np.sqrt(var + eps) * gamma + beta


def max_pool_3x3_s2_same(x):
    """tf.nn.max_pool(x,[1,3,3,1],[1,2,2,1],'SAME') (resnet.py:135): -inf padding."""
    _, pt, pb = same_pad(x.shape[1], 3, 2)
    _, pl, pr = same_pad(x.shape[2], 3, 2)
    xp = np.pad(x, ((0, 0), (pt, pb), (pl, pr), (0, 0)), constant_values=-np.inf)
    return _windows(np.ascontiguousarray(xp), 3, 3, 2, 2).max(axis=(3, 4))


def relu(x):
    return np.maximum(x, 0)


def sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


# ----------------------------------------------------------------------------------------
# pyutils/tflib/wrappers/core.py
# ----------------------------------------------------------------------------------------
def conv_2d(x, P, name, stride, padding='SAME', use_bias=True, use_batch_norm=False, act=relu):
    """tfw.conv_2d (core.py:156-220): conv, then BN *or* bias, then activation."""
    stride = (stride, stride) if np.isscalar(stride) else tuple(stride)
    x = nn_convolution(x, P[name + '/weights'], stride, padding)
    if use_batch_norm:
        x = batch_norm_train(x, P[name + '/bn/gamma'], P[name + '/bn/beta'])     # core.py:209-210
    elif use_bias:
        x = x + P[name + '/biases']                                              # core.py:213-214
    return act(x) if act is not None else x


def deconv_2d(x, P, name, stride, act=None):
    """tfw.deconv_2d (core.py:96-153): conv2d_transpose VALID + bias (+ activation)."""
    x = nn_conv2d_transpose(x, P[name + '/weights'], tuple(stride), 'VALID')
    x = x + P[name + '/biases']
    return act(x) if act is not None else x


def fully_connected(x, P, name, act=relu):
    """tfw.fully_connected (core.py:43-93): reshape [-1,in], matmul, bias, act, reshape back."""
    shp = x.shape
    y = x.reshape(-1, shp[-1]) @ P[name + '/weights'] + P[name + '/biases']
    if act is not None:
        y = act(y)
    return y.reshape(shp[:-1] + (-1,))


# ----------------------------------------------------------------------------------------
# myutils.py
# ----------------------------------------------------------------------------------------
def stft(inp, wind_size, n_overlap, fdt=np.float64):
    """myutils.stft (myutils.py:119-147), written the way the reference writes it."""
    inp_sz = list(inp.shape)
    if len(inp_sz) > 2:
        inp = inp.reshape(int(np.prod(inp_sz[:-1])), inp_sz[-1])
    batch_size, n_frames = inp.shape
    n_winds = int(np.floor(n_frames / wind_size)) - 1
    x_crops = []
    for ss in range(0, wind_size, wind_size // n_overlap):
        x_crops.append(inp[:, ss:ss + wind_size * n_winds])
    x = np.stack(x_crops, 1).reshape(batch_size, n_overlap, -1, wind_size)
    hann = (0.5 - 0.5 * np.cos(2 * np.pi / wind_size * np.arange(wind_size))).astype(np.float32)  # :134
    x = x * hann.astype(fdt)[None, None]
    s = np.fft.fft(x.astype(np.complex128 if fdt == np.float64 else np.complex64), axis=-1)
    s = s.transpose(0, 2, 1, 3)
    sz = s.shape
    s = s.reshape(sz[0], sz[1] * sz[2], sz[3])
    if len(inp_sz) > 2:
        s = s.reshape(inp_sz[:-1] + list(s.shape[-2:]))
    return s


def istft(inp, n_overlap):
    """myutils.istft (myutils.py:181-211): plain average of the overlaps, no synthesis window."""
    inp_sz = list(inp.shape)
    if len(inp_sz) > 3:
        inp = inp.reshape(int(np.prod(inp_sz[:-2])), inp_sz[-2], inp_sz[-1])
    batch_size, n_frames, n_freqs = inp.shape
    n_frames = int(int(float(n_frames) / n_overlap) * n_overlap)
    inp = inp[:, :n_frames, :]
    x = np.real(np.fft.ifft(inp, axis=-1))
    x = x.reshape(batch_size, -1, n_overlap, n_freqs).transpose(0, 2, 1, 3)
    x = x.reshape(batch_size, n_overlap, -1)
    skip = n_freqs // n_overlap
    acc = None
    for i in range(n_overlap):
        xi = x[:, i]
        xi = xi[:, (n_overlap - i - 1) * skip:] if i == 0 else xi[:, (n_overlap - i - 1) * skip:-i * skip]
        acc = xi if acc is None else acc + xi
    x = acc / float(n_overlap)
    if len(inp_sz) > 3:
        x = x.reshape(inp_sz[:-2] + [x.shape[-1]])
    return x


# ----------------------------------------------------------------------------------------
# pyutils/tflib/models/image/resnet.py
# ----------------------------------------------------------------------------------------
def resnet18_conv5_2(x, P, scope, ends=None):
    """ResNet18.inference_ops(x, is_training=True, truncate_at='conv5_2') (resnet.py:123-190).
    BN runs with batch statistics: model.py:197 passes finetune=True into is_training."""
    def block(x, name):                                                           # resnet.py:224-236
        sc = x
        x = conv_2d(x, P, name + '/conv_1', 1, use_bias=False, use_batch_norm=True, act=relu)
        x = conv_2d(x, P, name + '/conv_2', 1, use_bias=False, use_batch_norm=True, act=None)
        return relu(x + sc)

    def block_first(x, name, stride):                                             # resnet.py:200-222
        sc = conv_2d(x, P, name + '/shortcut', stride, use_bias=False, act=None)
        x = conv_2d(x, P, name + '/conv_1', stride, use_bias=False, use_batch_norm=True, act=relu)
        x = conv_2d(x, P, name + '/conv_2', 1, use_bias=False, use_batch_norm=True, act=None)
        return relu(x + sc)

    x = conv_2d(x, P, scope + '/conv1/conv', 2, use_bias=False, use_batch_norm=True, act=relu)
    if ends is not None:
        ends[scope + '/conv1'] = x
    x = max_pool_3x3_s2_same(x)
    if ends is not None:
        ends[scope + '/pool1'] = x
    for stage in (2, 3, 4, 5):
        for unit in (1, 2):
            name = '%s/conv%d_%d' % (scope, stage, unit)
            x = block_first(x, name, 2) if (unit == 1 and stage > 2) else block(x, name)
            if ends is not None:
                ends[name] = x
    return x


# ----------------------------------------------------------------------------------------
# model.py
# ----------------------------------------------------------------------------------------
class SptAudioGenOracle(object):
    """SptAudioGen (model.py:24-434), inference only."""

    def __init__(self, ambi_order=1, audio_rate=48000, video_rate=10, context=1., sample_duration=0.1,
                 encoders=None, separation=FREQ_MASK, sep_num_tracks=32, loc_fc_units=(512, 512),
                 sep_fft_window=0.025, dtype=np.float64):
        assert float(audio_rate) / video_rate == int(audio_rate) // int(video_rate)
        self.ambi_order = ambi_order
        self.snd_rate, self.vid_rate = audio_rate, video_rate
        self.snd_contx = int(context * audio_rate)
        self.snd_dur = int(sample_duration * audio_rate)
        self.snd_size = self.snd_contx + self.snd_dur - 1
        self.encoders = [AUDIO, VIDEO, FLOW] if encoders is None else list(encoders)
        self.separation = separation
        self.sep_num_tracks = sep_num_tracks
        self.loc_fc_units = list(loc_fc_units)
        self.wind_size = int(sep_fft_window * self.snd_rate)
        self.wind_size = int(2 ** np.round(np.log2(self.wind_size)))              # model.py:59-60
        self.dtype = dtype
        self.ends = OrderedDict()

    # model.py:161-187
    def audio_encoder_ops(self, stft_c, P):
        n_filters = [32, 64, 128, 256, 512]
        filter_size = [(7, 16), (3, 7), (3, 5), (3, 5), (3, 5)]
        stride = [(4, 8), (2, 4), (2, 2), (1, 1), (1, 1)]
        inp_dim = 95.
        ss = (self.snd_contx / 2.) * (4. / self.wind_size)
        ss = int(ss - (inp_dim - 1) / 2.)
        tt = (self.snd_contx / 2. + self.snd_dur) * (4. / self.wind_size)
        tt = int(tt + (inp_dim - 1) / 2.)
        tt = int((np.ceil((tt - ss - inp_dim) / 16.)) * 16 + inp_dim + ss)
        s = stft_c[:, :, ss:tt, :].transpose(0, 2, 3, 1)
        x = np.abs(s).astype(self.dtype)
        self.ends['audio_encoder/mag'] = x
        down = [x]
        for l, (nf, fs, st) in enumerate(zip(n_filters, filter_size, stride)):
            name = 'audio_encoder/conv%d' % (l + 1)
            x = conv_2d(x, P, name, st, padding='VALID', act=relu)
            self.ends[name] = x
            down.append(x)
        return down

    # model.py:189-201
    def visual_encoding_ops(self, inp, P, scope):
        shp = inp.shape
        x = inp.reshape((shp[0] * shp[1],) + shp[2:])
        return resnet18_conv5_2(x, P, scope, self.ends)

    # model.py:203-239
    def bottleneck_ops(self, x_enc, P):
        bottleneck = []
        audio_sz = x_enc[AUDIO][-1].shape
        for k in [AUDIO, VIDEO, FLOW]:
            if k in x_enc:
                x = x_enc[k][-1] if k == AUDIO else x_enc[k]
                if k != AUDIO:
                    x = fully_connected(x, P, 'bottleneck/%s-fc-red' % k, act=relu)
                sz = x.shape
                out_shape = (sz[0], sz[1], sz[2] * sz[3]) if k == AUDIO else (sz[0], 1, sz[1] * sz[2] * sz[3])
                x = x.reshape(out_shape)
                x = fully_connected(x, P, 'bottleneck/%s-fc' % k, act=relu)
                self.ends['bottleneck/%s-fc' % k] = x
                if k in [VIDEO, FLOW]:
                    x = np.tile(x, (1, audio_sz[1], 1))
                bottleneck.append(x)
        return np.concatenate(bottleneck, 2)

    # model.py:241-271
    def localization_ops(self, x, P):
        num_out = (self.ambi_order + 1) ** 2 - self.ambi_order ** 2
        num_in = self.ambi_order ** 2
        for i, u in enumerate(self.loc_fc_units):
            x = fully_connected(x, P, 'localization/fc%d' % (i + 1), act=relu)
        nsep = self.sep_num_tracks if self.separation != NO_SEPARATION else 1
        x = fully_connected(x, P, 'localization/fc%d' % (len(self.loc_fc_units) + 1), act=None)
        sz = x.shape
        x = x.reshape(sz[0], sz[1], num_out, num_in, nsep + 1)
        self.ends['localization/coeffs'] = x
        sz = x.shape
        x = np.tile(x[:, :, None], (1, 1, self.snd_dur // sz[1], 1, 1, 1))
        x = x.reshape(sz[0], self.snd_dur, sz[2], sz[3], sz[4])
        return x[..., :-1], x[..., -1]

    # model.py:273-354
    def separation_ops(self, mono, stft_c, audio_enc, feats, P):
        if self.separation == NO_SEPARATION:
            ss = self.snd_contx // 2
            return mono[:, :, ss:ss + self.snd_dur][:, None]
        n_filters = [32, 64, 128, 256, 512]
        filter_size = [(7, 16), (3, 7), (3, 5), (3, 5), (3, 5)]
        stride = [(4, 8), (2, 4), (2, 2), (1, 1), (1, 1)]
        feats = fully_connected(feats, P, 'separation/fc-feats', act=relu)
        self.ends['separation/fc-feats'] = feats
        enc_sz = audio_enc[-1].shape
        feats = np.tile(feats[:, :, None], (1, 1, enc_sz[2], 1))
        x = np.concatenate([audio_enc[-1], feats], axis=3)
        n_chann_in = mono.shape[1]
        nfs = [self.sep_num_tracks * n_chann_in] + n_filters[:-1]
        for l in reversed(range(5)):
            name = 'separation/deconv%d' % (l + 1)
            x = deconv_2d(x, P, name, stride[l], act=None)
            self.ends[name] = x
            if l == 0:
                break
            x = np.concatenate((relu(x), audio_enc[:-1][l]), 3)
        ss = np.floor((self.snd_contx / 2. - self.wind_size) * (4. / self.wind_size))
        tt = np.ceil((self.snd_contx / 2. + self.snd_dur + self.wind_size) * (4. / self.wind_size))
        inp_dim = 95.
        skip = (self.snd_contx / 2.) * (4. / self.wind_size)
        skip = int(skip - (inp_dim - 1) / 2.)
        stft_c = stft_c[:, :, int(ss):int(tt)]
        x = x[:, int(ss - skip):int(tt - skip), :]
        x = x.transpose(0, 3, 1, 2)
        x_sz = x.shape
        x = x.reshape(x_sz[0], n_chann_in, -1, x_sz[2], x_sz[3])
        f_mask = sigmoid(x)
        self.ends['separation/mask'] = f_mask
        stft_sep = stft_c[:, :, None] * f_mask
        x_sep = istft(stft_sep, 4)
        ss = self.snd_contx / 2.
        skip = np.floor((self.snd_contx / 2. - self.wind_size) * (4. / self.wind_size)) * (self.wind_size / 4.)
        skip += 3. * self.wind_size / 4.
        x_sep = x_sep[:, :, :, int(ss - skip):int(ss - skip) + self.snd_dur]
        return x_sep

    # model.py:356-434
    def inference_ops(self, audio, P, video=None, flow=None):
        dt = self.dtype
        P = {k: v.astype(dt) for k, v in P.items()}
        audio = np.asarray(audio, dtype=dt).transpose(0, 2, 1)
        stft_c = stft(audio, self.wind_size, 4, fdt=dt)
        self.ends['stft'] = stft_c
        x_enc = {}
        if AUDIO in self.encoders:
            x_enc[AUDIO] = self.audio_encoder_ops(stft_c, P)
        if VIDEO in self.encoders:
            x_enc[VIDEO] = self.visual_encoding_ops(np.asarray(video, dtype=dt), P, 'video_encoder')
        if FLOW in self.encoders:
            x_enc[FLOW] = self.visual_encoding_ops(np.asarray(flow, dtype=dt), P, 'flow_encoder')
        feats = self.bottleneck_ops(x_enc, P)
        self.ends['bottleneck'] = feats
        weights, biases = self.localization_ops(feats, P)
        x_sep = self.separation_ops(audio, stft_c, x_enc[AUDIO], feats, P)
        self.ends['separation/all_channels'] = x_sep
        x_sep = x_sep.transpose(0, 3, 1, 2)
        x_ambi = (weights * x_sep[:, :, None]).sum(axis=4).sum(axis=3) + biases[:, :, :, 0]   # model.py:430
        self.ends['decoder/ambix'] = x_ambi
        return x_ambi


# ----------------------------------------------------------------------------------------
# deploy.py / feeder.py window arithmetic (row 13) — host float64 quirks reproduced verbatim
# ----------------------------------------------------------------------------------------
def audio_pow_times(n_audio_files):
    """Times listed in audio_pow.lst (scraping/preprocess.py:146-153): t = i/10.+0.5 for
    i < (duration-1)*10, where duration = number of 1-s wav chunks; written with '{}'.format(t)
    (Python 2: 12 significant digits) and parsed back with float() (feeder.py:217)."""
    return [float('%.12g' % (i / 10. + 0.5)) for i in range((int(n_audio_files) - 1) * 10)]


def deploy_window_table(chunks_t, deploy_start=0., deploy_duration=10., audio_rate=48000,
                        video_rate=10, context=1.0, batch_size=10, num_audio_frames=None):
    """(t, start_frame, pad_before, frame_idx, batch_id, read_start) per window, following
    feeder.py:228-231 (filters), deploy.py:106-107 (time shift), feeder.py:64-72 (audio start /
    padding), feeder.py:79-82 (first sample actually read), feeder.py:121 (frame index),
    deploy.py:112-139 (groups of 10)."""
    ts = list(chunks_t)
    if deploy_start > 0.5:
        ts = [t for t in ts if t >= deploy_start]
    if deploy_duration is not None:
        ts = [t for t in ts if t < deploy_start + deploy_duration]
    dt = ts[0] - deploy_start
    ts = [t - dt for t in ts]
    rows = []
    for i, t in enumerate(ts):
        start_time = t - context / 2
        start_frame = int(start_time * audio_rate)
        pad_before = abs(start_frame) if start_frame < 0 else 0
        st = 0. if start_frame < 0 else start_time
        read_start = int(st) * int(audio_rate) + int((st - int(st)) * audio_rate)          # feeder.py:79-82
        frame_idx = max(int(t * video_rate), 0)
        rows.append((t, start_frame, pad_before, frame_idx, i // batch_size, read_start))
    return rows


# ----------------------------------------------------------------------------------------
# pyutils/ambisonics: first-order projection decode + RMS power map (row 14)
# ----------------------------------------------------------------------------------------
def spherical_mesh(angular_res):
    """distance.py:9-13."""
    phi_rg = np.flip(np.arange(-180., 180., angular_res) / 180. * np.pi, 0)
    nu_rg = np.arange(-90., 90.1, angular_res) / 180. * np.pi
    return np.meshgrid(phi_rg, nu_rg)


def sh_matrix_order1(phi, nu):
    """spherical_harmonics_matrix for order 1, ACN/SN3D (common.py:121-178):
    Y = [1, cos(nu) sin(phi), sin(nu), cos(nu) cos(phi)]  (W, Y, Z, X)."""
    phi, nu = np.asarray(phi, np.float64).reshape(-1), np.asarray(nu, np.float64).reshape(-1)
    return np.stack([np.ones_like(phi), np.cos(nu) * np.sin(phi), np.sin(nu), np.cos(nu) * np.cos(phi)], 1)


def power_map(ambi, angular_res):
    """AmbiDecoder.decode 'projection' (decoder.py:24-26) + RMS + flipud (distance.py:41-52)."""
    phi_mesh, nu_mesh = spherical_mesh(angular_res)
    Y = sh_matrix_order1(phi_mesh, nu_mesh)
    decoded = np.asarray(ambi, np.float64) @ Y.T
    rms = np.sqrt(np.mean(decoded ** 2, 0)).reshape(phi_mesh.shape)
    return np.flipud(rms)


# ----------------------------------------------------------------------------------------
# evaluation metrics on the graph (model.py:62-154, myutils.stft_for_loss myutils.py:151-178)
# ----------------------------------------------------------------------------------------
def stft_for_loss(signal, window, n_overlap):
    """myutils.stft_for_loss (myutils.py:151-178). signal [BS, N, nC] -> complex [BS, nC, nW, window2]."""
    BS, N, nC = signal.shape
    window = int(2 ** np.ceil(np.log(window) / np.log(2)))
    hann_window = (0.5 - (0.5 * np.cos(2 * np.pi / window * np.arange(window)))).astype(np.float32).astype(np.float64)
    if n_overlap == 1:
        nW = int(float(N) / window)
        if nW > 1:
            if N > window * nW:
                signal = signal[:, :window * nW, :]
            windows = signal.reshape(BS, nW, window, nC)
        else:
            windows = signal
    else:
        windows = []
        stride = int(window / n_overlap)
        for i in range(n_overlap):
            nW = int(float(N - i * stride - 1) / window)
            y = signal[:, (i * stride):(i * stride) + window * nW, :]
            windows.append(y.reshape(BS, nW, window, nC))
        windows = np.concatenate(windows, 1)
    windows = windows.transpose(0, 3, 1, 2) * hann_window[None, None, None, :]
    return np.fft.fft(windows, axis=-1)


def evaluation_ops(preds, targets, mask_channels=None, snd_rate=48000, fft_window=0.025, fft_overlap=2):
    """SptAudioGen.evaluation_ops (model.py:110-154): preds/targets [B, N, 3]; mask [B, 3].
    Returns (metrics dict, stft_dist_ps, lsd_ps, mse_ps, snr_ps)."""
    preds, targets = np.asarray(preds, np.float64), np.asarray(targets, np.float64)
    B, _, nC = preds.shape
    if mask_channels is None:
        mask_channels = np.ones((B, nC))
    mask_channels = np.asarray(mask_channels, np.float64)
    num_masked = np.maximum(mask_channels.sum(axis=0), 1)
    window = int(fft_window * snd_rate)
    metrics = OrderedDict()
    # _stft_mse_ops (model.py:62-76)
    diff = np.abs(stft_for_loss(targets, window, fft_overlap) - stft_for_loss(preds, window, fft_overlap))
    stft_ps = (diff ** 2).mean(axis=3).mean(axis=2)
    stft_dist = (stft_ps * mask_channels).sum(axis=0) / num_masked * 100.
    metrics['stft/avg'] = stft_dist.mean()
    for i, ch in enumerate('YZX'):
        metrics['stft/' + ch] = stft_dist[i]
    # _lsd_ops (model.py:78-94)
    EPS = 1e-2
    s_gt = stft(targets.transpose(0, 2, 1), window, fft_overlap)
    s_pr = stft(preds.transpose(0, 2, 1), window, fft_overlap)
    ps = lambda x: 10 * np.log(np.abs(x) + EPS) / np.log(10.)
    lsd_ps = np.sqrt(((ps(s_gt) - ps(s_pr)) ** 2).mean(axis=3)).mean(axis=2)
    lsd = (lsd_ps * mask_channels).sum(axis=0) / num_masked
    metrics['lsd/avg'] = lsd.mean()
    for i, ch in enumerate('YZX'):
        metrics['lsd/' + ch] = lsd[i]
    # _temporal_mse_ops / _temporal_snr_ops (model.py:96-108)
    mse_ps = ((targets - preds) ** 2).mean(axis=1)
    mse = (mse_ps * mask_channels).sum(axis=0) / num_masked * 5e3
    metrics['mse/avg'] = mse.mean()
    for i, ch in enumerate('YZX'):
        metrics['mse/' + ch] = mse[i]
    snr_ps = 10. * np.log(((targets ** 2).sum(axis=1) + 1e-1) / (((targets - preds) ** 2).sum(axis=1) + 1e-1)) / np.log(10.)
    snr = (snr_ps * mask_channels).sum(axis=0) / num_masked
    metrics['snr/avg'] = snr.mean()
    for i, ch in enumerate('YZX'):
        metrics['snr/' + ch] = snr[i]
    metrics['pow/pred'] = (preds ** 2).mean(axis=2).mean(axis=0).sum()
    metrics['pow/gt'] = (targets ** 2).mean(axis=2).mean(axis=0).sum()
    return metrics, stft_ps, lsd_ps, mse_ps, snr_ps


# ----------------------------------------------------------------------------------------
# training loss and optimiser (model.py:156-159, myutils.py:214-222) - checker for spatialaudiogen_amd/train.py
# ----------------------------------------------------------------------------------------
def stft_loss(pred, gt, mask_channels=None):
    """losses['stft/mse'] = metrics['stft/avg'] (model.py:156-159): the FFT form of model.py:62-76,122-127."""
    return evaluation_ops(pred, gt, mask_channels)[0]['stft/avg']


def stft_loss_grad_fd(pred, gt, mask_channels, probes, h=1e-3):
    """Central finite differences of stft_loss at the (b, n, c) index triples `probes` (fp64)."""
    out = []
    for (b, n, c) in probes:
        p1, p2 = pred.copy(), pred.copy()
        p1[b, n, c] += h
        p2[b, n, c] -= h
        out.append((stft_loss(p1, gt, mask_channels) - stft_loss(p2, gt, mask_channels)) / (2 * h))
    return np.array(out)


def adam_tf(p, g, m, v, t, lr, beta1=0.9, beta2=0.999, eps=1e-8):
    """tf.train.AdamOptimizer, dense update, as TF 1.4 documents it (the 'epsilon hat' form):
    lr_t = lr sqrt(1 - beta2^t) / (1 - beta1^t);  m, v moments;  p -= lr_t m / (sqrt(v) + eps).  Returns (p, m, v)."""
    lr_t = lr * np.sqrt(1 - beta2 ** t) / (1 - beta1 ** t)
    # the variables are float32, so TF casts beta1 / beta2 to float32 and forms (1 - beta) in float32 (training_ops:
    # ApplyAdam on float): 1 - 0.999f = 9.9998713e-4, 1.3e-5 off the real 1e-3.  The coefficients are restated with that
    # rounding; the accumulation itself is done in float64 here.
    b1, b2 = np.float32(beta1), np.float32(beta2)
    omb1, omb2 = float(np.float32(1) - b1), float(np.float32(1) - b2)
    m = float(b1) * m + omb1 * g
    v = float(b2) * v + omb2 * g * g
    return p - lr_t * m / (np.sqrt(v) + eps), m, v


def exponential_decay_staircase(lr, step, decay_steps, decay_rate):
    """tf.train.exponential_decay(..., staircase=True) (myutils.py:215-218)."""
    return lr * decay_rate ** (step // decay_steps)
