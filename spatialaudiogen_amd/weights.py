"""Variable inventory of the path and seeded synthetic initialisation.

Names are the TF1 checkpoint keys of the reference graph (scopes opened at
model.py:378-428; per-layer 'weights'/'biases'/'bn/*' from
pyutils/tflib/wrappers/core.py:21,69,127,191,210; ResNet18 block names from
pyutils/tflib/models/image/resnet.py:131-236).  Layouts are the TF ones:
conv HWIO [kh,kw,Cin,Cout]; conv2d_transpose [kh,kw,Cout,Cin] (core.py:118);
fully_connected [in,out].
"""
from collections import OrderedDict
import numpy as np

from .definitions import (AUDIO, VIDEO, FLOW, NO_SEPARATION, FREQ_MASK,
                          AENC_FILTERS, AENC_KERNELS)
from .geometry import Geometry


def resnet18_specs(scope, in_ch=3):
    """conv1 .. conv5_2 of the truncated ResNet18 (resnet.py:123-190)."""
    s = OrderedDict()

    def bn(prefix, c):
        for v in ('beta', 'gamma', 'moving_mean', 'moving_variance'):
            s['%s/bn/%s' % (prefix, v)] = (c,)

    s[scope + '/conv1/conv/weights'] = (7, 7, in_ch, 64)
    bn(scope + '/conv1/conv', 64)
    cin = 64
    for stage, cout in zip((2, 3, 4, 5), (64, 128, 256, 512)):
        for unit in (1, 2):
            p = '%s/conv%d_%d' % (scope, stage, unit)
            if unit == 1 and cin != cout:
                s[p + '/shortcut/weights'] = (1, 1, cin, cout)       # resnet.py:211-212 (no bias, no bn)
            s[p + '/conv_1/weights'] = (3, 3, cin, cout)
            bn(p + '/conv_1', cout)
            s[p + '/conv_2/weights'] = (3, 3, cout, cout)
            bn(p + '/conv_2', cout)
            cin = cout
    return s


def variable_specs(encoders, separation=FREQ_MASK, num_sep_tracks=32, loc_units=(512, 512),
                   geom=None):
    """Ordered {name: shape} for one model configuration."""
    geom = geom or Geometry()
    assert AUDIO in encoders, 'the audio encoder is mandatory (model.py:207 reads it unconditionally)'
    s = OrderedDict()
    cin = 1
    for l, (nf, k) in enumerate(zip(AENC_FILTERS, AENC_KERNELS)):
        s['audio_encoder/conv%d/weights' % (l + 1)] = (k[0], k[1], cin, nf)
        s['audio_encoder/conv%d/biases' % (l + 1)] = (nf,)
        cin = nf
    for enc in (VIDEO, FLOW):
        if enc in encoders:
            s.update(resnet18_specs(enc + '_encoder'))
    # bottleneck (model.py:203-239)
    enc_shapes = geom.encoder_shapes()
    h5, w5, c5 = enc_shapes[-1]
    cb = 0
    s['bottleneck/audio-fc/weights'] = (w5 * c5, 1024)
    s['bottleneck/audio-fc/biases'] = (1024,)
    cb += 1024
    for enc in (VIDEO, FLOW):
        if enc in encoders:
            s['bottleneck/%s-fc-red/weights' % enc] = (512, 128)
            s['bottleneck/%s-fc-red/biases' % enc] = (128,)
            s['bottleneck/%s-fc/weights' % enc] = (7 * 14 * 128, 512)
            s['bottleneck/%s-fc/biases' % enc] = (512,)
            cb += 512
    # localization (model.py:241-271)
    cin = cb
    for i, u in enumerate(loc_units):
        s['localization/fc%d/weights' % (i + 1)] = (cin, u)
        s['localization/fc%d/biases' % (i + 1)] = (u,)
        cin = u
    if separation == NO_SEPARATION and num_sep_tracks not in (None, 1):
        # the reference sizes fc3 with sep_num_tracks+1 whatever the mode (model.py:254) while NO_SEPARATION yields ONE track
        # (model.py:274-280): only sep_num_tracks = 1 is consistent, and that is what its drivers pass (deploy.py:62-63)
        raise ValueError("separation 'none' needs num_sep_tracks = 1 (got %r)" % (num_sep_tracks,))
    nsep = num_sep_tracks if separation != NO_SEPARATION else 1
    nlast = geom.num_out * geom.num_in * (nsep + 1)
    s['localization/fc%d/weights' % (len(loc_units) + 1)] = (cin, nlast)
    s['localization/fc%d/biases' % (len(loc_units) + 1)] = (nlast,)
    # separation (model.py:282-311)
    if separation == FREQ_MASK:
        s['separation/fc-feats/weights'] = (cb, AENC_FILTERS[-1])
        s['separation/fc-feats/biases'] = (AENC_FILTERS[-1],)
        nfs = [nsep * 1] + AENC_FILTERS[:-1]            # outputs of deconv1..5
        cin = 2 * AENC_FILTERS[-1]
        for l in reversed(range(5)):
            k = AENC_KERNELS[l]
            s['separation/deconv%d/weights' % (l + 1)] = (k[0], k[1], nfs[l], cin)
            s['separation/deconv%d/biases' % (l + 1)] = (nfs[l],)
            cin = nfs[l] + (AENC_FILTERS[l - 1] if l > 0 else 0)
    return s


def load_pretrained_resnet18(variables, pretrained, specs=None):
    """ResNet18.restore_pretrained (resnet.py:238-249), run by train.py:182-184 right after the initialisers and BEFORE an optional
    --resume restore: every MODEL variable under `video_encoder/` and `flow_encoder/` (conv weights and the four batch-norm vectors of
    every layer, model.py:198-199 -> cnn.restore_pretrained(inp_shape[-1], scope)) is assigned `pretrained[<name without the scope>]`.

    `pretrained`: the dict of `resnet18.npy` (np.load(...).all() in the reference: one pickled {name: array}) or a path to such a
    file.  As in the reference a variable of the trunk that the blob lacks is a KeyError, and an array of another shape an error
    (tf.assign would refuse it) - nothing is assigned in that case.  Both trunks receive the SAME ImageNet values (the flow trunk too:
    its input has three channels as well).  Returns the names assigned, in order; `variables` is updated in place."""
    if isinstance(pretrained, str):
        blob = np.load(pretrained, allow_pickle=True, encoding='latin1')       # (a Python 2 pickle)
        pretrained = blob.item() if hasattr(blob, 'item') and getattr(blob, 'dtype', None) == object else dict(blob)
    staged = OrderedDict()
    for name in (specs if specs is not None else variables):
        for scope in ('video_encoder', 'flow_encoder'):
            if name.startswith(scope + '/'):
                key = name[len(scope) + 1:]
                if key not in pretrained:
                    raise KeyError('resnet18 blob has no %r (needed by %s)' % (key, name))
                v = np.asarray(pretrained[key], np.float32)
                want = tuple(specs[name]) if specs is not None else tuple(np.shape(variables[name]))
                if tuple(v.shape) != want:
                    raise ValueError('resnet18 blob: %r has shape %s, %s needs %s' % (key, tuple(v.shape), name, want))
                staged[name] = np.ascontiguousarray(v)
    variables.update(staged)
    return list(staged)


def count_params(specs):
    return int(sum(int(np.prod(v)) for v in specs.values()))


def init_weights(specs, seed=0, mode='test', fc3_std=0.05):
    """Seeded synthetic weights, float32 numpy.

    mode='test'  : Xavier-uniform kernels, N(0,0.1) biases, gamma~U(0.5,1.5), beta~N(0,0.2)
                   (exercises every bias/BN path).
    mode='bench' : Xavier-uniform kernels, zero biases, gamma=1, beta=0 (the reference's
                   own initialisers, core.py:13,34), SURVEY 8(d).
    The last localization FC uses N(0, fc3_std^2): the reference's 0.001 (model.py:255)
    makes the output ~0 and an absolute-error test vacuous.
    Inputs/weights are regenerated from numpy PCG64(seed) — never stored.
    """
    rng = np.random.Generator(np.random.PCG64(seed))
    out = OrderedDict()
    last_fc = max(int(n.split('/')[1][2:]) for n in specs if n.startswith('localization/fc')
                  and n.endswith('weights'))
    for name, shape in specs.items():
        leaf = name.split('/')[-1]
        if leaf == 'weights':
            if name == 'localization/fc%d/weights' % last_fc:
                w = rng.normal(0.0, fc3_std, size=shape)
            else:
                rec = int(np.prod(shape[:-2])) if len(shape) > 2 else 1
                fan_in, fan_out = shape[-2] * rec, shape[-1] * rec
                lim = np.sqrt(6.0 / (fan_in + fan_out))
                w = rng.uniform(-lim, lim, size=shape)
        elif leaf == 'biases':
            w = rng.normal(0.0, 0.1, size=shape) if mode == 'test' else np.zeros(shape)
        elif leaf == 'gamma':
            w = rng.uniform(0.5, 1.5, size=shape) if mode == 'test' else np.ones(shape)
        elif leaf == 'beta':
            w = rng.normal(0.0, 0.2, size=shape) if mode == 'test' else np.zeros(shape)
        elif leaf == 'moving_mean':
            w = np.zeros(shape)
        elif leaf == 'moving_variance':
            w = np.ones(shape)
        else:
            raise ValueError(name)
        out[name] = np.ascontiguousarray(w, dtype=np.float32)
    return out


def synth_inputs(batch, encoders, seed=1234, geom=None):
    """Seeded synthetic inputs of SURVEY 8(d): audio [B,snd_size,1], video/flow [B,1,224,448,3]."""
    geom = geom or Geometry()
    rng = np.random.Generator(np.random.PCG64(seed))
    n = np.arange(geom.snd_size)[None, :]
    audio = np.zeros((batch, geom.snd_size))
    for _ in range(3):
        f = rng.uniform(100, 8000, size=(batch, 1))
        ph = rng.uniform(0, 2 * np.pi, size=(batch, 1))
        audio += np.sin(2 * np.pi * f * n / geom.audio_rate + ph)
    audio = np.clip(0.25 * audio + 0.05 * rng.normal(size=audio.shape), -1, 1)
    out = {AUDIO: audio.astype(np.float32)[:, :, None]}

    def smooth(x):   # 8x8 box blur by block-mean + nearest upsample (cheap, deterministic)
        b, h, w, c = x.shape
        y = x.reshape(b, h // 8, 8, w // 8, 8, c).mean(axis=(2, 4))
        return np.repeat(np.repeat(y, 8, axis=1), 8, axis=2)

    if VIDEO in encoders:
        img = rng.integers(0, 256, size=(batch, 224, 448, 3)).astype(np.float64)
        img = 0.5 * img + 0.5 * smooth(img)
        out[VIDEO] = (np.round(img) / 255. - 0.5).astype(np.float32)[:, None]     # myutils.py:88-89
    if FLOW in encoders:
        m = np.abs(rng.normal(0, 2, size=(batch, 224, 448, 1)))
        th = rng.uniform(0, 2 * np.pi, size=(batch, 224, 448, 1))
        fl = np.concatenate([m * np.cos(th), m * np.sin(th), m], axis=3)         # feeder.py:147-161
        fl = 0.5 * fl + 0.5 * smooth(fl)
        out[FLOW] = fl.astype(np.float32)[:, None]
    return out
