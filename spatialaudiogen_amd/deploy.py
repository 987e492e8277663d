"""W2XYZ — the deploy driver: same surface as the reference class (deploy.py:41-152) on the HIP path.

    model = W2XYZ(model_dir)                      # reads <model_dir>/train-params.txt + weights
    ambi = model.deploy(input_folder, 0., 10.)    # -> ndarray [N*4800, 4]  (W, Y, Z, X)

Host-side only: window arithmetic, batching in groups of 10 with zero-padded last group
(deploy.py:112-139 — the padding is part of the result because batch-norm runs on batch statistics),
H2D staging and the final [mono | prediction] assembly.  All tensor arithmetic runs in
libsagen_hip.so through SptAudioGen.inference_ops.
"""
import os
import numpy as np

from .definitions import AUDIO, VIDEO, FLOW, NO_SEPARATION
from .model import SptAudioGen, SptAudioGenParams


def load_params(model_dir):
    """train-params.txt parser with the reference's type coercions and legacy defaults
    (myutils.py:40-85)."""
    params = {}
    with open(os.path.join(model_dir, 'train-params.txt')) as f:
        for l in f:
            if ':' in l:
                params[l.split(':')[0]] = l.strip().split(':')[1].strip()
    for k in ['encoders', 'separation']:
        params[k] = params[k].lower()
    for k in ('ambi_order', 'audio_rate', 'video_rate'):
        params[k] = int(params[k])
    params['context'] = float(params['context'])
    params['sample_dur'] = float(params.get('sample_dur', 0.1))
    params['encoders'] = [enc.strip()[1:-1] for enc in params['encoders'][1:-1].split(',')]
    params['num_sep_tracks'] = int(params.get('num_sep_tracks', '64'))
    params['fft_window'] = float(params.get('fft_window', '0.025'))

    def int_list(key, default):
        s = params.get(key, default)
        return [int(v.strip()) for v in s[1:-1].split(',')] if len(s[1:-1]) > 0 else []
    params['context_units'] = int_list('context_units', '[64, 128, 128]')
    params['freq_mask_units'] = int_list('freq_mask_units', '[]')
    params['loc_units'] = int_list('loc_units', '[256, 256]')

    class Struct(object):
        def __init__(self, **entries):
            self.__dict__.update(entries)
    return Struct(**params)


def window_times(chunks_t, deploy_start, deploy_duration):
    """feeder.py:228-231 filters + deploy.py:106-107 shift (float64 arithmetic kept as written)."""
    ts = list(chunks_t)
    if deploy_start > 0.5:
        ts = [t for t in ts if t >= deploy_start]
    if deploy_duration is not None:
        ts = [t for t in ts if t < deploy_start + deploy_duration]
    if not ts:
        return []
    dt = ts[0] - deploy_start
    return [t - dt for t in ts]


def audio_window(audio, t, context, size, rate):
    """AudioReader.get (feeder.py:64-90): `size` samples starting at int((t - context/2) * rate), zero
    padded before/after.  audio: [n_samples, C]."""
    start_time = t - context / 2
    start_frame = int(start_time * rate)
    num_frames = audio.shape[0]
    pad_before = pad_after = 0
    if start_frame < 0:
        pad_before = abs(start_frame)
        size -= pad_before
        start_frame = 0
    if start_frame + size > num_frames:
        pad_after = start_frame + size - num_frames
        size -= pad_after
    chunk = audio[start_frame:start_frame + max(size, 0)]
    if pad_before or pad_after:
        chunk = np.concatenate([np.zeros((pad_before, audio.shape[1]), audio.dtype), chunk,
                                np.zeros((pad_after, audio.shape[1]), audio.dtype)], 0)
    return chunk


def frame_index(t, video_rate):
    """VideoReader.get_by_index (feeder.py:121)."""
    return max(int(t * video_rate), 0)


class ClipArrays(object):
    """An in-memory clip: the arrays SampleReader would read from <folder>/ambix, /video, /flow
    (feeder.py:164-239).  audio [n, C>=1] float; video/flow [n_frames, 224, 448, 3] already
    preprocessed (x/255-0.5, myutils.py:88-89; flow de-quantised, feeder.py:147-161)."""

    def __init__(self, audio, video=None, flow=None, audio_rate=48000, chunks_t=None, duration=0.1, context=1.0):
        self.audio, self.video, self.flow = audio, video, flow
        self.audio_rate = audio_rate
        if chunks_t is None:   # audio_pow.lst times (scraping/preprocess.py:146-153), one 1-s wav chunk per second
            n_files = int(np.ceil(audio.shape[0] / float(audio_rate)))
            chunks_t = [float('%.12g' % (i / 10. + 0.5)) for i in range((n_files - 1) * 10)]
        self.chunks_t = chunks_t


class W2XYZ(object):
    batch_size = 10            # deploy.py:50
    duration = 0.1             # deploy.py:49

    def __init__(self, model_dir=None, params=None, variables=None, device=None):
        if params is None:
            params = load_params(model_dir)
        self.params = params
        num_sep = params.num_sep_tracks if params.separation != NO_SEPARATION else 1
        net_params = SptAudioGenParams(sep_num_tracks=num_sep, ctx_feats_fc_units=params.context_units,
                                       loc_fc_units=params.loc_units, sep_freq_mask_fc_units=params.freq_mask_units,
                                       sep_fft_window=params.fft_window)
        self.model = SptAudioGen(ambi_order=params.ambi_order, audio_rate=params.audio_rate,
                                 video_rate=params.video_rate, context=params.context,
                                 sample_duration=self.duration, encoders=params.encoders,
                                 separation=params.separation, params=net_params, device=device)
        self.audio_size = self.model.snd_dur + self.model.snd_contx - 1
        self.video_size = int(self.duration * params.video_rate)
        if variables is None:
            variables = self._load_variables(model_dir)
        self.model.load_variables(variables)

    @staticmethod
    def _load_variables(model_dir):
        """Weights exported as one .npz keyed by the TF variable names (a TF1 tensor-bundle reader is
        SURVEY 8f-1, not part of this round)."""
        fn = os.path.join(model_dir, 'variables.npz')
        if not os.path.exists(fn):
            raise IOError('%s not found: export the checkpoint variables to an .npz keyed by TF names' % fn)
        with np.load(fn) as z:
            return {k: z[k] for k in z.files}

    def deploy(self, clip, deploy_start=0., deploy_duration=10.):
        """deploy.py:90-152.  `clip` is a ClipArrays. Returns [n_windows*snd_dur, 4] = W,Y,Z,X."""
        import torch
        from . import ops
        p, m = self.params, self.model
        ts = window_times(clip.chunks_t, deploy_start, deploy_duration)
        use_v, use_f = VIDEO in p.encoders, FLOW in p.encoders
        outs = []
        for g in range(0, len(ts), self.batch_size):
            group = ts[g:g + self.batch_size]
            n = len(group)
            audio = np.zeros((self.batch_size, self.audio_size, 1), np.float32)       # zero rows = deploy.py:125-127
            video = np.zeros((self.batch_size, 1, 224, 448, 3), np.float32) if use_v else None
            flow = np.zeros((self.batch_size, 1, 224, 448, 3), np.float32) if use_f else None
            for i, t in enumerate(group):
                audio[i, :, 0] = audio_window(clip.audio, t, p.context, self.audio_size, p.audio_rate)[:, 0]
                fi = frame_index(t, p.video_rate)
                if use_v:
                    video[i, 0] = clip.video[fi]
                if use_f:
                    flow[i, 0] = clip.flow[fi]
            a_dev = torch.as_tensor(audio).to(m.device)
            pred = m.inference_ops(a_dev, video, flow)                                 # deploy.py:141
            wyzx = ops.assemble_wyzx(a_dev[:, :, 0].contiguous(), pred, m.snd_contx)   # deploy.py:143-152
            outs.append(wyzx[:n].reshape(n * m.snd_dur, 4).cpu().numpy())
        if not outs:
            return np.zeros((0, 4), np.float32)
        return np.concatenate(outs, 0)
