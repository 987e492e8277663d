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
    """AudioReader.get (feeder.py:64-90) on one contiguous array: `size` samples for the window at time t, zero
    padded before/after.  The padding is derived from start_frame = int((t - context/2) * rate), but the
    samples are read from int(start)*rate + int((start - int(start)) * rate) (feeder.py:81) — one sample
    earlier whenever the fractional part truncates differently (e.g. start = 1.2 -> 57599, not 57600).  Kept:
    it is what the reference feeds the network.  audio: [n_samples, C]."""
    start_time = t - context / 2
    start_frame = int(start_time * rate)
    num_frames = int(int(np.ceil(audio.shape[0] / float(rate))) * rate)      # AudioReader.num_frames = n_files * rate
    pad_before = pad_after = 0
    if start_frame < 0:
        pad_before = abs(start_frame)
        size -= pad_before
        start_time, start_frame = 0., 0
    if start_frame + size > num_frames:
        pad_after = start_frame + size - num_frames
        size -= pad_after
    read_start = int(start_time) * int(rate) + int((start_time - int(start_time)) * rate)
    chunk = audio[read_start:read_start + max(size, 0)]
    if pad_before or pad_after:
        chunk = np.concatenate([np.zeros((pad_before, audio.shape[1]), audio.dtype), chunk,
                                np.zeros((pad_after, audio.shape[1]), audio.dtype)], 0)
    return chunk


def frame_index(t, video_rate):
    """VideoReader.get_by_index (feeder.py:121)."""
    return max(int(t * video_rate), 0)


class ClipArrays(object):
    """An in-memory clip with SampleReader's interface (feeder.py:164-265): the arrays a reader would decode
    from <folder>/ambix, /video, /flow.  audio [n, C>=1] float; video/flow [n_frames, 224, 448, 3] already
    preprocessed (x/255-0.5, myutils.py:88-89; flow de-quantised, feeder.py:147-161)."""

    def __init__(self, audio, video=None, flow=None, audio_rate=48000, video_rate=10, chunks_t=None, duration=0.1,
                 context=1.0, start_time=0.5, sample_duration=None):
        self.audio, self.video, self.flow = audio, video, flow
        self.audio_rate, self.video_rate, self.context = audio_rate, video_rate, context
        self.audio_size = int(duration * audio_rate) + int(context * audio_rate) - 1
        if chunks_t is None:   # audio_pow.lst times (scraping/preprocess.py:146-153), one 1-s wav chunk per second
            n_files = int(np.ceil(audio.shape[0] / float(audio_rate)))
            chunks_t = [float('%.12g' % (i / 10. + 0.5)) for i in range((n_files - 1) * 10)]
        if start_time > 0.5:                                             # feeder.py:228-231
            chunks_t = [t for t in chunks_t if t >= start_time]
        if sample_duration is not None:
            chunks_t = [t for t in chunks_t if t < start_time + sample_duration]
        self.chunks_t = chunks_t
        self.head = -1

    def windowed(self, start_time, sample_duration):
        return ClipArrays(self.audio, self.video, self.flow, self.audio_rate, self.video_rate, None, 0.1, self.context,
                          start_time, sample_duration)

    def get(self):
        self.head += 1
        if self.head >= len(self.chunks_t):
            return None
        t = self.chunks_t[self.head]
        out = {'id': 'clip ' + str(t), 'ambix': audio_window(self.audio, t, self.context, self.audio_size, self.audio_rate)}
        fi = frame_index(t, self.video_rate)
        if self.video is not None:
            out['video'] = self.video[fi][np.newaxis]
        if self.flow is not None:
            out['flow'] = self.flow[fi][np.newaxis]
        return out


class W2XYZ(object):
    batch_size = 10            # deploy.py:50
    duration = 0.1             # deploy.py:49
    on_saturation = 'rerun'    # fp16x2 guard (SptAudioGen.inference_ops_checked): 'rerun' on bf16 planes | 'raise'
    # batches per forward call (round 6, include/sagen.h: grouped launch): `groups` consecutive batches of 10 windows run as ONE launch per
    # layer, each batch with its own batch-norm statistics - the output is bit-identical to groups = 1 (deploy.py:141 runs one batch
    # per sess.run); batches that do not fill a group (the clip's tail, a zero-padded partial batch) run one at a time
    groups = 1

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
        """Weights: the TF1 checkpoint of the model directory (tensor-bundle reader, checkpoint.py), or an
        .npz keyed by the TF variable names."""
        fn = os.path.join(model_dir, 'variables.npz')
        if os.path.exists(fn):
            with np.load(fn) as z:
                return {k: z[k] for k in z.files}
        from .checkpoint import latest_checkpoint, load_checkpoint
        prefix = latest_checkpoint(model_dir)
        if prefix is None:
            raise IOError('no checkpoint (or variables.npz) found in %s' % model_dir)
        return load_checkpoint(prefix)

    def _reader(self, source, deploy_start, deploy_duration):
        p = self.params
        if isinstance(source, ClipArrays):
            return source.windowed(deploy_start, deploy_duration)
        from .feeder import SampleReader
        # img_prep=None: frames stay the uint8 the JPEG decoder produced; x/255 - 0.5 (myutils.py:88-89) runs on the device
        # (sagen_forward_u8), bit-identical to img_prep_fcn() + float32 cast, at a quarter of the H2D bytes
        return SampleReader(source, ambi_order=p.ambi_order, audio_rate=p.audio_rate, video_rate=p.video_rate,
                            context=p.context, duration=self.duration, return_video=VIDEO in p.encoders,
                            img_prep=None, return_flow=FLOW in p.encoders, start_time=deploy_start,
                            sample_duration=deploy_duration, skip_silence_thr=None, shuffle=False,
                            random_rotations=False, skip_rate=None)                   # deploy.py:91-105

    def deploy(self, input_folder, deploy_start=0., deploy_duration=10., prefetch=True):
        """deploy.py:90-152.  `input_folder` is a clip directory (or a ClipArrays).  Returns
        [n_windows*snd_dur, 4] = W,Y,Z,X."""
        import torch
        from . import ops
        from .feeder import BatchPrefetcher, frames_to_float
        p, m = self.params, self.model
        reader = self._reader(input_folder, deploy_start, deploy_duration)
        if not reader.chunks_t:
            return np.zeros((0, 4), np.float32)
        dt = reader.chunks_t[0] - deploy_start                           # deploy.py:106-107
        reader.chunks_t = [t - dt for t in reader.chunks_t]
        use_v, use_f = VIDEO in p.encoders, FLOW in p.encoders

        def batches():                                                   # deploy.py:112-139
            while True:
                batch = []
                for _ in range(self.batch_size):
                    chunk = reader.get()
                    if chunk is None:
                        break
                    batch.append(chunk)
                if not batch:
                    return
                n = len(batch)
                out = {'n': n}
                audio = np.zeros((self.batch_size, self.audio_size, 1), np.float32)   # zero rows = deploy.py:125-127
                audio[:n] = np.stack([b['ambix'] for b in batch], 0)[:, :, :1]
                out['audio'] = audio
                for key, on in (('video', use_v), ('flow', use_f)):
                    if on:
                        clip = np.stack([b[key] for b in batch], 0)
                        if clip.dtype == np.uint8 and n == self.batch_size:
                            out[key] = clip                                   # decoded frames as they are (sagen_forward_u8)
                        else:                       # a partial batch is padded with 0.0 AFTER normalisation: float frames
                            x = np.zeros((self.batch_size, 1, 224, 448, 3), np.float32)
                            x[:n] = frames_to_float(clip)
                            out[key] = x
                yield out

        src = BatchPrefetcher(batches(), depth=2 * max(1, self.groups), pin=True) if prefetch else batches()
        outs = []
        G = max(1, int(self.groups))
        mg = self._grouped_model(G) if G > 1 else None

        def run(net, bs):               # one forward call over the batches `bs` (len(bs) == net.groups), results appended in order
            cat = lambda k: torch.cat([torch.as_tensor(b[k]) for b in bs], 0).to(m.device, non_blocking=True) if k in bs[0] else None
            a_dev = cat('audio')
            # deploy.py:141 - with the fp16x2 guard: a batch whose trunk planes clamped anything is re-run on bf16 planes
            pred = net.inference_ops_checked(a_dev, cat('video'), cat('flow'), on_saturation=self.on_saturation)
            wyzx = ops.assemble_wyzx(a_dev[:, :, 0].contiguous(), pred, m.snd_contx)   # deploy.py:143-152
            for i, b in enumerate(bs):
                outs.append(wyzx[i * self.batch_size:i * self.batch_size + b['n']].reshape(b['n'] * m.snd_dur, 4).cpu().numpy())

        pending = []
        groupable = lambda b: b['n'] == self.batch_size and all(b[k].dtype == pending[0][k].dtype for k in ('video', 'flow') if k in b)
        for b in src:
            if G > 1 and (not pending or groupable(b)):
                pending.append(b)
                if len(pending) == G:
                    run(mg, pending)
                    pending = []
                continue
            for q in pending:           # a batch that cannot join the group (partial / other frame type): everything in order, singly
                run(m, [q])
            pending = []
            run(m, [b])
        for q in pending:
            run(m, [q])
        return np.concatenate(outs, 0)

    def _grouped_model(self, G):
        """A second facade over the SAME device variables whose native contexts carry G batches per call (SptAudioGen(groups=G))."""
        if getattr(self, '_mg', None) is None or self._mg.groups != G:
            m = self.model
            self._mg = SptAudioGen(ambi_order=m.ambi_order, audio_rate=m.snd_rate, video_rate=m.vid_rate, context=m.context,
                                   sample_duration=m.duration, encoders=m.encoders, separation=m.separation, params=m.params,
                                   device=m.device, groups=G)
            self._mg.load_variables(m._variables)
        return self._mg


def parse_arguments(argv=None):
    """deploy.py:14-38 (the ffmpeg / 360-video outputs are outside this path; the ambisonic wav is written)."""
    import argparse
    parser = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    parser.add_argument('model_dir', help='Directory containing model snapshot.')
    parser.add_argument('input_folder', default='', help='Folder with input sample.')
    parser.add_argument('--deploy_start', default=0., type=float)
    parser.add_argument('--deploy_duration', default=10., type=float)
    parser.add_argument('--output_fn', default='output.wav', help='Output 4-channel (W,Y,Z,X) wav.')
    parser.add_argument('--gpu', type=int, default=0, help='GPU id')
    parser.add_argument('--groups', type=int, default=1,
                        help='batches of 10 windows per forward call (grouped launch: one launch per layer for all of them, each batch with '
                             'its own batch-norm statistics; output bit-identical to 1)')
    args = parser.parse_args(argv)
    if args.deploy_duration <= 0:
        args.deploy_duration = None
    return args


def main(argv=None):
    """deploy.py:155-198 up to save_wav."""
    import torch
    from .feeder import save_wav
    args = parse_arguments(argv)
    torch.cuda.set_device(args.gpu)
    model = W2XYZ(args.model_dir)
    model.groups = max(1, args.groups)
    ambi_pred = model.deploy(args.input_folder, args.deploy_start, args.deploy_duration)
    save_wav(args.output_fn, ambi_pred, model.params.audio_rate)
    print('wrote %s: %d samples x 4 channels (ACN W,Y,Z,X / SN3D)' % (args.output_fn, ambi_pred.shape[0]))


if __name__ == '__main__':
    main()
