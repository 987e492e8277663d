"""On-disk sample readers — the counterparts of the reference's feeder.py readers (AudioReader
feeder.py:50-103, VideoReader :106-132, FlowReader :135-161, SampleReader :164-278) and of
pyutils/iolib/audio.py load_wav / save_wav (:11-34), for the directory layout produced by
scraping/preprocess.py:

    <clip>/ambix/%06d.wav      1-second 4-channel (W,Y,Z,X) 48 kHz chunks
    <clip>/video/%06d.jpg      224x448 RGB frames at 10 fps
    <clip>/flow/%06d.jpg       polar-encoded optical flow + flow/flow_limits.npy   (optional)
    <clip>/audio_pow.lst       "<t> <rms power>" per 0.1 s window

Host-side only (numpy / scipy.io.wavfile / PIL); nothing here touches the GPU.  Arithmetic that decides
WHICH samples and frames a window gets (float truncations included) is kept exactly as the reference
writes it, because it is part of the deploy result.
"""
import os
import random
import threading

import numpy as np

try:                      # queue: Python 3
    import queue
except ImportError:       # pragma: no cover
    import Queue as queue


# ------------------------------------------------------------------------------------------------
# wav / image I/O
# ------------------------------------------------------------------------------------------------
def load_wav(fname, rate=None):
    """pyutils/iolib/audio.py:11-28.  Returns (float64 [n, channels] in [-1, 1), rate).  libsndfile
    scales integer PCM by 1/2^(bits-1); resampling (resampy in the reference) is not available offline, so a
    rate mismatch is an error."""
    from scipy.io import wavfile
    _rate, data = wavfile.read(fname)
    if data.ndim == 1:
        data = data.reshape(-1, 1)
    if data.dtype == np.int16:
        sig = data.astype(np.float64) / 32768.0
    elif data.dtype == np.int32:
        sig = data.astype(np.float64) / 2147483648.0
    elif data.dtype == np.uint8:
        sig = (data.astype(np.float64) - 128.0) / 128.0
    else:
        sig = data.astype(np.float64)
    if rate is not None and int(rate) != int(_rate):
        raise ValueError('%s is sampled at %d Hz, expected %d (no resampler available offline)' % (fname, _rate, rate))
    return sig, _rate


def save_wav(fname, signal, rate, subtype='PCM_16'):
    """pyutils/iolib/audio.py:31-34 (Sndfile Format('wav') = 16-bit PCM).  signal [n, channels] float."""
    from scipy.io import wavfile
    signal = np.asarray(signal)
    if subtype == 'PCM_16':
        pcm = np.clip(np.rint(np.clip(signal, -1.0, 1.0) * 32767.0), -32768, 32767).astype(np.int16)
        wavfile.write(fname, int(rate), pcm)
    elif subtype == 'FLOAT':
        wavfile.write(fname, int(rate), signal.astype(np.float32))
    else:
        raise ValueError(subtype)


def imread(fname):
    from PIL import Image
    with Image.open(fname) as im:
        return np.asarray(im.convert('RGB'))


# ------------------------------------------------------------------------------------------------
# readers
# ------------------------------------------------------------------------------------------------
class AudioReader(object):
    """feeder.py:50-103."""

    def __init__(self, audio_folder, rate=None, ambi_order=1):
        self.audio_folder = audio_folder
        fns = sorted(f for f in os.listdir(audio_folder) if f.endswith('.wav'))
        self.num_files = len(fns)
        sig, file_rate = load_wav(os.path.join(audio_folder, fns[0]))
        self.rate = float(file_rate) if rate is None else rate
        self.num_channels = min((sig.shape[1], (ambi_order + 1) ** 2))
        self.duration = self.num_files
        self.num_frames = int(self.duration * self.rate)

    def get(self, start_time, size, rotation=None):
        start_frame = int(start_time * self.rate)                       # feeder.py:66 (float truncation kept)
        pad_before, pad_after = 0, 0
        if start_frame < 0:
            pad_before = abs(start_frame)
            size -= pad_before
            start_time, start_frame = 0., 0
        if start_frame + size > self.num_frames:
            pad_after = start_frame + size - self.num_frames
            size -= pad_after
        index = range(int(start_time), min(int(np.ceil(start_time + size / float(self.rate))), self.num_files))
        fns = ['{}/{:06d}.wav'.format(self.audio_folder, i) for i in index]
        chunk = [load_wav(fn, self.rate)[0] for fn in fns]
        chunk = np.concatenate(chunk, axis=0) if len(chunk) > 1 else chunk[0]
        ss = int((start_time - int(start_time)) * self.rate)            # feeder.py:81
        chunk = chunk[ss:ss + size, :self.num_channels]
        if pad_before > 0:
            chunk = np.concatenate((np.zeros((pad_before, self.num_channels)), chunk), axis=0)
        if pad_after > 0:
            chunk = np.concatenate((chunk, np.zeros((pad_after, self.num_channels))), axis=0)
        if rotation is not None:
            assert -np.pi <= rotation < np.pi
            c, s = np.cos(rotation), np.sin(rotation)
            rot_mtx = np.array([[1, 0, 0, 0],      # W' = W
                                [0, c, 0, s],      # Y' = X sin + Y cos
                                [0, 0, 1, 0],      # Z' = Z
                                [0, -s, 0, c]])    # X' = X cos - Y sin
            chunk = np.dot(chunk, rot_mtx.T)
        return chunk


class VideoReader(object):
    """feeder.py:106-132."""

    def __init__(self, video_folder, rate=None, img_prep=None):
        raw_rate = 10.
        self.video_folder = video_folder
        self.rate = rate if rate is not None else raw_rate
        self.img_prep = img_prep if img_prep is not None else (lambda x: x)
        frame_fns = [fn for fn in os.listdir(video_folder) if fn.endswith('.jpg')]
        self.num_frames = len(frame_fns)
        self.duration = self.num_frames / raw_rate
        img = imread(os.path.join(video_folder, sorted(frame_fns)[0]))
        self.frame_shape = self.img_prep(img).shape

    def get_by_index(self, start_time, size, rotation=None):
        ss = max(int(start_time * self.rate), 0)                        # feeder.py:121
        chunk = [self.img_prep(imread(os.path.join(self.video_folder, '{:06d}.jpg'.format(fno))))
                 for fno in range(ss, ss + size)]
        chunk = np.stack(chunk, 0) if len(chunk) > 1 else chunk[0][np.newaxis]
        if rotation is not None:
            roll = -int(rotation / (2. * np.pi) * self.frame_shape[1])
            chunk = np.roll(chunk, roll, axis=2)
        return chunk


class FlowReader(object):
    """feeder.py:135-161: (angle, -, magnitude) bytes -> (m cos a, m sin a, m) with per-frame limits."""

    def __init__(self, flow_dir, flow_lims_fn, rate=None, flow_prep=None):
        self.reader = VideoReader(flow_dir, rate=rate)
        self.lims = np.load(flow_lims_fn)
        self.rate = self.reader.rate
        self.duration = self.reader.duration
        self.flow_prep = flow_prep if flow_prep is not None else (lambda x: x)

    def get_by_index(self, start_time, size, rotation=None):
        chunk = self.reader.get_by_index(start_time, size, rotation).astype(np.float32)
        ss = max(int(start_time * self.rate), 0)
        t = chunk.shape[0]
        m_min = self.lims[ss:ss + t, 0].reshape((-1, 1, 1))
        m_max = self.lims[ss:ss + t, 1].reshape((-1, 1, 1))
        chunk[:, :, :, 2] *= (m_max - m_min) / 255.
        chunk[:, :, :, 2] += m_min
        chunk[:, :, :, 0] *= (2 * np.pi) / 255.
        chunk[:, :, :, 1] = chunk[:, :, :, 2] * np.sin(chunk[:, :, :, 0])
        chunk[:, :, :, 0] = chunk[:, :, :, 2] * np.cos(chunk[:, :, :, 0])
        return chunk


def img_prep_fcn():
    """myutils.py:88-89."""
    return lambda x: x / 255. - 0.5


class SampleReader(object):
    """feeder.py:164-278: one clip folder -> 0.1 s samples {'id', 'ambix', 'video', 'flow'}."""

    def __init__(self, folder, ambi_order=1, audio_rate=48000, video_rate=10, context=1.0, duration=0.1,
                 return_video=True, img_prep=None, return_flow=False, flow_prep=None, skip_silence_thr=None,
                 shuffle=True, start_time=0.5, sample_duration=None, skip_rate=None, random_rotations=True,
                 num_threads=1, thread_id=0):
        a2v = float(audio_rate) / video_rate
        snd_dur, vid_dur, snd_ctx = duration * audio_rate, duration * video_rate, context * audio_rate
        self.video_id = os.path.split(folder)[-1]
        assert a2v == int(a2v) and float(snd_dur) == int(snd_dur) and float(vid_dur) == int(vid_dur) and float(snd_ctx) == int(snd_ctx)
        self.audio_reader = AudioReader(os.path.join(folder, 'ambix'), audio_rate, ambi_order)
        self.video_reader = VideoReader(os.path.join(folder, 'video'), video_rate, img_prep) if return_video else None
        if return_flow:
            flow_dir = os.path.join(folder, 'flow')
            self.flow_reader = FlowReader(flow_dir, os.path.join(flow_dir, 'flow_limits.npy'), video_rate, flow_prep)
        self.folder, self.duration, self.context = folder, duration, context
        self.audio_rate, self.video_rate = audio_rate, video_rate
        self.audio_size = int(snd_dur) + int(snd_ctx) - 1
        self.video_size = int(vid_dur)
        self.return_video, self.return_flow, self.random_rotations = return_video, return_flow, random_rotations

        lines = [l.strip().split() for l in open(os.path.join(folder, 'audio_pow.lst')) if l.strip()]
        chunks_t = [float(l[0]) for l in lines]
        chunks_pow = [float(l[1]) for l in lines]
        if skip_rate is not None:
            chunks_t, chunks_pow = chunks_t[::skip_rate], chunks_pow[::skip_rate]
        if skip_silence_thr is not None:
            chunks_t = [t for t, p in zip(chunks_t, chunks_pow) if p > skip_silence_thr]
        if start_time > 0.5:
            chunks_t = [t for t in chunks_t if t >= start_time]
        if sample_duration is not None:
            chunks_t = [t for t in chunks_t if t < start_time + sample_duration]
        if num_threads > 1:
            lims = np.linspace(0, len(chunks_t), num_threads + 1).astype(int)
            chunks_t = chunks_t[lims[thread_id]:lims[thread_id + 1]]
        if shuffle:
            random.shuffle(chunks_t)
        self.chunks_t = chunks_t
        self.head = -1

    def get(self):
        self.head += 1
        if self.head >= len(self.chunks_t):
            return None
        cur_t = self.cur_t = self.chunks_t[self.head]
        rotation = random.random() * 2 * np.pi - np.pi if self.random_rotations else None
        chunks = {'id': self.video_id + ' ' + str(cur_t)}
        chunks['ambix'] = self.audio_reader.get(cur_t - self.context / 2, self.audio_size, rotation)
        if self.return_video:
            chunks['video'] = self.video_reader.get_by_index(cur_t, self.video_size, rotation)
        if self.return_flow:
            chunks['flow'] = self.flow_reader.get_by_index(cur_t, self.video_size, rotation)
        return chunks

    def loop_chunks(self, n=np.inf):
        k = 0
        while True:
            k += 1
            if k > n:
                break
            chunks = self.get()
            if chunks is None:
                break
            yield chunks


# ------------------------------------------------------------------------------------------------
# background batching (the role of feeder.Feeder's threads + tf.PaddingFIFOQueue, feeder.py:281-435)
# ------------------------------------------------------------------------------------------------
class BatchPrefetcher(object):
    """Reader threads decode samples into a bounded queue of ready batches (numpy, optionally pinned torch
    tensors) so the GPU path is not starved by wav/jpg decoding.  `make_batches` is any iterator of dicts of
    stacked arrays; order is preserved (one producer thread per prefetcher)."""

    def __init__(self, make_batches, depth=2, pin=False):
        self.q = queue.Queue(maxsize=depth)
        self.pin = pin
        self._stop = False
        self.thread = threading.Thread(target=self._run, args=(make_batches,))
        self.thread.daemon = True
        self.thread.start()

    def _run(self, make_batches):
        try:
            for batch in make_batches:
                if self._stop:
                    break
                if self.pin:
                    import torch
                    batch = {k: (torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)).pin_memory()
                                 if isinstance(v, np.ndarray) else v) for k, v in batch.items()}
                self.q.put(batch)
        finally:
            self.q.put(None)

    def __iter__(self):
        while True:
            b = self.q.get()
            if b is None:
                return
            yield b

    def close(self):
        self._stop = True
        try:
            while self.q.get_nowait() is not None:
                pass
        except queue.Empty:
            pass
