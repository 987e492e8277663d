"""Clip folders -> network inputs: the host side of the data path either end of the HIP forward.

Directory layout (what scraping/preprocess.py of the reference leaves behind, one folder per clip):

    ambix/%06d.wav      1-second pieces of the 4-channel first-order ambisonic track (W,Y,Z,X), 48 kHz PCM16
    video/%06d.jpg      224x448 RGB frames, 10 per second
    flow/%06d.jpg       optical flow, polar-coded in the R (angle) and B (magnitude) bytes, + flow/flow_limits.npy
    audio_pow.lst       one "<window time> <rms power>" line per 0.1 s window

This module is organised around a *window table*: for a list of window times it computes, in one vectorised pass,
which samples and which frame every window reads (`window_table`), and the stores below (`WavPieces`, `JpgFrames`,
`FlowFrames`) serve those reads from small decode caches, so consecutive windows - which overlap by 90 % - do not
decode the same wav piece or jpg twice.  `SampleReader` puts the reference's reader interface (constructor keywords,
`chunks_t`, `get()`, `loop_chunks()`; reference feeder.py:164-278) on top, because deploy.py / eval.py drive it.

What is pinned to the reference, and only that: the float arithmetic that decides WHICH samples and frame a window
gets -
    first sample of the padded window   int((t - context/2) * rate)                    feeder.py:66
    sample offset inside its 1-s piece  int((start - int(start)) * rate)               feeder.py:81
    frame index                         max(int(t * video_rate), 0)                    feeder.py:121
    time filters on audio_pow.lst       skip_rate, silence, start, duration, threads   feeder.py:222-238
- the truncations differ by one sample for about a fifth of the windows, and the network sees that.
Nothing here touches the GPU.
"""
import collections
import os
import random
import threading

import numpy as np

try:
    import queue
except ImportError:       # pragma: no cover
    import Queue as queue


# ------------------------------------------------------------------------------------------------
# file formats
# ------------------------------------------------------------------------------------------------
_PCM_SCALE = {np.dtype(np.int16): (0.0, 32768.0), np.dtype(np.int32): (0.0, 2147483648.0), np.dtype(np.uint8): (128.0, 128.0)}


def load_wav(fname, rate=None):
    """(float64 [n, channels] in [-1, 1), rate) - integer PCM scaled by 1/2^(bits-1) the way libsndfile does for
    the reference (pyutils/iolib/audio.py:11-28).  There is no resampler offline: a rate mismatch raises."""
    from scipy.io import wavfile
    file_rate, pcm = wavfile.read(fname)
    if rate is not None and int(rate) != int(file_rate):
        raise ValueError('%s is sampled at %d Hz, expected %d (no resampler available offline)' % (fname, file_rate, rate))
    pcm = pcm[:, None] if pcm.ndim == 1 else pcm
    offset, scale = _PCM_SCALE.get(pcm.dtype, (0.0, 1.0))
    return (pcm.astype(np.float64) - offset) / scale, file_rate


def save_wav(fname, signal, rate, subtype='PCM_16'):
    """[n, channels] float -> wav; 16-bit PCM by default (the reference's Format('wav'), pyutils/iolib/audio.py:31-34)."""
    from scipy.io import wavfile
    x = np.asarray(signal)
    if subtype == 'FLOAT':
        wavfile.write(fname, int(rate), x.astype(np.float32))
    elif subtype == 'PCM_16':
        wavfile.write(fname, int(rate), np.rint(np.clip(x, -1.0, 1.0) * 32767.0).astype(np.int16))
    else:
        raise ValueError('unsupported wav subtype %r' % (subtype,))


def imread(fname):
    from PIL import Image
    with Image.open(fname) as im:
        return np.asarray(im.convert('RGB'))


def img_prep_fcn():
    """Pixel normalisation of the video encoder input, x/255 - 0.5 (myutils.py:88-89)."""
    return lambda x: x / 255. - 0.5


def frames_to_float(frames):
    """Decoded uint8 frames -> the float32 tensor the reference's feeder hands to the graph (img_prep_fcn in double precision, then the
    float32 cast of the batch assembly).  The readers of this package keep video frames as decoded (uint8: a quarter of the H2D bytes;
    sagen_forward_u8 / sagen_train_step_u8 apply the same normalisation on the device, bit-identical) - this is the host-side form
    for the one case that needs float frames: a zero-padded partial batch, whose padding is 0.0 AFTER normalisation (deploy.py:125-127),
    a value no uint8 pixel maps to."""
    frames = np.asarray(frames)
    if frames.dtype != np.uint8:
        return frames.astype(np.float32)
    return (frames.astype(np.float64) / 255. - 0.5).astype(np.float32)


# ------------------------------------------------------------------------------------------------
# window table: the pinned arithmetic, vectorised over the window times
# ------------------------------------------------------------------------------------------------
WindowTable = collections.namedtuple('WindowTable', 'lead read_from read_count trail frame')


def window_table(times, context, size, audio_rate, total_samples, video_rate=10):
    """For every window time t: `lead` zeros, then `read_count` samples starting at absolute sample `read_from`, then
    `trail` zeros make up the `size`-sample audio input; `frame` is the video / flow frame index.
    Float64 / truncation semantics are those of the reference lines quoted in the module docstring."""
    t = np.asarray(times, dtype=np.float64)
    rate = audio_rate
    start = t - context / 2
    first = np.trunc(start * rate).astype(np.int64)              # int() truncates toward zero
    lead = np.where(first < 0, -first, 0)
    start = np.where(first < 0, 0.0, start)
    first = np.maximum(first, 0)
    count = size - lead
    trail = np.maximum(first + count - int(total_samples), 0)
    count = np.maximum(count - trail, 0)
    whole = np.trunc(start)
    read_from = whole.astype(np.int64) * int(rate) + np.trunc((start - whole) * rate).astype(np.int64)
    frame = np.maximum(np.trunc(t * video_rate).astype(np.int64), 0)
    return WindowTable(lead, read_from, count, trail, frame)


def select_times(times, powers=None, skip_rate=None, skip_silence_thr=None, start_time=0.5, sample_duration=None,
                 num_threads=1, thread_id=0):
    """The window-time filters a reader applies to audio_pow.lst, in the reference's order (feeder.py:222-238)."""
    times = list(times)
    powers = list(powers) if powers is not None else [np.inf] * len(times)
    keep_every = 1 if skip_rate is None else skip_rate
    times, powers = times[::keep_every], powers[::keep_every]
    floor = -np.inf if skip_silence_thr is None else skip_silence_thr
    times = [t for t, p in zip(times, powers) if p > floor]
    lo = start_time if start_time > 0.5 else -np.inf                 # (the list itself starts at 0.5)
    hi = np.inf if sample_duration is None else start_time + sample_duration
    times = [t for t in times if lo <= t < hi]
    if num_threads > 1:                                              # contiguous share of this reader thread
        cut = np.linspace(0, len(times), num_threads + 1).astype(int)
        times = times[cut[thread_id]:cut[thread_id + 1]]
    return times


def read_pow_list(fname):
    """audio_pow.lst -> (times, powers)."""
    times, powers = [], []
    with open(fname) as f:
        for line in f:
            cols = line.split()
            if cols:
                times.append(float(cols[0]))
                powers.append(float(cols[1]))
    return times, powers


def rotation_matrix_z(angle):
    """First-order ambisonic (W,Y,Z,X) rotation about the vertical axis: W, Z unchanged; (Y, X) rotate as a vector
    (the augmentation of feeder.py:92-101)."""
    c, s = np.cos(angle), np.sin(angle)
    m = np.eye(4)
    m[1, 1], m[1, 3], m[3, 1], m[3, 3] = c, s, -s, c
    return m


# ------------------------------------------------------------------------------------------------
# decode caches
# ------------------------------------------------------------------------------------------------
class _Lru(object):
    def __init__(self, capacity):
        self.capacity, self.items = capacity, collections.OrderedDict()

    def fetch(self, key, make):
        if key in self.items:
            self.items.move_to_end(key)
            return self.items[key]
        value = make(key)
        self.items[key] = value
        while len(self.items) > self.capacity:
            self.items.popitem(last=False)
        return value


class WavPieces(object):
    """The 1-second wav pieces of a clip as one virtual sample axis (piece i covers samples [i*rate, (i+1)*rate))."""

    def __init__(self, folder, rate=None, ambi_order=1, cache=4):
        self.folder = str(folder)
        self.num_files = sum(1 for f in os.listdir(folder) if f.endswith('.wav'))
        if self.num_files == 0:
            raise IOError('no wav pieces in %s' % folder)
        first, file_rate = load_wav(self._path(0))
        self.rate = float(file_rate) if rate is None else rate
        self.num_channels = min(first.shape[1], (ambi_order + 1) ** 2)
        self.duration = self.num_files
        self.num_frames = int(self.duration * self.rate)
        self._cache = _Lru(cache)

    def _path(self, i):
        return os.path.join(self.folder, '%06d.wav' % i)

    def _piece(self, i):
        return self._cache.fetch(i, lambda k: load_wav(self._path(k), self.rate)[0][:, :self.num_channels])

    def read(self, read_from, count):
        """`count` samples from absolute sample `read_from` (pieces decoded once and cached)."""
        out = np.zeros((count, self.num_channels))
        per = int(self.rate)
        done = 0
        while done < count:
            i, off = divmod(read_from + done, per)
            if i >= self.num_files:
                break
            piece = self._piece(i)
            n = min(count - done, piece.shape[0] - off)
            if n <= 0:
                break
            out[done:done + n] = piece[off:off + n]
            done += n
        return out

    def window(self, t_start, size, rotation=None):
        """`size` samples from time `t_start` (seconds), zero-padded outside the clip."""
        tab = window_table([t_start], 0.0, size, self.rate, self.num_frames)
        return self.assemble(tab, 0, size, rotation)

    def assemble(self, tab, k, size, rotation=None):
        out = np.zeros((size, self.num_channels))
        lead, n = int(tab.lead[k]), int(tab.read_count[k])
        out[lead:lead + n] = self.read(int(tab.read_from[k]), n)
        if rotation is not None:
            if not -np.pi <= rotation < np.pi:
                raise ValueError('rotation must lie in [-pi, pi)')
            out = out.dot(rotation_matrix_z(rotation)[:self.num_channels, :self.num_channels].T)
        return out


class AudioReader(WavPieces):
    """Reference-named view of WavPieces: get(start_time, size, rotation) (feeder.py:50-103)."""

    def get(self, start_time, size, rotation=None):
        return self.window(start_time, size, rotation)


class JpgFrames(object):
    """%06d.jpg frames of a folder, preprocessed on decode and cached."""

    RAW_RATE = 10.

    def __init__(self, folder, rate=None, prep=None, cache=4):
        self.folder = str(folder)
        self.rate = self.RAW_RATE if rate is None else rate
        self.prep = prep
        names = sorted(f for f in os.listdir(folder) if f.endswith('.jpg'))
        if not names:
            raise IOError('no jpg frames in %s' % folder)
        self.num_frames = len(names)
        self.duration = self.num_frames / self.RAW_RATE
        self._cache = _Lru(cache)
        self.frame_shape = self.frame(int(os.path.splitext(names[0])[0])).shape

    def frame(self, index):
        def decode(i):
            img = imread(os.path.join(self.folder, '%06d.jpg' % i))
            return self.prep(img) if self.prep is not None else img
        return self._cache.fetch(index, decode)

    def frames(self, first, count, rotation=None):
        clip = np.stack([self.frame(i) for i in range(first, first + count)], 0)
        if rotation:                      # equirectangular frames: a yaw rotation is a horizontal roll
            clip = np.roll(clip, -int(rotation / (2. * np.pi) * self.frame_shape[1]), axis=2)
        return clip

    def get_by_index(self, start_time, size, rotation=None):
        return self.frames(max(int(start_time * self.rate), 0), size, rotation)


VideoReader = JpgFrames


class FlowFrames(object):
    """Optical flow frames: byte 0 = angle / 2pi * 255, byte 2 = magnitude scaled into the frame's [min, max] of
    flow_limits.npy; decoded to (m cos a, m sin a, m) float32 (feeder.py:135-161)."""

    def __init__(self, folder, limits_fn, rate=None, flow_prep=None):
        self.jpgs = JpgFrames(folder, rate=rate)
        self.limits = np.load(limits_fn)
        self.rate, self.duration = self.jpgs.rate, self.jpgs.duration
        self.flow_prep = flow_prep

    def frames(self, first, count, rotation=None):
        # the reference's op order and rounding points (feeder.py:147-160): the limits are float64, so each in-place update of the
        # float32 frames is computed in double and rounded once; the angle scale is a Python scalar (float32 arithmetic)
        out = self.jpgs.frames(first, count, rotation).astype(np.float32)
        lo = self.limits[first:first + count, 0].reshape(-1, 1, 1)
        hi = self.limits[first:first + count, 1].reshape(-1, 1, 1)
        out[..., 2] *= (hi - lo) / 255.
        out[..., 2] += lo
        out[..., 0] *= (2 * np.pi) / 255.
        out[..., 1] = out[..., 2] * np.sin(out[..., 0])
        out[..., 0] = out[..., 2] * np.cos(out[..., 0])
        return self.flow_prep(out) if self.flow_prep is not None else out

    def get_by_index(self, start_time, size, rotation=None):
        return self.frames(max(int(start_time * self.rate), 0), size, rotation)


FlowReader = FlowFrames


# ------------------------------------------------------------------------------------------------
# one clip -> 0.1 s samples
# ------------------------------------------------------------------------------------------------
class SampleReader(object):
    """Samples {'id', 'ambix' [audio_size, C], 'video' / 'flow' [n, 224, 448, 3]} of one clip folder, one per selected
    window time, with the reference reader's interface (feeder.py:164-278).  `chunks_t` may be re-assigned before
    reading (deploy.py:106-107 shifts it)."""

    def __init__(self, folder, ambi_order=1, audio_rate=48000, video_rate=10, context=1.0, duration=0.1,
                 return_video=True, img_prep=None, return_flow=False, flow_prep=None, skip_silence_thr=None,
                 shuffle=True, start_time=0.5, sample_duration=None, skip_rate=None, random_rotations=True,
                 num_threads=1, thread_id=0):
        for name, v in (('audio/video rate ratio', float(audio_rate) / video_rate), ('duration*audio_rate', duration * audio_rate),
                        ('duration*video_rate', duration * video_rate), ('context*audio_rate', context * audio_rate)):
            if float(v) != int(v):
                raise ValueError('%s must be an integer (got %r)' % (name, v))
        self.folder = folder
        self.video_id = os.path.basename(os.path.normpath(folder))
        self.duration, self.context = duration, context
        self.audio_rate, self.video_rate = audio_rate, video_rate
        self.audio_size = int(duration * audio_rate) + int(context * audio_rate) - 1
        self.video_size = int(duration * video_rate)
        self.return_video, self.return_flow, self.random_rotations = return_video, return_flow, random_rotations
        self.audio_reader = AudioReader(os.path.join(folder, 'ambix'), audio_rate, ambi_order)
        self.video_reader = JpgFrames(os.path.join(folder, 'video'), video_rate, img_prep) if return_video else None
        self.flow_reader = None
        if return_flow:
            fdir = os.path.join(folder, 'flow')
            self.flow_reader = FlowFrames(fdir, os.path.join(fdir, 'flow_limits.npy'), video_rate, flow_prep)
        times, powers = read_pow_list(os.path.join(folder, 'audio_pow.lst'))
        times = select_times(times, powers, skip_rate, skip_silence_thr, start_time, sample_duration, num_threads, thread_id)
        if shuffle:
            random.shuffle(times)
        self.chunks_t = times
        self.head = -1
        self.cur_t = None

    def table(self):
        """Window table of the current `chunks_t` (see window_table)."""
        return window_table(self.chunks_t, self.context, self.audio_size, self.audio_rate, self.audio_reader.num_frames,
                            self.video_rate)

    def sample_at(self, t, rotation=None):
        """The sample of window time `t` (any time, not only the listed ones)."""
        tab = window_table([t], self.context, self.audio_size, self.audio_rate, self.audio_reader.num_frames, self.video_rate)
        sample = {'id': '%s %s' % (self.video_id, t),
                  'ambix': self.audio_reader.assemble(tab, 0, self.audio_size, rotation)}
        if self.return_video:
            sample['video'] = self.video_reader.frames(int(tab.frame[0]), self.video_size, rotation)
        if self.return_flow:
            sample['flow'] = self.flow_reader.frames(int(tab.frame[0]), self.video_size, rotation)
        return sample

    def get(self):
        self.head += 1
        if self.head >= len(self.chunks_t):
            return None
        self.cur_t = self.chunks_t[self.head]
        return self.sample_at(self.cur_t, random.random() * 2 * np.pi - np.pi if self.random_rotations else None)

    def loop_chunks(self, n=np.inf):
        served = 0
        while served < n:
            sample = self.get()
            if sample is None:
                return
            served += 1
            yield sample


# ------------------------------------------------------------------------------------------------
# background batching (the role of the reference's feeder threads + TF queue, feeder.py:281-435)
# ------------------------------------------------------------------------------------------------
class BatchPrefetcher(object):
    """One producer thread decodes batches ahead of the GPU into a bounded queue (optionally as pinned torch tensors).
    Order is preserved.  A failure in the producer (missing / short wav or jpg, bad flow limits) is re-raised in the
    consumer at the position where it happened - it never looks like a clean end of stream."""

    _END = object()

    def __init__(self, make_batches, depth=2, pin=False):
        self.q = queue.Queue(maxsize=max(1, depth))
        self.pin = pin
        self._stop = threading.Event()
        self.thread = threading.Thread(target=self._run, args=(make_batches,))
        self.thread.daemon = True
        self.thread.start()

    def _put(self, item):
        while not self._stop.is_set():
            try:
                self.q.put(item, timeout=0.05)
                return True
            except queue.Full:
                continue
        return False

    def _run(self, make_batches):
        try:
            for batch in make_batches:
                if self.pin:
                    import torch
                    # (decoded frames stay uint8 - the device normalises them; everything else travels as float32)
                    batch = {k: (torch.from_numpy(np.ascontiguousarray(v, dtype=np.uint8 if v.dtype == np.uint8 else np.float32)).pin_memory()
                                 if isinstance(v, np.ndarray) else v) for k, v in batch.items()}
                if not self._put(batch):
                    return
            self._put(self._END)
        except BaseException as e:          # handed to the consumer
            self._put(e)

    def __iter__(self):
        while True:
            item = self.q.get()
            if item is self._END:
                return
            if isinstance(item, BaseException):
                raise item
            yield item

    def close(self):
        """Stop the producer (it may be blocked on a full queue) and drop what is queued."""
        self._stop.set()
        try:
            while True:
                self.q.get_nowait()
        except queue.Empty:
            pass
        self.thread.join(timeout=5.0)
