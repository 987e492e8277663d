"""On-disk readers: a synthetic clip directory in the reference's layout is read back
through SampleReader and must give exactly what the in-memory ClipArrays path gives."""
import os

import numpy as np
import pytest

from spatialaudiogen_amd import feeder as F
from spatialaudiogen_amd.deploy import ClipArrays, audio_window, frame_index
from util import rng


def make_clip(root, secs=3, flow=False, seed=0, video=True):
    from PIL import Image
    r = rng(seed)
    os.makedirs(os.path.join(root, 'ambix')); os.makedirs(os.path.join(root, 'video'))
    audio = np.clip(0.3 * r.normal(size=(secs * 48000, 4)), -1, 1)
    for i in range(secs):
        F.save_wav(os.path.join(root, 'ambix', '%06d.wav' % i), audio[i * 48000:(i + 1) * 48000], 48000)
    with open(os.path.join(root, 'audio_pow.lst'), 'w') as f:
        for i in range((secs - 1) * 10):
            t = i / 10. + 0.5
            f.write('{} {}\n'.format(t, 0.1 + 0.01 * i))
    if not video:                                                  # audio-only clip (no frames to decode)
        return audio
    frames = r.integers(0, 256, size=(secs * 10, 224, 448, 3)).astype(np.uint8)
    frames = (frames // 32) * 32                                  # coarse levels survive JPEG poorly anyway: decode is the truth
    for i, fr in enumerate(frames):
        Image.fromarray(fr).save(os.path.join(root, 'video', '%06d.jpg' % i), quality=95)
    if flow:
        os.makedirs(os.path.join(root, 'flow'))
        for i in range(secs * 10):
            Image.fromarray(frames[(i * 7) % len(frames)]).save(os.path.join(root, 'flow', '%06d.jpg' % i), quality=95)
        np.save(os.path.join(root, 'flow', 'flow_limits.npy'), np.stack([np.zeros(secs * 10), np.linspace(1, 4, secs * 10)], 1))
    return audio


def test_wav_round_trip_is_pcm16(tmp_path):
    x = np.clip(0.5 * rng(1).normal(size=(1000, 4)), -1, 1)
    fn = str(tmp_path / 'a.wav')
    F.save_wav(fn, x, 48000)
    y, rate = F.load_wav(fn, 48000)
    assert rate == 48000 and y.shape == x.shape
    assert np.abs(y - x).max() <= 2.0 / 32768          # written x32767 (rounded), read /32768 like libsndfile
    with pytest.raises(ValueError):
        F.load_wav(fn, 44100)


def test_sample_reader_matches_in_memory_clip(tmp_path):
    root = str(tmp_path / 'clip0')
    make_clip(root, secs=3, flow=True)
    prep = F.img_prep_fcn()
    rd = F.SampleReader(root, return_video=True, img_prep=prep, return_flow=True, shuffle=False, random_rotations=False,
                        start_time=0., sample_duration=10.)
    assert len(rd.chunks_t) == 20 and rd.chunks_t[:3] == [0.5, 0.6, 0.7]
    # decoded truth
    audio = np.concatenate([F.load_wav(os.path.join(root, 'ambix', '%06d.wav' % i))[0] for i in range(3)], 0)
    video = np.stack([prep(F.imread(os.path.join(root, 'video', '%06d.jpg' % i))) for i in range(30)], 0)
    clip = ClipArrays(audio, video, start_time=0., sample_duration=10.)
    assert clip.chunks_t == rd.chunks_t
    lims = np.load(os.path.join(root, 'flow', 'flow_limits.npy'))
    for _ in range(20):
        a, b = rd.get(), clip.get()
        t = rd.cur_t
        assert a['ambix'].shape == (52799, 4) and np.array_equal(a['ambix'], b['ambix'])
        assert np.array_equal(a['ambix'], audio_window(audio, t, 1.0, 52799, 48000))
        assert a['video'].shape == (1, 224, 448, 3) and np.array_equal(a['video'], b['video'])
        fi = frame_index(t, 10)
        raw = F.imread(os.path.join(root, 'flow', '%06d.jpg' % fi)).astype(np.float32)
        mag = raw[..., 2] * (lims[fi, 1] - lims[fi, 0]) / 255. + lims[fi, 0]
        ang = raw[..., 0] * (2 * np.pi) / 255.
        assert np.allclose(a['flow'][0, ..., 2], mag) and np.allclose(a['flow'][0, ..., 0], mag * np.cos(ang), atol=1e-5)
        assert np.allclose(a['flow'][0, ..., 1], mag * np.sin(ang), atol=1e-5)
        # bit-exact against the reference's rounding points (feeder.py:147-160), restated with explicit casts: the float64 limits
        # make the two magnitude updates double-precision operations rounded to float32 once each; the angle is scaled in float32
        m32 = (raw[..., 2].astype(np.float64) * ((lims[fi, 1] - lims[fi, 0]) / 255.)).astype(np.float32)
        m32 = (m32.astype(np.float64) + lims[fi, 0]).astype(np.float32)
        a32 = raw[..., 0] * np.float32((2 * np.pi) / 255.)
        assert np.array_equal(a['flow'][0, ..., 2], m32)
        assert np.array_equal(a['flow'][0, ..., 0], m32 * np.cos(a32)) and np.array_equal(a['flow'][0, ..., 1], m32 * np.sin(a32))
    assert rd.get() is None and clip.get() is None


def test_reader_filters(tmp_path):
    root = str(tmp_path / 'clip1')
    make_clip(root, secs=4)
    base = dict(return_video=False, shuffle=False, random_rotations=False)
    assert len(F.SampleReader(root, **base).chunks_t) == 30
    assert len(F.SampleReader(root, skip_rate=10, **base).chunks_t) == 3                       # eval: every 10th window
    assert F.SampleReader(root, skip_silence_thr=0.25, **base).chunks_t[0] == pytest.approx(2.1)  # pow > thr
    assert F.SampleReader(root, start_time=1.0, sample_duration=1.0, **base).chunks_t == pytest.approx([1.0 + 0.1 * i for i in range(10)])
    parts = [F.SampleReader(root, num_threads=3, thread_id=i, **base).chunks_t for i in range(3)]
    assert sum(parts, []) == F.SampleReader(root, **base).chunks_t
    rot = F.AudioReader(os.path.join(root, 'ambix'), 48000).get(1.0, 100, rotation=np.pi / 2)
    ref = F.AudioReader(os.path.join(root, 'ambix'), 48000).get(1.0, 100)
    assert np.allclose(rot[:, 0], ref[:, 0]) and np.allclose(rot[:, 1], ref[:, 3]) and np.allclose(rot[:, 3], -ref[:, 1])


def test_batch_prefetcher_preserves_order():
    gen = ({'i': i, 'x': np.full((2, 2), i, np.float32)} for i in range(7))
    got = [b['i'] for b in F.BatchPrefetcher(gen, depth=2)]
    assert got == list(range(7))


def test_prefetcher_reraises_producer_failures_in_order():
    """A reader failure must surface in the consumer where it happened, not end the stream silently (a truncated deploy)."""
    def gen():
        yield {'i': 0}
        yield {'i': 1}
        raise IOError('frame 000002.jpg is missing')
    it = iter(F.BatchPrefetcher(gen(), depth=1))
    assert next(it)['i'] == 0 and next(it)['i'] == 1
    with pytest.raises(IOError, match='000002'):
        next(it)
    # a failing FIRST batch raises the reader's error too
    with pytest.raises(ZeroDivisionError):
        list(F.BatchPrefetcher((1 // 0 for _ in range(1)), depth=1))


def test_prefetcher_close_unblocks_the_producer():
    p = F.BatchPrefetcher(({'i': i} for i in range(100)), depth=1)
    assert next(iter(p))['i'] == 0
    p.close()
    assert not p.thread.is_alive()


def test_window_table_matches_the_scalar_transcription():
    """The vectorised window table against the scalar restatement of feeder.py:64-90,121 in deploy.py, over the deploy
    times (t = chunks_t[i] - (chunks_t[0] - start), float64) and a grid of awkward ones."""
    audio = rng(3).normal(size=(6 * 48000, 1))
    times = [(0.5 + i / 10.) - 0.5 for i in range(50)] + [0.5 + i / 10. for i in range(45)] + [1.2, 2.3000000000000003, 4.999999, 5.45]
    tab = F.window_table(times, 1.0, 52799, 48000, audio.shape[0], 10)
    for k, t in enumerate(times):
        ref = audio_window(audio, t, 1.0, 52799, 48000)
        got = np.zeros((52799, 1))
        n = int(tab.read_count[k])
        got[int(tab.lead[k]):int(tab.lead[k]) + n] = audio[int(tab.read_from[k]):int(tab.read_from[k]) + n]
        assert ref.shape == got.shape and np.array_equal(ref, got), t
        assert int(tab.frame[k]) == frame_index(t, 10)
        assert int(tab.lead[k]) + n + int(tab.trail[k]) == 52799


# ------------------------------------------------------------------------------------------------------------------------
# training augmentation: random yaw rotation (feeder.py:92-101 ambisonics, :125-129 frames)
# ------------------------------------------------------------------------------------------------------------------------
def _ref_rotate_ambix(chunk, rotation):
    """feeder.py:93-101 restated: chunk [n, 4] in ACN order (W, Y, Z, X)."""
    assert -np.pi <= rotation < np.pi
    c, s = np.cos(rotation), np.sin(rotation)
    rot_mtx = np.array([[1, 0, 0, 0],       # W' = W
                        [0, c, 0, s],       # Y' = X sin + Y cos
                        [0, 0, 1, 0],       # Z' = Z
                        [0, -s, 0, c]])     # X' = X cos - Y sin
    return np.dot(chunk, rot_mtx.T)


def _ref_roll_frames(chunk, rotation, width):
    """feeder.py:127-129: chunk [t, H, W, 3]; int() truncates towards zero, so +theta and -theta roll by opposite amounts."""
    return np.roll(chunk, -int(rotation / (2. * np.pi) * width), axis=2)


@pytest.mark.parametrize('rotation', [0.3, -0.3, -np.pi, np.pi - 1e-9, 1.0, -2.5, 0.0])
def test_rotation_augmentation_matches_the_reference_formulas(tmp_path, rotation):
    root = str(tmp_path / 'clipr')
    make_clip(root, secs=3, flow=True, seed=5)
    prep = F.img_prep_fcn()
    rd = F.SampleReader(root, return_video=True, img_prep=prep, return_flow=True, shuffle=False, random_rotations=False,
                        start_time=0., sample_duration=10.)
    t = rd.chunks_t[7]
    plain = rd.sample_at(t)
    rot = rd.sample_at(t, rotation)
    # ambisonics: W and Z untouched, (Y, X) rotated as a vector about the vertical axis
    want = _ref_rotate_ambix(plain['ambix'], rotation)
    assert np.allclose(rot['ambix'], want, rtol=0, atol=1e-12)
    assert np.array_equal(rot['ambix'][:, 0], plain['ambix'][:, 0]) and np.array_equal(rot['ambix'][:, 2], plain['ambix'][:, 2])
    c, s = np.cos(rotation), np.sin(rotation)
    assert np.allclose(rot['ambix'][:, 1], plain['ambix'][:, 3] * s + plain['ambix'][:, 1] * c, atol=1e-12)
    assert np.allclose(rot['ambix'][:, 3], plain['ambix'][:, 3] * c - plain['ambix'][:, 1] * s, atol=1e-12)
    assert np.allclose(rot['ambix'][:, 1] ** 2 + rot['ambix'][:, 3] ** 2, plain['ambix'][:, 1] ** 2 + plain['ambix'][:, 3] ** 2, atol=1e-12)
    # frames: equirectangular yaw = horizontal roll by -int(rotation / 2pi * W) pixels, video and flow alike (flow is rolled as
    # the raw jpg, then decoded: FlowReader.get_by_index hands the rotation to its VideoReader, feeder.py:148)
    assert np.array_equal(rot['video'], _ref_roll_frames(plain['video'], rotation, 448))
    assert np.array_equal(rot['flow'], _ref_roll_frames(plain['flow'], rotation, 448))
    if abs(rotation) > 0.05:
        assert not np.array_equal(rot['video'], plain['video'])
    # the reference-named readers take the same argument
    a = F.AudioReader(os.path.join(root, 'ambix'), 48000).get(1.0, 100, rotation=rotation)
    b = F.AudioReader(os.path.join(root, 'ambix'), 48000).get(1.0, 100)
    assert np.allclose(a, _ref_rotate_ambix(b, rotation), atol=1e-12)


def test_rotation_out_of_range_is_rejected(tmp_path):
    root = str(tmp_path / 'clipq')
    make_clip(root, secs=2, video=False)
    rd = F.AudioReader(os.path.join(root, 'ambix'), 48000)
    for bad in (np.pi, 4.0, -3.5):                                  # feeder.py:94: assert -pi <= rotation < pi
        with pytest.raises(ValueError):
            rd.get(0.5, 10, rotation=bad)


def test_random_rotations_draw_one_angle_per_sample_for_sound_and_picture(tmp_path, monkeypatch):
    """SampleReader.get with random_rotations=True (feeder.py:248-252): ONE angle, uniform in [-pi, pi), rotates the sound field and
    rolls the frames of the same sample - the spatial correspondence the network learns from must survive the augmentation."""
    import random
    root = str(tmp_path / 'clips')
    make_clip(root, secs=3, seed=9)
    prep = F.img_prep_fcn()
    kw = dict(return_video=True, img_prep=prep, shuffle=False, start_time=0., sample_duration=10.)
    rd = F.SampleReader(root, random_rotations=True, **kw)
    plain = F.SampleReader(root, random_rotations=False, **kw)
    draws = iter([0.0, 0.25, 0.75, 0.999999])
    angles = []

    def fake_random():
        u = next(draws)
        angles.append(u * 2 * np.pi - np.pi)
        return u
    monkeypatch.setattr(random, 'random', fake_random)
    for _ in range(4):
        a, b = rd.get(), plain.get()
        th = angles[-1]
        assert -np.pi <= th < np.pi
        assert np.allclose(a['ambix'], _ref_rotate_ambix(b['ambix'], th), atol=1e-12)
        assert np.array_equal(a['video'], _ref_roll_frames(b['video'], th, 448))
    assert len(angles) == 4                                          # exactly one draw per sample


def test_training_feeder_cuts_rotated_samples_into_batches(tmp_path):
    """train.folder_batches: the input is W of the rotated field, the target its (Y, Z, X) centre crop; Z is invariant under the yaw
    augmentation and W too, so they can be checked against the unrotated clip whatever angles were drawn."""
    from types import SimpleNamespace
    from spatialaudiogen_amd.train import folder_batches, silence_threshold
    root = str(tmp_path / 'db')
    audio = {}
    for k in range(2):
        audio['c%d' % k] = make_clip(os.path.join(root, 'c%d' % k), secs=3, seed=20 + k)
    prm = SimpleNamespace(ambi_order=1, audio_rate=48000, video_rate=10, context=1.0, encoders=['audio', 'video'])
    it = folder_batches(root, ['c0', 'c1'], prm, batch=4, seed=3, subset_fn='meta/subsets/REC-Street.train.1.lst')
    a, v, f, tgt, mask = next(it)
    assert a.shape == (4, 52799, 1) and v.shape == (4, 1, 224, 448, 3) and f is None and tgt.shape == (4, 4800, 3) and mask.shape == (4, 4)
    # frames stay as decoded: Trainer.forward_backward dispatches uint8 frames to sagen_train_step_u8 (pixel normalisation on the device)
    assert v.dtype == np.uint8 and a.dtype == np.float32 and tgt.dtype == np.float32
    # every window is a window of one of the two clips: W and Z (not rotated) identify it
    for i in range(4):
        w_centre = a[i, 24000:28800, 0]
        found = False
        for cid, x in audio.items():
            pcm = np.round(x * 32767.) / 32768.                      # what the wav round trip makes of the clip
            for t10 in range(5, 25):
                win = audio_window(pcm, 0.5 + (t10 - 5) / 10., 1.0, 52799, 48000)[24000:28800]   # (incl. the reader's truncation quirk)
                if np.allclose(win[:, 0], w_centre, atol=1e-6):
                    assert np.allclose(win[:, 2], tgt[i, :, 1], atol=1e-6)                       # Z = target channel 1 of (Y, Z, X)
                    assert np.allclose((win[:, [1, 3]] ** 2).sum(1), tgt[i, :, 0] ** 2 + tgt[i, :, 2] ** 2, atol=1e-5)   # rotated as a vector
                    found = True
        assert found, 'window %d not found in any clip' % i
    # the silence threshold follows the SUBSET FILE name (feeder.py:310), not the dataset folder
    assert silence_threshold('meta/subsets/REC-Street.train.1.lst', 'data/frames') == 0.01
    assert silence_threshold('meta/subsets/YT-All.train.1.lst', 'data/frames') == 0.2
    assert silence_threshold(None, 'data/frames') == 0.2
