"""W2XYZ.deploy (deploy.py:90-152) on the HIP path vs the oracle driven by the same window table:
groups of 10, zero-padded last group, [W | Y Z X] assembly."""
import numpy as np
import pytest

from oracle import np_oracle as O
from spatialaudiogen_amd.weights import variable_specs, init_weights
from util import rms, rng, ensure_lib, oracle_window

pytestmark = pytest.mark.gpu


class Params(object):
    ambi_order, audio_rate, video_rate, context, sample_dur = 1, 48000, 10, 1.0, 0.1
    separation, num_sep_tracks, fft_window = 'unet_mask', 32, 0.025
    context_units, freq_mask_units, loc_units = [64, 128, 128], [], [512, 512]

    def __init__(self, encoders):
        self.encoders = encoders


# (['audio'], 12, 10.) = BASELINE configs[0] / SURVEY 8(d)-1: a 12 s clip deployed for 10 s = 95 windows, 10 groups, the last 5 real + 5 zero
@pytest.mark.parametrize('encoders,secs,duration', [(['audio'], 4, 10.), (['audio', 'video'], 3, 1.25), (['audio'], 12, 10.)])
def test_deploy_matches_oracle(encoders, secs, duration):
    import torch
    assert torch.cuda.is_available()
    ensure_lib()
    from spatialaudiogen_amd.deploy import W2XYZ, ClipArrays, audio_window, frame_index
    r = rng(secs)
    audio = (0.3 * r.normal(size=(secs * 48000, 4))).astype(np.float32)              # W,Y,Z,X ground truth clip
    video = (r.integers(0, 256, size=(secs * 10, 224, 448, 3)) / 255. - 0.5).astype(np.float32) if 'video' in encoders else None
    P = init_weights(variable_specs(encoders), seed=4, mode='test')
    model = W2XYZ(params=Params(encoders), variables=P)
    clip = ClipArrays(audio, video)
    got = model.deploy(clip, 0., duration)

    rows = O.deploy_window_table(O.audio_pow_times(secs), 0., duration)
    assert got.shape == (len(rows) * 4800, 4)
    if secs == 12:
        assert len(rows) == 95 and got.shape == (456000, 4)
    orc = O.SptAudioGenOracle(encoders=encoders)
    ref = []
    for g in range(0, len(rows), 10):
        grp = rows[g:g + 10]
        a = np.zeros((10, 52799, 1)); v = np.zeros((10, 1, 224, 448, 3)) if video is not None else None
        for i, row in enumerate(grp):
            fi = row[3]
            a[i, :, 0] = oracle_window(audio, row)[:, 0]
            if v is not None:
                v[i, 0] = video[fi]
        y = orc.inference_ops(a, P, video=v)
        ref.append(np.concatenate([a[:len(grp), 24000:28800, :1], y[:len(grp)]], 2).reshape(-1, 4))
    ref = np.concatenate(ref, 0)
    assert np.array_equal(got[:, 0], ref[:, 0].astype(np.float32))                   # W is a bit-exact copy of the mono crop
    err = rms(got[:, 1:] - ref[:, 1:])
    assert err <= 1e-4 and err <= 1e-3 * rms(ref[:, 1:])


@pytest.mark.parametrize('encoders,secs,groups', [(['audio'], 12, 3), (['audio', 'video'], 5, 2)])
def test_grouped_deploy_is_bit_identical_to_one_batch_per_call(encoders, secs, groups):
    """W2XYZ.groups > 1 (round 6): consecutive batches of 10 windows as one grouped forward call - own batch-norm statistics per batch,
    so the wav must not change by a bit; the zero-padded partial batch and the batches that do not fill a group run singly."""
    import torch
    assert torch.cuda.is_available()
    ensure_lib()
    from spatialaudiogen_amd.deploy import W2XYZ, ClipArrays
    r = rng(secs + 40)
    audio = (0.3 * r.normal(size=(secs * 48000, 4))).astype(np.float32)
    video = r.integers(0, 256, size=(secs * 10, 224, 448, 3)).astype(np.uint8) if 'video' in encoders else None      # decoded frames
    P = init_weights(variable_specs(encoders), seed=5, mode='test')
    model = W2XYZ(params=Params(encoders), variables=P)
    want = model.deploy(ClipArrays(audio, video), 0., 10.)
    model.groups = groups
    got = model.deploy(ClipArrays(audio, video), 0., 10.)
    assert got.shape == want.shape and got.shape[0] >= 35 * 4800
    assert np.array_equal(got, want)
    assert model._mg.groups == groups and model._mg._ctx            # the grouped contexts really ran


def test_deploy_cli_from_disk(tmp_path):
    """model_dir (train-params.txt + a TF tensor bundle ASSEMBLED BY HAND, tests/bundle_by_hand.py - not by the product's writer) and a
    clip folder in the scraping/preprocess.py layout, through the deploy CLI: checkpoint reader -> feeder -> HIP path -> wav, against
    the ORACLE driven by the same window table (PCM16-quantised)."""
    import torch
    assert torch.cuda.is_available()
    ensure_lib()
    import os
    import bundle_by_hand as bh
    from test_feeder import make_clip
    from spatialaudiogen_amd import feeder as F
    from spatialaudiogen_amd.deploy import audio_window, main
    enc = ['audio', 'video']
    model_dir = tmp_path / 'model'; model_dir.mkdir()
    P = init_weights(variable_specs(enc), seed=8, mode='test')
    extra = dict(P); extra['step'] = np.array(150000, np.int64)
    for k in list(P)[:6]:                                        # optimiser slots the loader has to skip (deploy.py:79, eval.py:105)
        extra[k + '/Adam'] = np.zeros_like(P[k]); extra[k + '/Adam_1'] = np.zeros_like(P[k])
    bh.write_bundle(str(model_dir / 'model.ckpt-150000'), extra)
    (model_dir / 'train-params.txt').write_text(
        "encoders: ['audio', 'video']\nseparation: unet_mask\nambi_order: 1\naudio_rate: 48000\nvideo_rate: 10\n"
        "context: 1.0\nsample_dur: 0.1\nnum_sep_tracks: 32\nloc_units: [512, 512]\nfft_window: 0.025\n"
        "context_units: [64, 128, 128]\nfreq_mask_units: []\nlr: 0.0001\nbatch_size: 32\n")
    clip_dir = str(tmp_path / 'clip'); make_clip(clip_dir, secs=3)
    out_fn = str(tmp_path / 'out.wav')
    main([str(model_dir), clip_dir, '--deploy_duration', '1.2', '--output_fn', out_fn])
    wav, rate = F.load_wav(out_fn)
    assert rate == 48000 and wav.shape == (7 * 4800, 4)          # t in {0.5..1.1} < 0 + 1.2  (feeder.py:230-231)

    prep = F.img_prep_fcn()
    audio = np.concatenate([F.load_wav(os.path.join(clip_dir, 'ambix', '%06d.wav' % i))[0] for i in range(3)], 0)
    video = np.stack([prep(F.imread(os.path.join(clip_dir, 'video', '%06d.jpg' % i))) for i in range(30)], 0)
    rows = O.deploy_window_table(O.audio_pow_times(3), 0., 1.2)
    assert len(rows) == 7
    a = np.zeros((10, 52799, 1)); v = np.zeros((10, 1, 224, 448, 3))             # one group of 10: 7 real windows + 3 zero windows
    for i, row in enumerate(rows):
        fi = row[3]
        a[i, :, 0] = oracle_window(audio, row)[:, 0]
        v[i, 0] = video[fi]
    y = O.SptAudioGenOracle(encoders=enc).inference_ops(a, P, video=v)
    ref = np.concatenate([a[:7, 24000:28800, :1], y[:7]], 2).reshape(-1, 4)
    assert ref.shape == wav.shape
    assert np.abs(wav - np.clip(ref, -1, 1)).max() <= 1e-4 + 2.0 / 32768


def test_evaluate_driver_matches_oracle(tmp_path):
    """evaluate.py (eval.py's on-graph loop): clip folders -> every 10th window -> batches of 16 -> metrics file + means."""
    import torch, os
    assert torch.cuda.is_available()
    ensure_lib()
    from test_feeder import make_clip
    from spatialaudiogen_amd import feeder as F
    from spatialaudiogen_amd.evaluate import evaluate, METRIC_KEYS
    from spatialaudiogen_amd.deploy import audio_window
    enc = ['audio']
    P = init_weights(variable_specs(enc), seed=9, mode='test')
    db = tmp_path / 'db'; db.mkdir()
    model_dir = tmp_path / 'model'; model_dir.mkdir()
    clips = {}
    for i, name in enumerate(['clipA', 'clipB']):
        make_clip(str(db / name), secs=3, seed=20 + i)
        clips[name] = np.concatenate([F.load_wav(os.path.join(str(db / name), 'ambix', '%06d.wav' % k))[0] for k in range(3)], 0)
    (tmp_path / 'layouts.txt').write_text('clipA WXYZ\nclipB WXY\n')
    means, count = evaluate(str(model_dir), str(db), None, str(tmp_path / 'layouts.txt'), variables=P, params=Params(enc),
                            partial_batch='pad', power_maps=True)
    assert count == 4                                                # 20 windows per clip, every 10th -> 2 per clip
    lines = open(str(model_dir / 'eval-detailed.txt')).read().splitlines()
    assert lines[0] == 'SampleID | ' + ' '.join(METRIC_KEYS) and len(lines) == 5 and lines[1].startswith('clipA 0.5 |')

    # oracle: same 4 windows, zero-padded to the batch of 16
    amb = np.zeros((16, 52799, 4)); masks = np.ones((16, 4))
    k = 0
    for name in ('clipA', 'clipB'):
        for t in (0.5, 1.5):
            amb[k] = audio_window(clips[name], t, 1.0, 52799, 48000)
            if name == 'clipB':
                masks[k] = [1, 1, 0, 1]
            k += 1
    pred = O.SptAudioGenOracle(encoders=enc).inference_ops(amb[:, :, :1], P)
    target = amb[:, 24000:28800, 1:]
    _, stft_ps, lsd_ps, mse_ps, snr_ps = O.evaluation_ops(pred, target, masks[:, 1:])
    # eval.py:155-171 logs the raw per-sample values (the x5e3 / x100 scalings exist only in the on-graph summaries)
    ref = {'mse/avg': np.mean(mse_ps[:4]), 'stft/avg': np.mean(stft_ps[:4]), 'lsd/avg': np.mean(lsd_ps[:4]),
           'snr/avg': np.mean(snr_ps[:4]), 'stft/Z': np.mean(stft_ps[:4, 1]), 'amplitude/gt': np.mean(np.abs(target[:4]).max(axis=(1, 2)))}
    for key, v in ref.items():
        assert abs(means[key] - v) <= 2e-3 * max(1e-3, abs(v)) + 1e-6, (key, means[key], v)
    # per-sample directional RMS maps of the masked WYZX prediction / ground truth (inputs of eval.py:188-191's EMD), 30 degree mesh
    maps = evaluate.last_maps
    assert len(maps) == 2 and maps[0].shape == (4, 7, 12) and maps[1].shape == (4, 7, 12)
    for k in range(4):
        mono = amb[k, 24000:28800, :1]
        for got, x in ((maps[0][k], pred[k]), (maps[1][k], target[k])):
            wyzx = np.concatenate([mono, x], 1) * masks[k][None, :]
            assert np.abs(got - O.power_map(wyzx, 30.0)).max() <= 1e-4 * max(1.0, np.abs(got).max())


def _write_model_dir(model_dir, P, encoders):
    import os
    np.savez(os.path.join(model_dir, 'variables.npz'), **P)
    with open(os.path.join(model_dir, 'train-params.txt'), 'w') as f:
        f.write("encoders: [%s]\nseparation: unet_mask\nambi_order: 1\naudio_rate: 48000\nvideo_rate: 10\ncontext: 1.0\n"
                "num_sep_tracks: 32\nloc_units: [512, 512]\n" % ', '.join("'%s'" % e for e in encoders))


def _run_eval_cli(root, model_dir, db, world, port, extra=()):
    """python -m spatialaudiogen_amd.evaluate under `world` ranks that share this one GPU (gloo for the collective)."""
    import os, subprocess, sys
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                   SAGEN_DIST_BACKEND='gloo')
        procs.append(subprocess.Popen([sys.executable, '-m', 'spatialaudiogen_amd.evaluate', model_dir, db, '--overwrite'] + list(extra),
                                      env=env, cwd=root, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=900)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]
    return outs[0], open(os.path.join(model_dir, 'eval-detailed.txt')).read()


def test_evaluate_is_identical_for_one_and_two_ranks(tmp_path):
    """The real evaluate() (clips -> global window order -> whole batches of 16 dealt to ranks -> metrics -> ONE all-reduce +
    row gather) run as 1 rank and as 2 ranks on this GPU: every eval-detailed.txt row and every global mean must be
    bit-identical, because batch composition (hence batch-norm statistics) does not depend on the world size."""
    import os, socket
    ensure_lib()
    from test_feeder import make_clip
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    enc = ['audio']
    P = init_weights(variable_specs(enc), seed=9, mode='test')
    db = tmp_path / 'db'; db.mkdir()
    for i in range(19):                                  # 19 clips x 3 windows (every 10th of 30) = 57 = 3 batches + 9 dropped
        make_clip(str(db / ('clip%02d' % i)), secs=4, seed=100 + i, video=False)
    outs = {}
    for world in (1, 2):
        model_dir = tmp_path / ('model%d' % world); model_dir.mkdir()
        _write_model_dir(str(model_dir), P, enc)
        s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
        outs[world] = _run_eval_cli(root, str(model_dir), str(db), world, port)
    log1, rows1 = outs[1]
    log2, rows2 = outs[2]
    # ... and with --groups 2 (round 6: two batches per grouped forward call, own batch-norm statistics each; the third batch runs singly)
    model_dir = tmp_path / 'model_g'; model_dir.mkdir()
    _write_model_dir(str(model_dir), P, enc)
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    logg, rowsg = _run_eval_cli(root, str(model_dir), str(db), 1, port, extra=('--groups', '2'))
    assert rowsg == rows1, 'grouped evaluation changed the per-sample rows'
    assert [l for l in logg.splitlines() if l.startswith('EVAL | \t')] == [l for l in log1.splitlines() if l.startswith('EVAL | \t')]
    assert rows1 == rows2 and len(rows1.splitlines()) == 1 + 48
    means = lambda log: [l for l in log.splitlines() if l.startswith('EVAL | \t')]
    assert means(log1) == means(log2) and len(means(log1)) == 18
    assert 'EVAL | 48 samples' in log1 and 'EVAL | 48 samples' in log2
