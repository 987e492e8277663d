"""The fp16x2 trunk arithmetic cannot fail silently ("fp16x2 has a failure mode the bf16x3 path did not have"):

* the activation planes carry a RIGOROUS range bound next to the statistical one (Samuelson: no sample lies further than sqrt(N - 1)
  standard deviations from the mean of its own batch statistics) - a > 1000-sigma outlier, which the statistical bound of round 4 would
  have clamped, goes through exactly;
* the hosts (deploy / evaluate / train) read the saturation counter and re-run / stop;
* adversarial batch-norm parameters through the deploy CLI still meet the bar against the oracle;
* the training feeder's uint8 frames reach the uint8 entry point of the training step."""
import os
import numpy as np
import pytest

from oracle import np_oracle as O
from spatialaudiogen_amd.weights import variable_specs, init_weights, synth_inputs
from util import rms, rel_rms_err, ensure_lib, oracle_window

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def T():
    import torch
    assert torch.cuda.is_available(), 'GPU tests need a ROCm device'
    ensure_lib()
    return torch


def test_an_outlier_beyond_a_thousand_sigma_goes_through_the_fp16x2_planes_exactly(T):
    """One stem channel made two-valued: its filter is a single tap, every frame is constant grey, ONE pixel of one frame is white.
    That channel's batch statistics then put the white pixel sqrt(N - 1) = 1097 standard deviations from the mean (N = 48 x 112 x 224) -
    137 times the 8-sigma statistical bound, beyond the 64..128 x headroom fp16 leaves above it: round 4's planes clamped it (and only
    a counter nobody read said so).  With the rigorous bound the planes' scale steps down, nothing is clamped, and the result agrees
    with the three-plane bf16 arithmetic (fp32's exponent range; itself held against the oracle at batch 32 elsewhere)."""
    from spatialaudiogen_amd.model import SptAudioGen
    enc = ['audio', 'video']
    B, j = 48, 5
    P = init_weights(variable_specs(enc), seed=5, mode='test')
    w = P['video_encoder/conv1/conv/weights'].copy()
    w[:, :, :, j] = 0.0
    w[3, 3, 0, j] = 1.0                                  # y0[.., j] = x[2i + 1, 2k + 1, 0]: no border effect, exactly two values
    P['video_encoder/conv1/conv/weights'] = w
    g = P['video_encoder/conv1/conv/bn/gamma'].copy(); g[j] = 100.0          # ... and the channel that decides the tensor's bound
    P['video_encoder/conv1/conv/bn/gamma'] = g
    inp = synth_inputs(B, enc, seed=3)
    video = np.full((B, 1, 224, 448, 3), 128, np.uint8)
    video[:, 0] += (np.random.Generator(np.random.PCG64(1)).integers(0, 2, size=(B, 224, 448, 3))).astype(np.uint8) * np.array([0, 1, 1], np.uint8)
    video[17, 0, 101, 201, 0] = 255                      # (channel 0 stays constant but for this pixel; channels 1, 2 carry a little noise)
    net = SptAudioGen(1, encoders=enc, separation='unet_mask')
    net.load_variables(P)
    net.set_option(B, 'u8_fast_stem', 0)                 # the general stem kernels: this test is about the planes behind the pool
    got = net.inference_ops(inp['audio'], video).cpu().numpy()
    trunk = net.intermediate(B, 'video_encoder/conv5_2').cpu().numpy()
    assert net.counter(B, 'fp16x2_saturations') == 0
    net.set_option(B, 'fp16x2', 0)
    other = net.inference_ops(inp['audio'], video).cpu().numpy()
    trunk3 = net.intermediate(B, 'video_encoder/conv5_2').cpu().numpy()
    assert np.isfinite(got).all() and np.isfinite(trunk).all()
    e_t, e_o = rel_rms_err(trunk, trunk3), rel_rms_err(got, other)
    print('\noutlier at 1097 sigma: conv5_2 fp16x2 vs bf16x3 %.3g, output %.3g' % (e_t, e_o))
    assert e_t < 1e-4 and e_o < 1e-4, (e_t, e_o)


def test_checked_inference_reruns_a_batch_whose_planes_clamped(T, monkeypatch):
    """SptAudioGen.inference_ops_checked (what deploy.py / evaluate.py call): a batch during which the device counted clamped plane
    elements is run again with the trunk on three bf16 planes; 'raise' refuses instead.  (Since round 5 the counter cannot move on finite
    inputs, so the count is injected here; the re-run itself is real and is checked by the kernels it launches.)"""
    from spatialaudiogen_amd.model import SptAudioGen, _Ctx
    enc = ['audio', 'video']
    B = 2
    P = init_weights(variable_specs(enc), seed=2, mode='test')
    inp = synth_inputs(B, enc, seed=8)
    ref = O.SptAudioGenOracle(encoders=enc).inference_ops(inp['audio'], P, video=inp['video'])
    net = SptAudioGen(1, encoders=enc, separation='unet_mask')
    net.load_variables(P)
    y0 = net.inference_ops_checked(inp['audio'], inp['video']).cpu().numpy()
    assert getattr(net, 'saturation_events', []) == []
    real = _Ctx.counter
    fake = {'n': 0}

    def counter(self, name):
        return real(self, name) + fake['n']
    monkeypatch.setattr(_Ctx, 'counter', counter)
    fake['n'] = 7                                        # "this batch clamped 7 elements"
    net.profile_enable(B, True)
    with pytest.warns(UserWarning, match='clamped 7 elements'):
        y1 = net.inference_ops_checked(inp['audio'], inp['video']).cpu().numpy()
    kernels = {k for k, layer, us, fl in net.profile_report(B)}      # the LAST forward = the re-run
    net.profile_enable(B, False)
    assert net.saturation_events == [(B, 7)]
    assert not any(k.startswith('conv3h') for k in kernels), kernels          # no fp16x2 kernel in the re-run
    assert any(k.startswith('conv3p_kernel') or k.startswith('igemm3') for k in kernels)
    for y in (y0, y1):
        err = rms(y - ref)
        assert err <= 1e-4 and err <= 1e-3 * rms(ref)
    y2 = net.inference_ops_checked(inp['audio'], inp['video']).cpu().numpy()        # the counter did not move again: fp16x2 is back on
    assert np.array_equal(y2, y0) and net.saturation_events == [(B, 7)]
    fake['n'] = 9
    with pytest.raises(FloatingPointError, match='2 elements clamped'):
        net.inference_ops_checked(inp['audio'], inp['video'], on_saturation='raise')


def _adversarial(P, seed=1):
    """Batch-norm parameters chosen against the plane scaling: in every trunk batch-norm a third of the channels with gamma ~ 0 and
    beta = 0 (dead next to loud ones), a few with |beta| >> 1 (the tensor's bound is set by an offset, not by the spread), and one outlier
    channel with gamma x 100."""
    r = np.random.Generator(np.random.PCG64(seed))
    P = dict(P)
    for k in list(P):
        if k.startswith('video_encoder/') and k.endswith('/bn/gamma') and '/conv1/' not in k:
            g, b = P[k].copy(), P[k.replace('gamma', 'beta')].copy()
            C = g.shape[0]
            idx = r.permutation(C)
            dead, loud, out = idx[:C // 3], idx[C // 3:C // 3 + 3], idx[C // 3 + 3]
            g[dead] *= 1e-4; b[dead] = 0.0
            b[loud] = r.choice([-50.0, 50.0], size=3)
            g[out] *= 100.0
            P[k], P[k.replace('gamma', 'beta')] = g.astype(np.float32), b.astype(np.float32)
    return P


def test_deploy_cli_with_adversarial_batch_norm_parameters(T, tmp_path):
    """The deploy CLI (checkpoint reader -> feeder with uint8 frames -> guarded HIP forward -> wav) on weights built against the fp16x2
    plane scaling, against the oracle on the same windows: 15 windows = one full group of 10 (uint8 frames, sagen_forward_u8) and one of
    5 + 5 zero windows (float frames: the padding is 0.0 after normalisation)."""
    import bundle_by_hand as bh
    from test_feeder import make_clip
    from spatialaudiogen_amd import feeder as F
    from spatialaudiogen_amd.deploy import audio_window, main
    enc = ['audio', 'video']
    model_dir = tmp_path / 'model'; model_dir.mkdir()
    P = _adversarial(init_weights(variable_specs(enc), seed=8, mode='test'))
    bh.write_bundle(str(model_dir / 'model.ckpt-1'), dict(P))
    (model_dir / 'train-params.txt').write_text(
        "encoders: ['audio', 'video']\nseparation: unet_mask\nambi_order: 1\naudio_rate: 48000\nvideo_rate: 10\n"
        "context: 1.0\nsample_dur: 0.1\nnum_sep_tracks: 32\nloc_units: [512, 512]\nfft_window: 0.025\n"
        "context_units: [64, 128, 128]\nfreq_mask_units: []\nlr: 0.0001\nbatch_size: 32\n")
    clip_dir = str(tmp_path / 'clip'); make_clip(clip_dir, secs=3)
    out_fn = str(tmp_path / 'out.wav')
    main([str(model_dir), clip_dir, '--deploy_duration', '2.0', '--output_fn', out_fn])
    wav, rate = F.load_wav(out_fn)
    assert rate == 48000 and wav.shape == (15 * 4800, 4)

    prep = F.img_prep_fcn()
    audio = np.concatenate([F.load_wav(os.path.join(clip_dir, 'ambix', '%06d.wav' % i))[0] for i in range(3)], 0)
    video = np.stack([prep(F.imread(os.path.join(clip_dir, 'video', '%06d.jpg' % i))) for i in range(30)], 0)
    rows = O.deploy_window_table(O.audio_pow_times(3), 0., 2.0)
    assert len(rows) == 15
    orc = O.SptAudioGenOracle(encoders=enc)
    ref = []
    for g0 in range(0, 15, 10):
        grp = rows[g0:g0 + 10]
        a = np.zeros((10, 52799, 1)); v = np.zeros((10, 1, 224, 448, 3))
        for i, row in enumerate(grp):
            fi = row[3]
            a[i, :, 0] = oracle_window(audio, row)[:, 0]
            v[i, 0] = video[fi]
        y = orc.inference_ops(a, P, video=v)
        ref.append(np.concatenate([a[:len(grp), 24000:28800, :1], y[:len(grp)]], 2).reshape(-1, 4))
    ref = np.concatenate(ref, 0)
    assert ref.shape == wav.shape
    err = np.abs(wav - np.clip(ref, -1, 1))
    print('\nadversarial deploy: max |err| %.3g (PCM16 step %.3g), rms %.3g, output rms %.3g' % (err.max(), 1 / 32768., rms(wav - np.clip(ref, -1, 1)), rms(ref[:, 1:])))
    assert err.max() <= 1e-4 + 2.0 / 32768


def test_adversarial_batch_norm_parameters_hold_the_bar_directly(T):
    """The same weights without the wav quantisation in the way: fp32 output against the fp64 oracle at the usual bars, the trunk output
    relative, no clamped element, and the three-plane bf16 arithmetic as the yardstick."""
    from spatialaudiogen_amd.model import SptAudioGen
    enc = ['audio', 'video']
    B = 4
    P = _adversarial(init_weights(variable_specs(enc), seed=8, mode='test'))
    inp = synth_inputs(B, enc, seed=19)
    orc = O.SptAudioGenOracle(encoders=enc)
    ref = orc.inference_ops(inp['audio'], P, video=inp['video'])
    net = SptAudioGen(1, encoders=enc, separation='unet_mask')
    net.load_variables(P)
    got = net.inference_ops_checked(inp['audio'], inp['video'], on_saturation='raise').cpu().numpy()
    trunk = net.intermediate(B, 'video_encoder/conv5_2').cpu().numpy()
    net.set_option(B, 'fp16x2', 0)
    other = net.inference_ops(inp['audio'], inp['video']).cpu().numpy()
    trunk3 = net.intermediate(B, 'video_encoder/conv5_2').cpu().numpy()
    tr = orc.ends['video_encoder/conv5_2']
    e2, e3, o2, o3 = rel_rms_err(trunk, tr), rel_rms_err(trunk3, tr), rms(got - ref), rms(other - ref)
    print('\nadversarial: conv5_2 rel err fp16x2 %.3g / bf16x3 %.3g; output rms err %.3g / %.3g (output rms %.3g)' % (e2, e3, o2, o3, rms(ref)))
    assert net.counter(B, 'fp16x2_saturations') == 0
    assert o2 <= 1e-4 and o2 <= 1e-3 * rms(ref)
    assert e2 < 1e-4 and e2 <= 1.5 * e3 + 1e-7


def test_training_feeder_batches_reach_the_uint8_training_entry(T, tmp_path):
    """train.folder_batches hands the decoded uint8 frames on; Trainer.forward_backward must take sagen_train_step_u8 with them (the path
    bench.py --config train measures) and agree with the float-frame step on the same batch."""
    from types import SimpleNamespace
    from test_feeder import make_clip
    from spatialaudiogen_amd.model import SptAudioGen
    from spatialaudiogen_amd.train import Trainer, folder_batches
    from spatialaudiogen_amd.feeder import frames_to_float
    enc = ['audio', 'video']
    root = str(tmp_path / 'db')
    for k in range(2):
        make_clip(os.path.join(root, 'c%d' % k), secs=3, seed=30 + k)
    prm = SimpleNamespace(ambi_order=1, audio_rate=48000, video_rate=10, context=1.0, encoders=enc)
    a, v, f, tgt, mask = next(folder_batches(root, ['c0', 'c1'], prm, batch=2, seed=1, subset_fn='REC-Street'))
    assert v.dtype == np.uint8
    P = init_weights(variable_specs(enc), seed=6, mode='test')
    net = SptAudioGen(1, encoders=enc, separation='unet_mask')
    net.load_variables(P)
    tr = Trainer(net, batch=2)
    loss8 = float(tr.forward_backward(a, v, None, tgt, mask))
    assert tr.last_entry == 'sagen_train_step_u8'
    g8 = tr.grad('video_encoder/conv2_1/conv_1/weights').cpu().numpy().copy()
    lossf = float(tr.forward_backward(a, frames_to_float(v), None, tgt, mask))
    assert tr.last_entry == 'sagen_train_step'
    gf = tr.grad('video_encoder/conv2_1/conv_1/weights').cpu().numpy()
    from test_gpu_backward import FREE_RUNNING_BAR
    # (the uint8 stem convolves the exact (u - 127.5) / 255, the float one the rounded float32 x: the loss agrees to fp32 rounding, a trunk
    #  gradient at the level the free-running ReLU switching of a batch of 2 allows - tests/test_gpu_backward.py holds the same bars)
    assert np.isfinite(loss8) and abs(loss8 - lossf) <= 2e-5 * abs(lossf)
    assert rel_rms_err(g8, gf) < FREE_RUNNING_BAR
