"""Op-level parity: each HIP entry point (through the C ABI) against the numpy oracle (fp64).
Tolerance: relative RMS error <= 2e-5 per op (fp32 contraction vs fp64 truth); the end-to-end
1e-4 bar of BASELINE.md 4 is checked in test_gpu_model.py."""
import numpy as np
import pytest

from util import rel_rms_err, rng, ensure_lib
from oracle import np_oracle as O

pytestmark = pytest.mark.gpu
TOL = 2e-5


# SAGEN_LIB=<...>/libsagen_cpu.so: the same cases against the CPU twin of the op level (csrc_cpu/sagen_cpu.cpp) on host tensors, in a
# container without a GPU (tests/test_cpu_twin_ops.py runs them that way)
import os
TWIN = os.path.basename(os.environ.get('SAGEN_LIB', '')) == 'libsagen_cpu.so'


@pytest.fixture(scope='module')
def T():
    import torch
    if not TWIN:
        assert torch.cuda.is_available(), 'GPU tests need a ROCm device'
    ensure_lib()
    return torch


def dev(T, a):
    t = T.as_tensor(np.ascontiguousarray(a, dtype=np.float32))
    return t if TWIN else t.cuda()


def test_stft_mag_and_spec(T):
    from spatialaudiogen_amd import ops
    r = rng(0)
    audio = r.normal(size=(3, 52799)).astype(np.float32)
    mag, spec = ops.stft_mag(dev(T, audio), 46, 173, 89, 117)
    ref = O.stft(audio.astype(np.float64)[:, None, :], 1024, 4)[:, 0]          # [B,200,1024]
    assert rel_rms_err(mag.cpu().numpy(), np.abs(ref[:, 46:173])) < TOL
    s = spec.cpu().numpy()
    got = s[..., 0] + 1j * s[..., 1]
    assert rel_rms_err(np.stack([got.real, got.imag]), np.stack([ref[:, 89:117, :513].real, ref[:, 89:117, :513].imag])) < TOL


def test_stft_known_answer_sinusoid(T):
    """Bin-centred cosine A cos(2 pi k0 n / 1024): |X[k0]| = |X[1024-k0]| = 256 A, side bins 128 A."""
    from spatialaudiogen_amd import ops
    n = np.arange(52799)
    k0, A = 37, 0.7
    audio = (A * np.cos(2 * np.pi * k0 * n / 1024.))[None].astype(np.float32)
    mag, _ = ops.stft_mag(dev(T, audio), 0, 200)
    m = mag.cpu().numpy()[0]
    assert np.allclose(m[:, k0], 256 * A, rtol=1e-4) and np.allclose(m[:, 1024 - k0], 256 * A, rtol=1e-4)
    assert np.allclose(m[:, k0 + 1], 128 * A, rtol=1e-3) and np.allclose(m[:, k0 - 1], 128 * A, rtol=1e-3)
    assert np.abs(m[:, k0 + 3:500]).max() < 1e-2


CONV_CASES = [
    # name,               B, H,   W,    Cin, kh, kw, Cout, sh, sw, padding, bias, relu, prologue, stats
    ('audio_conv1',       2, 127, 1024, 1,   7, 16, 32,   4, 8, 'VALID', True,  True,  False, False),
    ('audio_conv2',       2, 31,  127,  32,  3, 7,  64,   2, 4, 'VALID', True,  True,  False, False),
    ('audio_conv5',       2, 5,   10,   256, 3, 5,  512,  1, 1, 'VALID', True,  True,  False, False),
    ('res_3x3_s1_bn',     2, 14,  28,   64,  3, 3,  64,   1, 1, 'SAME',  False, False, True,  True),
    ('res_3x3_s2',        2, 28,  56,   64,  3, 3,  128,  2, 2, 'SAME',  False, False, False, True),
    ('res_3x3_big',       3, 56,  112,  64,  3, 3,  64,   1, 1, 'SAME',  False, False, False, True),
    ('res_shortcut_1x1',  2, 28,  56,   64,  1, 1,  128,  2, 2, 'SAME',  False, False, False, False),
    ('res_conv1_7x7',     2, 64,  96,   3,   7, 7,  64,   2, 2, 'SAME',  False, False, False, True),
    ('odd_sizes_same',    1, 13,  9,    16,  3, 5,  40,   2, 1, 'SAME',  True,  False, False, False),
    ('wide_n',            1, 7,   14,   512, 3, 3,  512,  1, 1, 'SAME',  False, False, True,  True),
]


@pytest.mark.parametrize('case', CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_conv_2d(T, case):
    from spatialaudiogen_amd import ops
    name, B, H, W, Cin, kh, kw, Cout, sh, sw, padding, bias, relu, prologue, stats = case
    r = rng(hash(name) % 1000)
    x = r.normal(size=(B, H, W, Cin))
    w = r.normal(size=(kh, kw, Cin, Cout)) / np.sqrt(kh * kw * Cin)
    b = r.normal(size=(Cout,)) if bias else None
    sc = r.uniform(0.5, 1.5, size=(Cin,)) if prologue else None
    sf = r.normal(size=(Cin,)) if prologue else None
    xin = np.maximum(x * sc + sf, 0) if prologue else x
    ref = O.nn_convolution(xin, w, (sh, sw), padding)
    raw = ref.copy()
    if bias:
        ref = ref + b
    if relu:
        ref = np.maximum(ref, 0)
    out = ops.conv_2d(dev(T, x), dev(T, w), (sh, sw), padding, dev(T, b) if bias else None, relu,
                      dev(T, sc) if prologue else None, dev(T, sf) if prologue else None, return_bn_stats=stats)
    y, st = out if stats else (out, None)
    assert rel_rms_err(y.cpu().numpy(), ref) < TOL
    if stats:
        gamma, beta = r.uniform(0.5, 1.5, size=(Cout,)), r.normal(size=(Cout,))
        scale, shift = ops.bn_finalize(st, y.shape, dev(T, gamma), dev(T, beta))
        mu, var = raw.mean(axis=(0, 1, 2)), raw.var(axis=(0, 1, 2))
        rs = gamma / np.sqrt(var + 1e-3)
        assert rel_rms_err(scale.cpu().numpy(), rs) < TOL
        assert np.abs(shift.cpu().numpy() - (beta - mu * rs)).max() < 1e-4
        # BN + residual + ReLU epilogue kernel against contrib batch_norm (training mode)
        res = r.normal(size=raw.shape)
        z = ops.bn_apply_relu(y, scale, shift, dev(T, res))
        zref = np.maximum(O.batch_norm_train(raw, gamma, beta) + res, 0)
        assert rel_rms_err(z.cpu().numpy(), zref) < 5e-5


def test_maxpool_bn_relu(T):
    from spatialaudiogen_amd import ops
    r = rng(5)
    x = r.normal(size=(2, 112, 224, 64))
    sc, sf = r.uniform(-1.5, 1.5, size=(64,)), r.normal(size=(64,))       # negative scales too
    y = ops.maxpool3x3s2(dev(T, x), dev(T, sc), dev(T, sf))
    ref = O.max_pool_3x3_s2_same(np.maximum(x * sc + sf, 0))
    assert y.shape == (2, 56, 112, 64)
    assert rel_rms_err(y.cpu().numpy(), ref) < 1e-6
    y2 = ops.maxpool3x3s2(dev(T, x))
    assert np.array_equal(y2.cpu().numpy(), O.max_pool_3x3_s2_same(x.astype(np.float32)))
    yo = ops.maxpool3x3s2(dev(T, x[:, :13, :9, :8]))       # odd sizes: SAME pads (1,1)
    assert np.array_equal(yo.cpu().numpy(), O.max_pool_3x3_s2_same(x[:, :13, :9, :8].astype(np.float32)))


@pytest.mark.parametrize('M,K,N,relu', [(6, 3072, 1024, True), (2, 12544, 512, True), (96, 512, 99, False),
                                        (6, 1536, 512, True), (300, 512, 128, True)])
def test_fully_connected(T, M, K, N, relu):
    from spatialaudiogen_amd import ops
    r = rng(M + K + N)
    x, w, b = r.normal(size=(M, K)), r.normal(size=(K, N)) / np.sqrt(K), r.normal(size=(N,))
    y = ops.fully_connected(dev(T, x), dev(T, w), dev(T, b), relu)
    ref = x @ w + b
    if relu:
        ref = np.maximum(ref, 0)
    assert rel_rms_err(y.cpu().numpy(), ref) < TOL


DECONV_CASES = [  # B, H, W, Cin, kh, kw, Cout, sh, sw, relu    (the five decoder layers, model.py:302-308)
    (2, 3, 6, 1024, 3, 5, 256, 1, 1, True),
    (2, 5, 10, 512, 3, 5, 128, 1, 1, True),
    (2, 7, 14, 256, 3, 5, 64, 2, 2, True),
    (2, 15, 31, 128, 3, 7, 32, 2, 4, True),
    (1, 31, 127, 64, 7, 16, 32, 4, 8, False),
]


@pytest.mark.parametrize('case', DECONV_CASES, ids=['deconv5', 'deconv4', 'deconv3', 'deconv2', 'deconv1'])
def test_deconv_2d(T, case):
    from spatialaudiogen_amd import ops
    B, H, W, Cin, kh, kw, Cout, sh, sw, relu = case
    r = rng(Cin + kh)
    x = r.normal(size=(B, H, W, Cin))
    w = r.normal(size=(kh, kw, Cout, Cin)) / np.sqrt(kh * kw * Cin / (sh * sw))
    b = r.normal(size=(Cout,))
    ref = O.nn_conv2d_transpose(x, w, (sh, sw)) + b
    if relu:
        ref = np.maximum(ref, 0)
    y = ops.deconv_2d(dev(T, x), dev(T, w), (sh, sw), dev(T, b), relu)
    assert tuple(y.shape) == ref.shape
    assert rel_rms_err(y.cpu().numpy(), ref) < TOL


def test_deconv_delta_weight_pins_offsets(T):
    """One-hot kernel tap (p,q): output is the input scattered at (i*sh+p, j*sw+q) — pins the
    conv2d_transpose geometry independently of the oracle's implementation."""
    from spatialaudiogen_amd import ops
    r = rng(11)
    x = r.normal(size=(1, 4, 5, 4)).astype(np.float32)
    for (p, q) in [(0, 0), (2, 6), (1, 3)]:
        w = np.zeros((3, 7, 4, 4), np.float32)
        w[p, q] = np.eye(4)
        y = ops.deconv_2d(dev(T, x), dev(T, w), (2, 4)).cpu().numpy()
        ref = np.zeros((1, 9, 23, 4), np.float32)
        ref[:, p:p + 8:2, q:q + 20:4] = x
        assert np.array_equal(y, ref)


def _mask_ref(dmask, stft_c, coeffs):
    """model.py:326-347 + 421-434 on the oracle's ops."""
    m = O.sigmoid(dmask.transpose(0, 3, 1, 2))[:, None]                      # [B,1,K,28,1024]
    sep = stft_c[:, :, None] * m
    x_sep = O.istft(sep, 4)[:, :, :, 448:448 + 4800]                         # [B,1,K,4800]
    step = np.arange(4800) // 1600
    w, b = coeffs[..., :-1][:, step], coeffs[..., -1][:, step]              # [B,4800,3,K], [B,4800,3]
    return np.einsum('bnok,bkn->bno', w, x_sep[:, 0]) + b


@pytest.mark.parametrize('K', [32, 16])
def test_mask_istft_mix(T, K):
    from spatialaudiogen_amd import ops
    r = rng(K)
    B = 2
    audio = r.normal(size=(B, 1, 52799))
    stft_c = O.stft(audio, 1024, 4)[:, :, 89:117]                             # [B,1,28,1024]
    dmask = 2.0 * r.normal(size=(B, 28, 1024, K))
    coeffs = r.normal(size=(B, 3, 3, K + 1)) / np.sqrt(K)
    ref = _mask_ref(dmask, stft_c, coeffs)
    spec = np.stack([stft_c[:, 0, :, :513].real, stft_c[:, 0, :, :513].imag], -1)
    y = ops.mask_istft_mix(dev(T, dmask), dev(T, spec), dev(T, coeffs))
    assert rel_rms_err(y.cpu().numpy(), ref) < TOL


def test_identity_mask_round_trip(T):
    """All-pass mask (sigmoid(+inf) = 1), w = e_0, b = 0: output = 0.5 * mono[24000:28800]
    (Hann hop-N/4 COLA sum 2, divided by 4 — SURVEY.md 9.3)."""
    from spatialaudiogen_amd import ops
    r = rng(3)
    audio = r.normal(size=(1, 52799)).astype(np.float32)
    _, spec = ops.stft_mag(dev(T, audio), 46, 173, 89, 117)
    dmask = np.full((1, 28, 1024, 32), 40.0, np.float32)
    coeffs = np.zeros((1, 3, 3, 33), np.float32)
    coeffs[:, :, 0, 0] = 1.0
    coeffs[:, :, 1, 5] = -2.0
    y = ops.mask_istft_mix(dev(T, dmask), spec, dev(T, coeffs)).cpu().numpy()
    assert np.abs(y[0, :, 0] - 0.5 * audio[0, 24000:28800]).max() < 2e-6
    assert np.abs(y[0, :, 1] + 1.0 * audio[0, 24000:28800]).max() < 4e-6
    assert np.abs(y[0, :, 2]).max() == 0.0


@pytest.mark.parametrize('res', [30.0, 5.0])
def test_power_map(T, res):
    from spatialaudiogen_amd import ops
    r = rng(9)
    ambi = r.normal(size=(4800 * 3, 4)) * np.array([1.0, 0.3, 0.1, 0.6])
    phi, nu = O.spherical_mesh(res)
    Y = O.sh_matrix_order1(phi, nu)
    rms = ops.power_map(dev(T, ambi), dev(T, Y)).cpu().numpy().reshape(phi.shape)
    ref = O.power_map(ambi, res)
    assert rel_rms_err(np.flipud(rms), ref) < 1e-5


def test_power_map_per_frame_batched(T):
    """One RMS map per 0.1 s frame (SphericalAmbisonicsVisualizer.loop_frames, distance.py:41-59): the batched kernel
    against the oracle frame by frame, with the product-side mesh / harmonics."""
    from spatialaudiogen_amd import ops
    from spatialaudiogen_amd.ambisonics import sh_matrix, mesh_shape
    r = rng(19)
    nfr = 13
    ambi = r.normal(size=(nfr, 4800, 4)) * np.array([1.0, 0.3, 0.1, 0.6]) * r.uniform(0.1, 2.0, size=(nfr, 1, 1))
    ambi[5] = 0.0                                                        # a silent frame
    for res in (30.0, 5.0):
        rms = ops.power_map_batched(dev(T, ambi), dev(T, sh_matrix(res))).cpu().numpy()
        for f in range(nfr):
            ref = O.power_map(ambi[f], res)
            got = np.flipud(rms[f].reshape(mesh_shape(res)))
            assert np.abs(got - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max()), (res, f)


def test_assemble_wyzx_is_bit_exact(T):
    from spatialaudiogen_amd import ops
    r = rng(4)
    audio = r.normal(size=(3, 52799)).astype(np.float32)
    yzx = r.normal(size=(3, 4800, 3)).astype(np.float32)
    out = ops.assemble_wyzx(dev(T, audio), dev(T, yzx)).cpu().numpy()
    assert np.array_equal(out[:, :, 0], audio[:, 24000:28800])           # W = mono crop (deploy.py:149)
    assert np.array_equal(out[:, :, 1:], yzx)                            # then Y, Z, X (ACN 1,2,3)


@pytest.mark.parametrize('B', [3, 16])
def test_evaluation_metrics(T, B):
    """SptAudioGen.evaluation_ops (model.py:110-154) on device vs the oracle's FFT-based restatement, including the
    channel masks of the WXY-only videos (feeder.py:312-314)."""
    from spatialaudiogen_amd.model import SptAudioGen
    r = rng(B)
    gt = 0.3 * r.normal(size=(B, 4800, 3))
    n = np.arange(4800)[None, :, None]
    gt = gt + 0.2 * np.sin(2 * np.pi * r.uniform(200, 4000, size=(B, 1, 3)) * n / 48000.)
    pred = gt * r.uniform(0.5, 1.1, size=(B, 1, 3)) + 0.05 * r.normal(size=gt.shape)
    pred[0] = 0.0                                                    # silent prediction: exercises the EPS terms
    mask = np.ones((B, 3)); mask[1, 1] = 0.0; mask[2, 1] = 0.0       # WXY layout: no Z
    ref_m, ref_stft, ref_lsd, ref_mse, ref_snr = O.evaluation_ops(pred, gt, mask)
    net = SptAudioGen(1, encoders=['audio'], separation='unet_mask')
    m, stft_ps, lsd_ps, mse_ps, snr_ps = net.evaluation_ops(dev(T, pred), dev(T, gt), None, dev(T, mask))
    assert rel_rms_err(stft_ps.cpu().numpy(), ref_stft) < 1e-5
    assert rel_rms_err(mse_ps.cpu().numpy(), ref_mse) < 1e-5
    assert np.abs(snr_ps.cpu().numpy() - ref_snr).max() < 1e-3
    assert np.abs(lsd_ps.cpu().numpy() - ref_lsd).max() < 2e-3 * max(1.0, np.abs(ref_lsd).max())
    assert list(m.keys()) == list(ref_m.keys())
    for k in ref_m:
        assert abs(m[k] - ref_m[k]) <= 2e-3 * max(1.0, abs(ref_m[k])), (k, m[k], ref_m[k])


def _dw_cases():
    """Randomised 3x3 stride-1 SAME convolutions (the geometry igemm3dw_kernel takes over): odd / tiny images, ragged M
    tails, N off the tile sizes, multi-image batches (image-edge and batch-edge gap slots), with the fused BN prologue."""
    r = np.random.default_rng(20240917)
    cases = []
    for i in range(28):
        H = int(r.integers(2, 24)); W = int(r.integers(8, 40))
        B = int(r.integers(1, 6))
        Cin = int(r.choice([16, 32, 64, 128]))
        Cout = int(r.choice([8, 24, 32, 48, 64, 96, 128, 136]))
        cases.append(('dw%02d' % i, B, H, W, Cin, Cout, bool(i % 3 == 0), bool(i % 2 == 0)))
    cases += [('dw_w8', 3, 7, 8, 16, 64, True, True), ('dw_h2', 2, 2, 33, 32, 64, False, True),
              ('dw_one_row_tile', 1, 3, 131, 16, 32, True, False), ('dw_many_images', 19, 4, 9, 16, 16, False, True)]
    return cases


@pytest.mark.parametrize('case', _dw_cases(), ids=lambda c: c[0])
def test_conv3x3_stride1_randomised_geometry(T, case):
    from spatialaudiogen_amd import ops
    name, B, H, W, Cin, Cout, prologue, stats = case
    r = rng(sum(map(ord, name)))
    x = r.normal(size=(B, H, W, Cin))
    w = r.normal(size=(3, 3, Cin, Cout)) / np.sqrt(9 * Cin)
    sc = r.uniform(0.5, 1.5, size=(Cin,)) if prologue else None
    sf = r.normal(size=(Cin,)) if prologue else None
    xin = np.maximum(x * sc + sf, 0) if prologue else x
    ref = O.nn_convolution(xin, w, (1, 1), 'SAME')
    out = ops.conv_2d(dev(T, x), dev(T, w), (1, 1), 'SAME', None, False, dev(T, sc) if prologue else None,
                      dev(T, sf) if prologue else None, return_bn_stats=stats)
    y, st = out if stats else (out, None)
    assert rel_rms_err(y.cpu().numpy(), ref) < TOL
    if stats:
        gamma, beta = np.ones(Cout), np.zeros(Cout)
        scale, shift = ops.bn_finalize(st, y.shape, dev(T, gamma), dev(T, beta))
        rs = 1.0 / np.sqrt(ref.var(axis=(0, 1, 2)) + 1e-3)
        assert rel_rms_err(scale.cpu().numpy(), rs) < TOL
        assert rel_rms_err(shift.cpu().numpy(), -ref.mean(axis=(0, 1, 2)) * rs) < 50 * TOL + 1e-5


def test_every_dw_sharing_tile_on_odd_geometry(T):
    """Re-run the randomised 3x3 cases once per igemm3dw instantiation (SAGEN_FORCE_TILE is read when the library loads,
    hence one subprocess per tile; it applies to the cases with M > 128 and N >= 64, the others keep the heuristic)."""
    import os, subprocess, sys
    from spatialaudiogen_amd.model import SptAudioGen
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tiles = [i for i, nm in enumerate(SptAudioGen.tile_names()) if nm.startswith('igemm3dw_kernel')]
    assert len(tiles) >= 8
    for t in tiles:
        env = dict(os.environ); env['SAGEN_FORCE_TILE'] = str(t)
        r = subprocess.run([sys.executable, '-m', 'pytest', os.path.join(root, 'tests', 'test_gpu_ops.py'), '-m', 'gpu', '-q', '-x',
                            '-k', 'conv3x3_stride1_randomised'], env=env, cwd=root, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, 'tile %d (%s):\n%s' % (t, SptAudioGen.tile_names()[t], r.stdout[-2000:])


def _generic_cases():
    r = np.random.default_rng(777)
    cases = []
    for i in range(36):
        kh, kw = int(r.choice([1, 2, 3, 5, 7])), int(r.choice([1, 3, 4, 5, 7]))
        sh, sw = int(r.choice([1, 2, 3])), int(r.choice([1, 2, 4]))
        Cin = int(r.choice([4, 8, 16, 32, 64]))
        if kh * kw == 1:
            Cin = int(r.choice([4, 12, 20, 36, 64]))              # 1x1: any multiple of 4
        Cout = int(r.choice([4, 20, 32, 33, 64, 100, 128]))
        H = int(r.integers(kh, kh + 20)); W = int(r.integers(kw, kw + 28)); B = int(r.integers(1, 5))
        cases.append(('g%02d' % i, B, H, W, Cin, kh, kw, Cout, sh, sw, 'SAME' if i % 2 else 'VALID', bool(i % 3 == 0), bool(i % 4 == 1),
                      bool(i % 5 == 0)))
    return cases


@pytest.mark.parametrize('case', _generic_cases(), ids=lambda c: c[0])
def test_conv_2d_randomised_generic_geometry(T, case):
    """Kernel sizes / strides / paddings / channel counts off the model's own: per-chunk taps (Cin < 16), ragged K tails,
    N off the tile sizes, tiny M."""
    from spatialaudiogen_amd import ops
    name, B, H, W, Cin, kh, kw, Cout, sh, sw, padding, bias, relu, stats = case
    r = rng(sum(map(ord, name)) + 13)
    x = r.normal(size=(B, H, W, Cin))
    w = r.normal(size=(kh, kw, Cin, Cout)) / np.sqrt(kh * kw * Cin)
    b = r.normal(size=(Cout,)) if bias else None
    ref = O.nn_convolution(x, w, (sh, sw), padding)
    raw = ref.copy()
    if bias:
        ref = ref + b
    if relu:
        ref = np.maximum(ref, 0)
    out = ops.conv_2d(dev(T, x), dev(T, w), (sh, sw), padding, dev(T, b) if bias else None, relu, return_bn_stats=stats)
    y, st = out if stats else (out, None)
    assert y.shape == ref.shape
    assert rel_rms_err(y.cpu().numpy(), ref) < TOL
    if stats:
        scale, shift = ops.bn_finalize(st, y.shape, dev(T, np.ones(Cout)), dev(T, np.zeros(Cout)))
        rs = 1.0 / np.sqrt(raw.var(axis=(0, 1, 2)) + 1e-3)
        assert rel_rms_err(scale.cpu().numpy(), rs) < TOL


def _deconv_cases():
    r = np.random.default_rng(4242)
    cases = []
    for i in range(20):
        sh, sw = int(r.choice([1, 2, 3])), int(r.choice([1, 2, 4]))
        kh, kw = sh + int(r.integers(0, 4)), sw + int(r.integers(0, 5))          # kernel >= stride
        Cin = int(r.choice([4, 8, 16, 32, 64])); Cout = int(r.choice([4, 12, 32, 40, 64]))
        cases.append((int(r.integers(1, 4)), int(r.integers(1, 9)), int(r.integers(1, 14)), Cin, kh, kw, Cout, sh, sw, bool(i % 2)))
    return cases


@pytest.mark.parametrize('case', _deconv_cases(), ids=lambda c: 'b%d_%dx%d_c%d_k%dx%d_o%d_s%dx%d' % c[:9])
def test_deconv_2d_randomised_geometry(T, case):
    """conv2d_transpose as a stride-1 conv with a depth-to-space epilogue, over random kernel / stride / channel combinations."""
    test_deconv_2d(T, case)
