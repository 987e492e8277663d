"""End-to-end parity of the HIP path (through sagen_forward) against the fp64 numpy oracle on the
same seeded inputs and weights.  Bar: ambisonic output RMS error <= 1e-4 absolute
and <= 1e-3 relative to the output RMS; channel order [Y,Z,X] and sample indexing exact."""
import numpy as np
import pytest

from util import rms, rel_rms_err, ensure_lib
from oracle.np_oracle import SptAudioGenOracle
from spatialaudiogen_amd.weights import variable_specs, init_weights, synth_inputs

pytestmark = pytest.mark.gpu

ABS_TOL, REL_TOL = 1e-4, 1e-3


@pytest.fixture(scope='module')
def T():
    import torch
    assert torch.cuda.is_available(), 'GPU tests need a ROCm device'
    ensure_lib()
    return torch


def run_pair(T, encoders, separation='unet_mask', batch=2, seed=0, nsep=32, loc_units=(512, 512)):
    from spatialaudiogen_amd.model import SptAudioGen, SptAudioGenParams
    specs = variable_specs(encoders, separation, nsep, loc_units)
    P = init_weights(specs, seed=seed, mode='test')
    inp = synth_inputs(batch, encoders, seed=1234 + seed)
    orc = SptAudioGenOracle(encoders=encoders, separation=separation, sep_num_tracks=nsep, loc_fc_units=loc_units)
    ref = orc.inference_ops(inp['audio'], P, video=inp.get('video'), flow=inp.get('flow'))
    net = SptAudioGen(1, encoders=list(encoders), separation=separation,
                      params=SptAudioGenParams(sep_num_tracks=nsep, loc_fc_units=list(loc_units)))
    net.load_variables(P)
    out = net.inference_ops(inp['audio'], inp.get('video'), inp.get('flow'))
    T.cuda.synchronize()
    return net, orc, out.cpu().numpy(), ref


def check_out(got, ref):
    err = rms(got - ref)
    assert np.isfinite(got).all()
    assert err <= ABS_TOL, 'abs RMS err %g' % err
    assert err <= REL_TOL * rms(ref), 'rel RMS err %g (out rms %g)' % (err / rms(ref), rms(ref))


def test_audio_only_intermediates_and_output(T):
    net, orc, got, ref = run_pair(T, ['audio'])
    B = 2
    names = ['mag'] + ['audio_encoder/conv%d' % l for l in range(1, 6)] + ['bottleneck', 'localization/coeffs']
    for n in names:
        g = net.intermediate(B, n).cpu().numpy()
        r = orc.ends['audio_encoder/mag' if n == 'mag' else n]
        r = r.reshape(g.shape)
        e = rel_rms_err(g, r)
        assert e < 5e-5, (n, e)
    # the default forward folds sigmoid + track mix into deconv1's epilogue: the logits are not materialised, and asking says so
    with pytest.raises(Exception, match='materialize_mask'):
        net.intermediate(B, 'separation/deconv1')
    check_out(got, ref)
    # with the logits kept (two kernels): the logits against the oracle, and the same output to rounding
    net.set_option(B, 'materialize_mask', 1)
    inp = synth_inputs(B, ['audio'], seed=1234)
    got2 = net.inference_ops(inp['audio']).cpu().numpy()
    d1 = net.intermediate(B, 'separation/deconv1').cpu().numpy()          # rows 44..66 only
    assert rel_rms_err(d1, orc.ends['separation/deconv1'][:, 44:67]) < 5e-5
    check_out(got2, ref)
    assert rel_rms_err(got, got2) < 2e-6, rel_rms_err(got, got2)
    net.set_option(B, 'materialize_mask', 0)
    got3 = net.inference_ops(inp['audio']).cpu().numpy()
    assert np.array_equal(got3, got)


@pytest.mark.parametrize('tile_name', ['conv3g_kernel<128,64,64,32,3,true>', 'conv3g_kernel<64,64,32,32,4,true>',
                                       'conv3g_kernel<128,128,64,64,2,true>', 'conv3g_kernel<64,128,32,64,3,true>'])
@pytest.mark.parametrize('B', [2, 5])
def test_audio_encoder_convs_on_planes_of_the_concat_buffers(T, tile_name, B):
    """Round 6: conv2 .. conv5 of the audio encoder (model.py:161-187: VALID 3x7 / 3x5 convs, strides (2,4) (2,2) (1,1) (1,1)) on
    conv3g_kernel over fp16x2 planes of cat_l's encoder half, scaled by that half's EXACT maximum (published by conv_l's epilogue; conv3g's
    own epilogue publishes the next one).  Every conv3g tile forced on them: the layers really run it, their outputs and the network's
    hold the oracle's bars and agree with the register-staged kernels (fp32 operand, six products) to rounding."""
    from spatialaudiogen_amd.model import SptAudioGen
    enc = ['audio']
    P = init_weights(variable_specs(enc), seed=21, mode='test')
    inp = synth_inputs(B, enc, seed=55)
    orc = SptAudioGenOracle(encoders=enc)
    ref = orc.inference_ops(inp['audio'], P)
    net = SptAudioGen(1, encoders=enc, separation='unet_mask')
    net.load_variables(P)
    base = net.inference_ops(inp['audio']).cpu().numpy()          # untuned small batch: the register-staged kernels
    tid = SptAudioGen.tile_names().index(tile_name)
    layers = ['audio_encoder/conv%d' % l for l in range(2, 6)]
    for name in layers:
        net.plan_set(B, name, tid, 1)
    net.profile_enable(B, True)
    got = net.inference_ops(inp['audio']).cpu().numpy()
    rows = net.profile_report(B)
    net.profile_enable(B, False)
    for name in layers:
        assert [k for k, layer, _, _ in rows if layer == name] == [tile_name], (name, [k for k, layer, _, _ in rows if layer == name])
        assert [k for k, layer, _, _ in rows if layer == name + '/planes'] == ['h2_pack_rows_kernel']
        g = net.intermediate(B, name).cpu().numpy()
        assert rel_rms_err(g, orc.ends[name].reshape(g.shape)) < 5e-5, name
    check_out(got, ref)
    assert rel_rms_err(got, base) < 2e-5, rel_rms_err(got, base)
    assert net.counter(B, 'fp16x2_saturations') == 0


def test_audio_video(T):
    net, orc, got, ref = run_pair(T, ['audio', 'video'])
    g = net.intermediate(2, 'bottleneck').cpu().numpy()
    assert rel_rms_err(g, orc.ends['bottleneck']) < 2e-4
    # ResNet trunk output (block conv5_2, resnet.py:184-190) compared directly, not only through the bottleneck
    t = net.intermediate(2, 'video_encoder/conv5_2').cpu().numpy()
    assert t.shape == (2, 7, 14, 512) and rel_rms_err(t, orc.ends['video_encoder/conv5_2']) < 1e-4
    check_out(got, ref)


def test_audio_video_flow(T):
    net, orc, got, ref = run_pair(T, ['audio', 'video', 'flow'], seed=1)
    for enc in ('video', 'flow'):
        t = net.intermediate(2, enc + '_encoder/conv5_2').cpu().numpy()
        assert rel_rms_err(t, orc.ends[enc + '_encoder/conv5_2']) < 1e-4, enc
    check_out(got, ref)


def test_trunk_without_presplit_planes(T):
    """SAGEN_NO_P3=1 keeps the fp32-activation kernels (igemm3dw) on the 3x3 trunk convs: same bar (one subprocess, the
    switch is read when a context is created)."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ); env['SAGEN_NO_P3'] = '1'
    r = subprocess.run([sys.executable, '-m', 'pytest', os.path.join(root, 'tests', 'test_gpu_model.py'), '-m', 'gpu', '-q', '-x',
                        '-k', 'test_audio_video and not flow'], env=env, cwd=root, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:]


def test_no_separation_mode(T):
    _, _, got, ref = run_pair(T, ['audio'], separation='none', nsep=1)
    check_out(got, ref)
    from spatialaudiogen_amd.model import SptAudioGen, SptAudioGenParams
    with pytest.raises(ValueError):                     # fc3 would be sized for 33 tracks against one mono track (model.py:254 vs :274-280)
        SptAudioGen(1, encoders=['audio'], separation='none', params=SptAudioGenParams(sep_num_tracks=32))


def test_other_widths(T):
    _, _, got, ref = run_pair(T, ['audio'], nsep=64, loc_units=(256, 256), batch=3)
    check_out(got, ref)


def test_channel_order_and_step_indexing(T):
    """Decoder with hand-set localisation: fc weights zero, fc3 bias selects track k for channel o at
    step s — output channel o must equal that separated track on samples [1600 s, 1600 (s+1))."""
    import torch
    from spatialaudiogen_amd.model import SptAudioGen
    enc = ['audio']
    specs = variable_specs(enc)
    P = init_weights(specs, seed=3, mode='test')
    P['localization/fc3/weights'][:] = 0
    b = np.zeros((3, 1, 33), np.float32)
    b[0, 0, 4] = 1.0      # Y <- track 4
    b[1, 0, 32] = 0.25    # Z <- constant bias
    b[2, 0, 9] = -1.0     # X <- -track 9
    P['localization/fc3/biases'][:] = b.reshape(-1)
    inp = synth_inputs(2, enc, seed=7)
    orc = SptAudioGenOracle(encoders=enc)
    ref = orc.inference_ops(inp['audio'], P)
    tracks = orc.ends['separation/all_channels'][:, 0]                      # [B,32,4800]
    net = SptAudioGen(1, encoders=enc, separation='unet_mask')
    net.load_variables(P)
    got = net.inference_ops(inp['audio']).cpu().numpy()
    assert rms(got[:, :, 0] - tracks[:, 4]) < 1e-5
    assert np.abs(got[:, :, 1] - 0.25).max() < 1e-6
    assert rms(got[:, :, 2] + tracks[:, 9]) < 1e-5
    check_out(got, ref)


def test_batch_coupling_through_batchnorm(T):
    """Training-mode BN couples windows of a batch: zero-padding a partial batch the way
    deploy.py:125-132 does must change the real windows' outputs exactly as the oracle says."""
    from spatialaudiogen_amd.model import SptAudioGen
    enc = ['audio', 'video']
    specs = variable_specs(enc)
    P = init_weights(specs, seed=5, mode='test')
    inp = synth_inputs(3, enc, seed=11)
    a = inp['audio'].copy(); v = inp['video'].copy()
    a[2] = 0; v[2] = 0                                                     # dummy window of a short last batch
    orc = SptAudioGenOracle(encoders=enc)
    ref = orc.inference_ops(a, P, video=v)
    net = SptAudioGen(1, encoders=enc, separation='unet_mask')
    net.load_variables(P)
    got = net.inference_ops(a, v).cpu().numpy()
    check_out(got, ref)
    ref_full = SptAudioGenOracle(encoders=enc).inference_ops(inp['audio'], P, video=inp['video'])
    assert rms(ref_full[:2] - ref[:2]) > 10 * ABS_TOL                     # the coupling is real


def test_errors_are_loud(T):
    from spatialaudiogen_amd.model import SptAudioGen
    from spatialaudiogen_amd._lib import SagenError
    net = SptAudioGen(1, encoders=['audio'], separation='unet_mask')
    with pytest.raises(RuntimeError):
        net.inference_ops(np.zeros((1, 52799, 1), np.float32))             # no variables loaded
    P = init_weights(variable_specs(['audio']), seed=0)
    bad = dict(P); bad.pop('separation/deconv3/biases')
    with pytest.raises(KeyError):
        net.load_variables(bad)
    net.load_variables(P)
    with pytest.raises(ValueError):
        net.inference_ops(np.zeros((1, 1000, 1), np.float32))
    net2 = SptAudioGen(1, audio_rate=44100, video_rate=10, encoders=['audio'], separation='unet_mask')
    net2.load_variables(init_weights(net2.variable_specs(), seed=0))
    with pytest.raises(SagenError):
        net2.inference_ops(np.zeros((1, net2.snd_size, 1), np.float32))    # unsupported geometry: explicit error


def test_autotuned_plan_keeps_parity(T):
    """sagen_autotune picks (tile, split-K) per layer by timing; whatever it picks must stay inside the bar."""
    from spatialaudiogen_amd.model import SptAudioGen
    enc = ['audio', 'video']
    P = init_weights(variable_specs(enc), seed=2, mode='test')
    inp = synth_inputs(3, enc, seed=21)
    ref = SptAudioGenOracle(encoders=enc).inference_ops(inp['audio'], P, video=inp['video'])
    net = SptAudioGen(1, encoders=enc, separation='unet_mask')
    net.load_variables(P)
    plan = net.autotune(inp['audio'], inp['video'])
    timed = [row for row in plan if not row[0].endswith('#materialize')]       # (those rows are on/off switches)
    assert len(timed) >= 30 and all(us > 0 for _, _, _, us in timed)
    check_out(net.inference_ops(inp['audio'], inp['video']).cpu().numpy(), ref)


def _all_tiles():
    """Every contraction-kernel instantiation of the library (host-only ABI calls) with a rotating split-K factor."""
    try:
        from spatialaudiogen_amd import _lib
        n = _lib.lib().sagen_num_tiles()
    except Exception:          # library not built yet (collection on a fresh checkout): the table size at the time of writing
        n = 46
    sks = [1, 2, 1, 3, 1, 4, 1]
    return [(t, sks[t % len(sks)]) for t in range(n)] + [(0, 2), (21, 2), (34, 2), (40, 1)]


@pytest.mark.parametrize('tile,splitk', _all_tiles())
def test_forced_plans_cover_every_tile_and_splitk_path(T, tile, splitk):
    """Pin every batch-norm conv of the trunk (and the dense decoder / FC layers) to one tile shape and split-K
    factor: exercises split-K partials + reduce-with-statistics and each kernel instantiation end to end."""
    from spatialaudiogen_amd.model import SptAudioGen
    enc = ['audio', 'video']
    P = init_weights(variable_specs(enc), seed=6, mode='test')
    inp = synth_inputs(2, enc, seed=31)
    ref = SptAudioGenOracle(encoders=enc).inference_ops(inp['audio'], P, video=inp['video'])
    net = SptAudioGen(1, encoders=enc, separation='unet_mask')
    net.load_variables(P)
    net.inference_ops(inp['audio'], inp['video'])
    for name in variable_specs(enc):
        if name.endswith('/weights'):
            net.plan_set(2, name[:-len('/weights')], tile, splitk)
    check_out(net.inference_ops(inp['audio'], inp['video']).cpu().numpy(), ref)


@pytest.mark.parametrize('tile_name', ['conv3h_kernel<128,64,64,32,1>', 'conv3h_kernel<128,128,64,64,1>', 'conv3h_kernel<64,64,32,32,1>',
                                       'conv3h_kernel<256,64,64,64,1>', 'conv3h_kernel<128,64,64,32,2>', 'conv3h_kernel<64,64,32,32,2>',
                                       'conv3h_kernel<64,64,32,32,4>'])
@pytest.mark.parametrize('B', [2, 5])
def test_dh_split_of_every_conv3h_tile(T, tile_name, B):
    """Round 6: conv3h_kernel with one workgroup per (tile, FILTER ROW) - split-K = 3 over the vertical taps, raw partials [3][M][N],
    summed (+ batch-norm statistics) by splitk_reduce_stats_kernel.  Every conv3h tile forced onto every stride-1 3x3 trunk conv with the
    split: the launches really are split (the profile shows the reducer behind the conv), and the output holds the oracle's bar and
    agrees with the unsplit run of the same tile."""
    from spatialaudiogen_amd.model import SptAudioGen
    enc = ['audio', 'video']
    P = init_weights(variable_specs(enc), seed=8, mode='test')
    inp = synth_inputs(B, enc, seed=47)
    ref = SptAudioGenOracle(encoders=enc).inference_ops(inp['audio'], P, video=inp['video'])
    net = SptAudioGen(1, encoders=enc, separation='unet_mask')
    net.load_variables(P)
    net.inference_ops(inp['audio'], inp['video'])
    tid = SptAudioGen.tile_names().index(tile_name)
    trunk = [n[:-len('/weights')] for n in variable_specs(enc) if n.endswith('/weights') and n.startswith('video_encoder/conv') and '/conv1/' not in n
             and 'shortcut' not in n and not n.endswith(('3_1/conv_1/weights', '4_1/conv_1/weights', '5_1/conv_1/weights'))]
    assert len(trunk) == 13
    outs = {}
    for sk in (1, 3):
        for name in trunk:
            net.plan_set(B, name, tid, sk)
        net.profile_enable(B, True)
        outs[sk] = net.inference_ops(inp['audio'], inp['video']).cpu().numpy()
        rows = net.profile_report(B)
        net.profile_enable(B, False)
        for name in trunk:
            kernels = [k for k, layer, _, _ in rows if layer == name]
            assert tile_name in kernels, (name, kernels)
            assert ('splitk_reduce_kernel' in kernels) == (sk == 3), (name, sk, kernels)
    check_out(outs[3], ref)
    assert rms(outs[3] - outs[1]) <= 2e-5 * max(rms(outs[1]), 1e-9) + 1e-7, rms(outs[3] - outs[1])


_POISON_SRC = r"""
#include <hip/hip_runtime.h>
// every workgroup fills the whole LDS of its CU with fp16 NaN bit patterns and lingers until all CUs have one
__global__ __launch_bounds__(256, 1) void poison_kernel(unsigned* sink) {
    __shared__ unsigned lds[160 * 1024 / 4];
    for (int i = threadIdx.x; i < 160 * 1024 / 4; i += 256) lds[i] = 0x7fff7fffu;
    __syncthreads();
    unsigned acc = 0;
    for (int k = 0; k < 64; ++k) { acc += lds[(threadIdx.x * 97 + k * 1031) % (160 * 1024 / 4)]; __builtin_amdgcn_s_sleep(64); }
    if (acc == 12345u) sink[0] = acc;
}
extern "C" int poison_lds(void* sink) {
    hipLaunchKernelGGL(poison_kernel, dim3(2048), dim3(256), 0, 0, (unsigned*)sink);
    return (int)hipDeviceSynchronize();
}
"""


@pytest.mark.parametrize('tile_name', ['conv3hr_kernel<256,64,64,64,1>', 'conv3h_kernel<64,64,32,32,4>'])
def test_rows_a_conv3h_tile_drops_cannot_reach_the_statistics(T, tile_name, tmp_path):
    """The last two rows of a conv3h tile are contracted from whatever lies BEHIND the activation image in LDS (the filter image;
    with the three-deep ring an image that may still be in flight, i.e. bytes a previous kernel left).  The epilogue drops them by
    the out-of-range store offset and must keep them out of the batch-norm sums whatever their bit pattern: fill every CU's LDS
    with fp16 NaNs (a kernel built here with hipcc), then run the forward with the trunk pinned to the tile.  (The window in which a
    dropped row can meet foreign bytes is narrow - the image behind has usually landed - so this run is a guard on the property, not a
    proof of it: the select in the epilogue is.)"""
    import ctypes, shutil, subprocess
    from spatialaudiogen_amd.model import SptAudioGen
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    src, lib = tmp_path / 'poison.hip', tmp_path / 'libpoison.so'
    src.write_text(_POISON_SRC)
    subprocess.check_call([hipcc, '--offload-arch=gfx950', '-O2', '-fPIC', '-shared', str(src), '-o', str(lib)])
    poison = ctypes.CDLL(str(lib)).poison_lds
    poison.argtypes = [ctypes.c_void_p]
    enc = ['audio', 'video']
    P = init_weights(variable_specs(enc), seed=6, mode='test')
    inp = synth_inputs(2, enc, seed=31)
    ref = SptAudioGenOracle(encoders=enc).inference_ops(inp['audio'], P, video=inp['video'])
    net = SptAudioGen(1, encoders=enc, separation='unet_mask')
    net.load_variables(P)
    net.inference_ops(inp['audio'], inp['video'])
    tile = SptAudioGen.tile_names().index(tile_name)
    for name in variable_specs(enc):
        if name.endswith('/weights') and name.startswith('video_encoder'):
            net.plan_set(2, name[:-len('/weights')], tile, 1)
    import torch
    sink = torch.zeros(4, dtype=torch.int32, device='cuda')
    for _ in range(3):
        torch.cuda.synchronize()
        assert poison(ctypes.c_void_p(sink.data_ptr())) == 0
        out = net.inference_ops(inp['audio'], inp['video']).cpu().numpy()
        assert np.isfinite(out).all()
        check_out(out, ref)


@pytest.mark.parametrize('case', ['tiny', 'huge', 'mixed', 'offset'])
def test_fp16x2_trunk_planes_hold_over_the_range_of_batch_norm_parameters(T, case):
    """conv3h_kernel evaluates the trunk's 3x3 convs on TWO fp16 planes per operand (three products per multiply).  fp16 has no
    fp32 exponent range, so the plane pass scales each tensor by 2^ka derived from the batch-norm that produced it
    (max_c |beta_c| + 8 |gamma_c| -> [512, 1024)) and the filter pack scales each filter by 2^kw.  The scheme must hold whatever the
    learned gamma / beta are: activations 1000 times smaller ('tiny') or 300 times larger ('huge') than the usual O(1), channels of
    very different size inside one tensor ('mixed'), a large common offset ('offset').  Same bars against the fp64 oracle as every
    forward, and the result of the three-plane bf16 kernels (fp16x2 = 0) within 2e-5."""
    from spatialaudiogen_amd.model import SptAudioGen
    enc = ['audio', 'video']
    B = 3
    P = init_weights(variable_specs(enc), seed=21, mode='test')
    r = np.random.Generator(np.random.PCG64(3))
    for k in list(P):
        if k.startswith('video_encoder/') and k.endswith('/bn/gamma') and '/conv1/' not in k:
            g, b = P[k].copy(), P[k.replace('gamma', 'beta')].copy()
            if case == 'tiny':
                g *= 1e-3; b *= 1e-3
            elif case == 'huge':
                g *= 300.0; b *= 300.0
            elif case == 'mixed':
                g *= np.exp(r.uniform(np.log(1e-3), np.log(30.0), size=g.shape)).astype(np.float32)
            else:
                b += 40.0
            P[k], P[k.replace('gamma', 'beta')] = g.astype(np.float32), b.astype(np.float32)
    inp = synth_inputs(B, enc, seed=43)
    orc = SptAudioGenOracle(encoders=enc)
    ref = orc.inference_ops(inp['audio'], P, video=inp['video'])
    net = SptAudioGen(1, encoders=enc, separation='unet_mask')
    net.load_variables(P)
    got = net.inference_ops(inp['audio'], inp['video']).cpu().numpy()
    trunk = net.intermediate(B, 'video_encoder/conv5_2').cpu().numpy()
    net.profile_enable(B, True)
    net.inference_ops(inp['audio'], inp['video'])
    kernels = {k for k, layer, us, fl in net.profile_report(B)}
    net.profile_enable(B, False)
    assert any(k.startswith('conv3h') for k in kernels), kernels
    net.set_option(B, 'fp16x2', 0)
    other = net.inference_ops(inp['audio'], inp['video']).cpu().numpy()
    trunk3 = net.intermediate(B, 'video_encoder/conv5_2').cpu().numpy()
    # these parameter sets move the output's scale by orders of magnitude and put batch-norms into their eps-dominated regime (rounding
    # noise is amplified for EVERY arithmetic): the bars are relative, and the yardstick is the six-product bf16x3 path on the same input
    tr = orc.ends['video_encoder/conv5_2']
    e2, e3 = rel_rms_err(trunk, tr), rel_rms_err(trunk3, tr)
    o2, o3 = rel_rms_err(got, ref), rel_rms_err(other, ref)
    print('\n[%s] conv5_2 rel err: fp16x2 %.3g, bf16x3 %.3g; output rel err: fp16x2 %.3g, bf16x3 %.3g (output rms %.3g)' % (case, e2, e3, o2, o3, rms(ref)))
    assert np.isfinite(got).all() and np.isfinite(trunk).all()
    assert net.counter(B, 'fp16x2_saturations') == 0            # no activation had to be clamped: the statistical bounds held
    assert e2 < 1e-4 and o2 < 1e-3, (e2, o2)
    assert e2 <= 1.5 * e3 + 1e-7 and o2 <= 1.5 * o3 + 1e-7, (e2, e3, o2, o3)     # no less accurate than the six-product path
    assert rel_rms_err(trunk, trunk3) < 1e-4


def test_fp16x2_saturation_counter_reports_what_the_planes_cannot_hold(T):
    """The plane passes clamp to +-65000 and COUNT what they clamp (sagen_counter 'fp16x2_saturations'): 0 after ordinary forwards.
    The scale exponent is limited to 2^+-60, so a batch-norm with gamma = 1e25 (activations of 1e25) cannot be brought into fp16's
    range: the counter must say so."""
    from spatialaudiogen_amd.model import SptAudioGen
    enc = ['audio', 'video']
    B = 2
    P = init_weights(variable_specs(enc), seed=3, mode='test')
    inp = synth_inputs(B, enc, seed=9)
    net = SptAudioGen(1, encoders=enc, separation='unet_mask')
    net.load_variables(P)
    net.inference_ops(inp['audio'], inp['video'])
    net.inference_ops(inp['audio'], inp['video'])
    assert net.counter(B, 'fp16x2_saturations') == 0
    P2 = dict(P)
    P2['video_encoder/conv2_1/conv_1/bn/gamma'] = (P['video_encoder/conv2_1/conv_1/bn/gamma'] * 1e25).astype(np.float32)
    net2 = SptAudioGen(1, encoders=enc, separation='unet_mask')
    net2.load_variables(P2)
    net2.inference_ops(inp['audio'], inp['video'])
    assert net2.counter(B, 'fp16x2_saturations') > 0
    # (activations of 1e25 are beyond what ANY of the kernels can process: the fp32 (sum, sumsq) statistics of the next batch-norm
    #  overflow; a scale exponent within +-60 covers bounds up to 6e20, where the squares already leave fp32 - the counter cannot fire
    #  inside the range in which the fp32 path itself is valid)


def test_stride2_block_inputs_run_on_the_plane_fed_gather_kernel(T):
    """The first block of ResNet stages 3, 4 and 5: the last merge of the previous stage writes its block output as bf16x3 planes
    (and nothing else) and the 3x3 stride-2 conv_1 + the 1x1 stride-2 shortcut read THEM (conv3g_kernel: LDS-DMA gather -> MFMA, no
    operand split in the K loop).  With the option on the shape heuristic must pick it, the result must hold the usual bars, and it must equal the
    register-staged route (sagen_set_option 'plane_gather' = 0: fp32 block outputs, igemm3_kernel) to rounding."""
    from spatialaudiogen_amd.model import SptAudioGen
    enc = ['audio', 'video']
    B = 3
    P = init_weights(variable_specs(enc), seed=12, mode='test')
    inp = synth_inputs(B, enc, seed=41)
    orc = SptAudioGenOracle(encoders=enc)
    ref = orc.inference_ops(inp['audio'], P, video=inp['video'])
    net = SptAudioGen(1, encoders=enc, separation='unet_mask')
    net.load_variables(P)
    net.inference_ops(inp['audio'], inp['video'])
    net.set_option(B, 'plane_gather', 1)            # (opt-in: measured no faster than igemm3_kernel on these small-M layers)
    got = net.inference_ops(inp['audio'], inp['video']).cpu().numpy()
    check_out(got, ref)
    trunk = net.intermediate(B, 'video_encoder/conv5_2').cpu().numpy()
    assert rel_rms_err(trunk, orc.ends['video_encoder/conv5_2']) < 1e-4
    stride2 = ['video_encoder/conv%d_1/%s' % (st, nm) for st in (3, 4, 5) for nm in ('conv_1', 'shortcut')]

    def kernels_used():
        net.profile_enable(B, True)
        net.inference_ops(inp['audio'], inp['video'])
        rows = net.profile_report(B)
        net.profile_enable(B, False)
        return {layer: k for k, layer, us, fl in rows}
    used = kernels_used()
    for layer in stride2:
        assert used[layer].startswith('conv3g_kernel'), (layer, used[layer])
    net.set_option(B, 'plane_gather', 0)
    other = net.inference_ops(inp['audio'], inp['video']).cpu().numpy()
    used = kernels_used()
    for layer in stride2:
        assert used[layer].startswith('igemm3_kernel'), (layer, used[layer])
    check_out(other, ref)
    assert rel_rms_err(got, other) < 1e-5
    net.set_option(B, 'plane_gather', 1)
    assert np.array_equal(net.inference_ops(inp['audio'], inp['video']).cpu().numpy(), got)


def test_full_benchmark_batch_against_the_independent_cpu_reference(T):
    """BASELINE configs[1] at its real size (32 windows, audio + video): the heuristic launch plan AND the autotuned plan
    against oracle/torch_ref.py in fp64 (an implementation independent of both the numpy oracle and the HIP code).
    Catches anything that only shows at full tile counts: edge tiles, split-K ranges, the kernels the tuner picks."""
    import torch
    from oracle.torch_ref import TorchRef
    from spatialaudiogen_amd.model import SptAudioGen
    enc = ['audio', 'video']
    B = 32
    P = init_weights(variable_specs(enc), seed=11, mode='test')
    inp = synth_inputs(B, enc, seed=77)
    ref = TorchRef(P, enc, dtype=torch.float64).forward(inp['audio'], inp['video'])
    ref = ref.numpy() if isinstance(ref, torch.Tensor) else np.asarray(ref)
    net = SptAudioGen(1, encoders=enc, separation='unet_mask')
    net.load_variables(P)
    check_out(net.inference_ops(inp['audio'], inp['video']).cpu().numpy(), ref)
    plan = net.autotune(inp['audio'], inp['video'])
    assert any(row[1].startswith('igemm3') for row in plan)
    check_out(net.inference_ops(inp['audio'], inp['video']).cpu().numpy(), ref)


def test_in_launch_split_k_combine_agrees_with_the_reducer(T, monkeypatch):
    """SAGEN_SK_FUSED=1 (read when a context is created): the workgroup that draws the last ticket of an output tile adds the split-K
    partials itself in a fixed order - one launch fewer per split layer, the same sums (off by default: measured no faster)."""
    from spatialaudiogen_amd.model import SptAudioGen
    for enc, B in ((['audio'], 10), (['audio', 'video'], 4)):
        P = init_weights(variable_specs(enc), seed=3, mode='test')
        inp = synth_inputs(B, enc, seed=5)
        outs, counts = [], []
        for fused in (False, True):
            if fused:
                monkeypatch.setenv('SAGEN_SK_FUSED', '1')
            else:
                monkeypatch.delenv('SAGEN_SK_FUSED', raising=False)
            net = SptAudioGen(1, encoders=enc, separation='unet_mask')
            net.load_variables(P)
            for _ in range(2):                         # twice: the tickets must have been re-armed
                out = net.inference_ops(inp['audio'], inp.get('video')).cpu().numpy()
            net.profile_enable(B, True)
            net.inference_ops(inp['audio'], inp.get('video'))
            counts.append(sum(1 for k, layer, us, fl in net.profile_report(B) if k.startswith('splitk_reduce')))
            net.profile_enable(B, False)
            outs.append(out)
        monkeypatch.delenv('SAGEN_SK_FUSED', raising=False)
        # (a layer whose reducer also replicates rows without splitting K, or carries batch-norm statistics, or has N % 4 != 0 keeps it)
        assert counts[0] >= 6 and counts[1] <= 2, counts
        # the fixed-order sum of the partials is the same set of fp32 additions up to their association (the reducer of a layer with few
        # outputs and many partials sums in eight interleaved slices): agreement at rounding level, run to run identical
        assert rel_rms_err(outs[1], outs[0]) < 1e-6
        assert np.array_equal(outs[1], out)


def test_uint8_video_entry_point(T):
    """sagen_forward_u8: frames as the uint8 the decoder produced.
    (a) general kernels (sagen_set_option 'u8_fast_stem' = 0): x/255 - 0.5 (myutils.py:88-89) fused into the device-side pad pass -
        bit for bit the forward on the float32 frames the feeder would have produced (double-precision divide, one rounding);
    (b) default: the one-operand-plane stem (stem8.hip: u - 128 is exact in bf16, three MFMA products instead of six, conv + raw pool
        + statistics in one kernel).  Not bit-identical - it convolves the EXACT x where the float path convolves the rounded float32
        x - but fp32-equivalent: against the fp64 oracle the same bar as every forward, and within 1e-5 (relative RMS) of (a)."""
    from spatialaudiogen_amd.model import SptAudioGen
    enc = ['audio', 'video']
    B = 3
    P = init_weights(variable_specs(enc), seed=2, mode='test')
    inp = synth_inputs(B, enc, seed=77)
    r = np.random.Generator(np.random.PCG64(5))
    u8 = r.integers(0, 256, size=(B, 1, 224, 448, 3)).astype(np.uint8)
    f32 = (u8 / 255. - 0.5).astype(np.float32)                      # feeder.img_prep_fcn on the decoded frame
    net = SptAudioGen(1, encoders=enc, separation='unet_mask')
    net.load_variables(P)
    net.inference_ops(inp['audio'], f32)
    net.set_option(B, 'f16_fast_stem', 0)                           # the general float-frame kernels (igemm3s2 + pool pass) for (a)
    a = net.inference_ops(inp['audio'], f32).cpu().numpy()
    net.set_option(B, 'u8_fast_stem', 0)
    b = net.inference_ops(inp['audio'], u8).cpu().numpy()
    c = net.inference_ops(inp['audio'], T.as_tensor(u8).cuda()).cpu().numpy()
    assert np.isfinite(a).all() and np.array_equal(a, b) and np.array_equal(a, c)
    net.set_option(B, 'u8_fast_stem', 1)
    d = net.inference_ops(inp['audio'], T.as_tensor(u8).cuda()).cpu().numpy()
    trunk = net.intermediate(B, 'video_encoder/conv5_2').cpu().numpy()
    assert not np.array_equal(a, d)                                 # (it really ran other kernels)
    assert rel_rms_err(d, a) < 1e-5, rel_rms_err(d, a)
    orc = SptAudioGenOracle(encoders=enc)
    ref = orc.inference_ops(inp['audio'], P, video=f32)
    check_out(d, ref)
    assert rel_rms_err(trunk, orc.ends['video_encoder/conv5_2']) < 1e-4
    assert rms(d - ref) <= 1.5 * rms(a - ref) + 1e-7, (rms(d - ref), rms(a - ref))      # no less accurate than the float-frame kernels


@pytest.mark.parametrize('B', [1, 5, 32])
def test_uint8_stem_geometry_and_negative_gammas(T, B):
    """stem8pool_kernel on its own terms: image borders (the -0.5 padding IS x = 0), the partial last patch row / column, dummy MFMA
    rows, batch sizes that leave persistent workgroups without work (B = 1: 112 patches for 256 workgroups) or give them several
    (B = 32: 14 each), min-pooling for negative gammas and the gamma = 0 channel - the trunk output conv5_2 and the final output
    against the oracle."""
    from spatialaudiogen_amd.model import SptAudioGen
    enc = ['audio', 'video']
    P = init_weights(variable_specs(enc), seed=8 + B, mode='test')
    g = P['video_encoder/conv1/conv/bn/gamma'].copy()
    g[1::2] *= -1.0
    g[6] = 0.0
    P['video_encoder/conv1/conv/bn/gamma'] = g
    inp = synth_inputs(B, enc, seed=31 + B)
    r = np.random.Generator(np.random.PCG64(B))
    u8 = r.integers(0, 256, size=(B, 1, 224, 448, 3)).astype(np.uint8)
    u8[:, :, :3] = 255; u8[:, :, -3:] = 0; u8[:, :, :, :3] = 0; u8[:, :, :, -3:] = 255          # loud borders: a padding fault shows
    f32 = (u8 / 255. - 0.5).astype(np.float32)
    net = SptAudioGen(1, encoders=enc, separation='unet_mask')
    net.load_variables(P)
    net.profile_enable(B, True)
    got = net.inference_ops(inp['audio'], T.as_tensor(u8).cuda()).cpu().numpy()
    kernels = {k for k, layer, us, fl in net.profile_report(B)}
    trunk = net.intermediate(B, 'video_encoder/conv5_2').cpu().numpy()
    # default (round 5): ONE fp16 plane of u - 128 x TWO fp16 filter planes; 'u8_stem_h2' = 0: the bf16 plane x three bf16 filter planes
    assert 'stem8pool_kernel<h2>' in kernels, kernels
    net.set_option(B, 'u8_stem_h2', 0)
    got3 = net.inference_ops(inp['audio'], T.as_tensor(u8).cuda()).cpu().numpy()
    kernels3 = {k for k, layer, us, fl in net.profile_report(B)}
    net.profile_enable(B, False)
    trunk3 = net.intermediate(B, 'video_encoder/conv5_2').cpu().numpy()
    assert 'stem8pool_kernel' in kernels3 and 'stem8pool_kernel<h2>' not in kernels3, kernels3
    assert not np.array_equal(got, got3) and rel_rms_err(got, got3) < 1e-5 and rel_rms_err(trunk, trunk3) < 1e-5
    net.set_option(B, 'u8_stem_h2', 1)
    if B <= 5:
        orc = SptAudioGenOracle(encoders=enc)
        ref = orc.inference_ops(inp['audio'], P, video=f32)
        check_out(got, ref)
        check_out(got3, ref)
        assert rel_rms_err(trunk, orc.ends['video_encoder/conv5_2']) < 1e-4
        assert rel_rms_err(trunk3, orc.ends['video_encoder/conv5_2']) < 1e-4
        assert rms(got - ref) <= 1.5 * rms(got3 - ref) + 1e-7, (rms(got - ref), rms(got3 - ref))
    else:                                                           # full size: against the general kernels on the float frames
        ref = net.inference_ops(inp['audio'], f32).cpu().numpy()
        assert rel_rms_err(got, ref) < 1e-5
        assert rel_rms_err(trunk, net.intermediate(B, 'video_encoder/conv5_2').cpu().numpy()) < 1e-5


@pytest.mark.parametrize('B,scale', [(1, 1.0), (5, 37.0), (32, 1.0)])
def test_float_frame_stem_on_fp16x2_planes(T, B, scale):
    """Float frames (the flow encoder's input; video handed over as float32) run the F16 variant of the fused stem kernel: the frame as
    two fp16 planes scaled by the batch's exact maximum (stem8.hip).  Arbitrary float values (not images of uint8), a frame scale far
    from 1, loud borders, negative / zero gammas, batch sizes that starve or crowd the persistent workgroups: against the oracle (small
    batches) and against the general float-frame kernels (sagen_set_option 'f16_fast_stem' = 0) - and the kernel really ran."""
    from spatialaudiogen_amd.model import SptAudioGen
    enc = ['audio', 'video']
    P = init_weights(variable_specs(enc), seed=18 + B, mode='test')
    g = P['video_encoder/conv1/conv/bn/gamma'].copy()
    g[1::2] *= -1.0
    g[6] = 0.0
    P['video_encoder/conv1/conv/bn/gamma'] = g
    inp = synth_inputs(B, enc, seed=61 + B)
    r = np.random.Generator(np.random.PCG64(40 + B))
    f32 = (scale * r.uniform(-0.5, 0.5, size=(B, 1, 224, 448, 3))).astype(np.float32)
    f32[:, :, :3] = 0.5 * scale; f32[:, :, -3:] = -0.5 * scale; f32[:, :, :, :3] = -0.5 * scale; f32[:, :, :, -3:] = 0.5 * scale   # loud borders
    net = SptAudioGen(1, encoders=enc, separation='unet_mask')
    net.load_variables(P)
    net.profile_enable(B, True)
    got = net.inference_ops(inp['audio'], f32).cpu().numpy()
    kernels = {k for k, layer, us, fl in net.profile_report(B)}
    net.profile_enable(B, False)
    assert 'stem8pool_kernel<f16>' in kernels, kernels
    trunk = net.intermediate(B, 'video_encoder/conv5_2').cpu().numpy()
    assert net.counter(B, 'fp16x2_saturations') == 0
    net.set_option(B, 'f16_fast_stem', 0)
    other = net.inference_ops(inp['audio'], f32).cpu().numpy()
    trunk_o = net.intermediate(B, 'video_encoder/conv5_2').cpu().numpy()
    assert not np.array_equal(got, other)
    assert rel_rms_err(got, other) < 1e-5 and rel_rms_err(trunk, trunk_o) < 1e-5
    if B <= 5:
        orc = SptAudioGenOracle(encoders=enc)
        ref = orc.inference_ops(inp['audio'], P, video=f32)
        check_out(got, ref)
        assert rel_rms_err(trunk, orc.ends['video_encoder/conv5_2']) < 1e-4
        assert rms(got - ref) <= 1.5 * rms(other - ref) + 1e-7


def test_fused_stem_pool_with_negative_gammas(T):
    """stempool_kernel pools the RAW stem output with max or min per channel by the sign of gamma (relu(bn(.)) is monotone per
    channel).  Half of the stem's gammas negative, one exactly zero: same bar against the oracle, and the same result as the unfused
    stem + pool kernels (SAGEN_NO_STEMPOOL=1, in a subprocess) to rounding."""
    import os, subprocess, sys, tempfile
    from spatialaudiogen_amd.model import SptAudioGen
    enc = ['audio', 'video']
    P = init_weights(variable_specs(enc), seed=6, mode='test')
    g = P['video_encoder/conv1/conv/bn/gamma'].copy()
    g[::2] *= -1.0
    g[5] = 0.0
    P['video_encoder/conv1/conv/bn/gamma'] = g
    inp = synth_inputs(3, enc, seed=55)
    orc = SptAudioGenOracle(encoders=enc)
    ref = orc.inference_ops(inp['audio'], P, video=inp['video'])
    net = SptAudioGen(1, encoders=enc, separation='unet_mask')
    net.load_variables(P)
    got = net.inference_ops(inp['audio'], inp['video']).cpu().numpy()
    check_out(got, ref)
    t = net.intermediate(3, 'video_encoder/conv5_2').cpu().numpy()
    assert rel_rms_err(t, orc.ends['video_encoder/conv5_2']) < 1e-4
    fn = tempfile.mktemp(suffix='.npz')
    np.savez(fn, **{k.replace('/', '|'): v for k, v in P.items()})
    code = ("import sys, numpy as np; sys.path.insert(0, %r)\n"
            "from spatialaudiogen_amd.model import SptAudioGen\n"
            "from spatialaudiogen_amd.weights import synth_inputs\n"
            "P = {k.replace('|', '/'): v for k, v in np.load(%r).items()}\n"
            "inp = synth_inputs(3, ['audio', 'video'], seed=55)\n"
            "net = SptAudioGen(1, encoders=['audio', 'video'], separation='unet_mask'); net.load_variables(P)\n"
            "np.save(%r, net.inference_ops(inp['audio'], inp['video']).cpu().numpy())\n") % (
                os.path.dirname(os.path.dirname(os.path.abspath(__file__))), fn, fn + '.out.npy')
    subprocess.run([sys.executable, '-c', code], check=True, env=dict(os.environ, SAGEN_NO_STEMPOOL='1'), timeout=600)
    unfused = np.load(fn + '.out.npy')
    assert rms(got - unfused) <= 2e-5 * max(rms(unfused), 1e-9) + 1e-6


def test_full_benchmark_batch_of_uint8_frames_against_the_independent_cpu_reference(T):
    """The headline's exact kernel set - sagen_forward_u8 -> stem8pool_kernel on ONE fp16 operand plane x two fp16 filter planes - at its real size (32 windows)
    directly against oracle/torch_ref.py in fp64 (it was held against the float-frame kernels only, a two-link chain)."""
    import torch
    from oracle.torch_ref import TorchRef
    from spatialaudiogen_amd.model import SptAudioGen
    enc = ['audio', 'video']
    B = 32
    P = init_weights(variable_specs(enc), seed=13, mode='test')
    inp = synth_inputs(B, enc, seed=78)
    u8 = np.round((inp['video'].astype(np.float64) + 0.5) * 255.0).astype(np.uint8)
    assert np.array_equal((u8.astype(np.float64) / 255. - 0.5).astype(np.float32), inp['video'])      # the same frames, as decoded
    ref = TorchRef(P, enc, dtype=torch.float64).forward(inp['audio'], inp['video'])
    ref = ref.numpy() if isinstance(ref, torch.Tensor) else np.asarray(ref)
    net = SptAudioGen(1, encoders=enc, separation='unet_mask')
    net.load_variables(P)
    net.profile_enable(B, True)
    got = net.inference_ops(inp['audio'], u8).cpu().numpy()
    kernels = {k for k, layer, us, fl in net.profile_report(B)}
    net.profile_enable(B, False)
    assert 'stem8pool_kernel<h2>' in kernels, kernels
    check_out(got, ref)
    assert net.counter(B, 'fp16x2_saturations') == 0


def test_three_stream_batch_at_full_size_against_the_independent_cpu_reference(T):
    """BASELINE configs[2] at its real size: audio + video + flow, 32 windows, against fp64 oracle/torch_ref.py (it was compared with
    the oracle at B = 2 and through properties at B = 32 only)."""
    import torch
    from oracle.torch_ref import TorchRef
    from spatialaudiogen_amd.model import SptAudioGen
    enc = ['audio', 'video', 'flow']
    B = 32
    P = init_weights(variable_specs(enc), seed=14, mode='test')
    inp = synth_inputs(B, enc, seed=79)
    ref = TorchRef(P, enc, dtype=torch.float64).forward(inp['audio'], inp['video'], inp['flow'])
    ref = ref.numpy() if isinstance(ref, torch.Tensor) else np.asarray(ref)
    net = SptAudioGen(1, encoders=enc, separation='unet_mask')
    net.load_variables(P)
    check_out(net.inference_ops(inp['audio'], inp['video'], inp['flow']).cpu().numpy(), ref)
    assert net.counter(B, 'fp16x2_saturations') == 0


def test_skinny_fc_kernel_opt_in_agrees_with_the_oracle(T, monkeypatch):
    """SAGEN_FCM=1 (read when a context is created): the bottleneck / localisation / fc-feats layers on fcm_kernel (exact fp32 MFMA on the
    variables in their TF layout, fc1 + fc-feats in one launch) instead of the general contraction kernels - an experiment kept opt-in
    because it measured slower; same bars, and the kernels must really be the ones running."""
    from spatialaudiogen_amd.model import SptAudioGen
    monkeypatch.setenv('SAGEN_FCM', '1')
    for enc, B in ((['audio'], 10), (['audio', 'video'], 3)):
        net, orc, got, ref = run_pair(T, enc, batch=B, seed=4)
        check_out(got, ref)
        assert rel_rms_err(net.intermediate(B, 'bottleneck').cpu().numpy(), np.asarray(orc.ends['bottleneck']).reshape(B, 3, -1)) < 2e-5
        net.profile_enable(B, True)
        net.inference_ops(*[x for x in (synth_inputs(B, enc, seed=1238)[k] for k in ('audio', 'video')[:len(enc)])])
        kernels = [k for k, layer, us, fl in net.profile_report(B)]
        net.profile_enable(B, False)
        assert sum(1 for k in kernels if k.startswith('fcm_kernel')) == (4 if 'video' not in enc else 5), kernels


def test_scatter_form_decoder_on_planes_and_on_fp32_operands_and_the_round4_form_agree(T, monkeypatch):
    """The mask decoder at inference: deconv5 .. deconv2 in scatter form (GEMM over the live band of input rows + gather), on fp16x2 planes
    of the concat buffers scaled by the producers' exact maxima (from 16 windows on by default; forced here at B = 3) or on the fp32
    buffers (igemm3_kernel) - and the round-4 depth-to-space form over the full grids (SAGEN_NO_DECONV_SCATTER=1): all three against
    the oracle, the kernels that ran checked by name, and no plane element clamped."""
    from spatialaudiogen_amd.model import SptAudioGen
    enc, B = ['audio', 'video'], 3
    P = init_weights(variable_specs(enc), seed=15, mode='test')
    inp = synth_inputs(B, enc, seed=55)
    ref = SptAudioGenOracle(encoders=enc).inference_ops(inp['audio'], P, video=inp['video'])

    def run(net):
        net.profile_enable(B, True)
        y = net.inference_ops(inp['audio'], inp['video']).cpu().numpy()
        rows = net.profile_report(B)
        net.profile_enable(B, False)
        return y, {(layer, k.split('<')[0]) for k, layer, us, fl in rows if layer.startswith('separation/deconv')}
    net = SptAudioGen(1, encoders=enc, separation='unet_mask')
    net.load_variables(P)
    net.inference_ops(inp['audio'], inp['video'])
    y_f32, k_f32 = run(net)                               # B = 3 < 16: scatter form on the fp32 concat buffers
    assert ('separation/deconv5', 'deconv_gather_kernel') in k_f32 and ('separation/deconv5', 'h2_pack_rows_kernel') not in k_f32
    net.set_option(B, 'decoder_planes', 1)
    y_pl, k_pl = run(net)
    for l in (2, 3, 4, 5):
        assert ('separation/deconv%d' % l, 'h2_pack_rows_kernel') in k_pl and ('separation/deconv%d' % l, 'conv3g_kernel') in k_pl, k_pl
    assert ('separation/deconv1', 'conv3g_kernel') in k_pl
    assert net.counter(B, 'fp16x2_saturations') == 0
    monkeypatch.setenv('SAGEN_NO_DECONV_SCATTER', '1')
    monkeypatch.setenv('SAGEN_NO_DECONV1_PLANES', '1')
    old = SptAudioGen(1, encoders=enc, separation='unet_mask')
    old.load_variables(P)
    old.inference_ops(inp['audio'], inp['video'])
    y_r4, k_r4 = run(old)
    assert not any(k == 'deconv_gather_kernel' for _, k in k_r4) and ('separation/deconv1', 'igemm3_kernel') in k_r4
    for y in (y_f32, y_pl, y_r4):
        check_out(y, ref)
    assert rel_rms_err(y_pl, y_r4) < 1e-5 and rel_rms_err(y_f32, y_r4) < 1e-5
