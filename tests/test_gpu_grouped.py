"""Grouped launch (round 6; include/sagen.h: sagen_create_grouped / sagen_forward_grouped[_u8]): G independent batches of B windows as
ONE launch per layer.  The reference runs one sess.run per batch and its batch-norm couples the windows OF a batch (model.py:197,
resnet.py:123), so every batch must keep its own statistics: each group's output has to equal - bit for bit - what the ungrouped
forward gives for that batch alone."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from util import rms, ensure_lib          # noqa: E402
from oracle.np_oracle import SptAudioGenOracle          # noqa: E402
from spatialaudiogen_amd.weights import variable_specs, init_weights, synth_inputs          # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def T():
    import torch
    assert torch.cuda.is_available()
    ensure_lib()
    return torch


def _u8(T, v):
    return T.round((T.as_tensor(v).double() + 0.5) * 255.0).clamp(0, 255).to(T.uint8)


@pytest.mark.parametrize('enc,B,G,frames', [
    (['audio', 'video'], 4, 3, 'u8'),
    (['audio', 'video'], 5, 2, 'float'),
    (['audio', 'video', 'flow'], 2, 3, 'u8'),
    (['audio'], 10, 3, None),
    (['audio', 'video'], 32, 3, 'u8'),            # BASELINE configs[1]'s batch, three of them per launch
    (['audio', 'video'], 2, 16, 'u8'),            # many groups: what bench.py's auto mode reaches (10 - 16 per call) ...
    (['audio'], 3, 32, None),                     # ... and the native limit
])
def test_every_group_equals_the_single_batch_forward_bit_for_bit(T, enc, B, G, frames):
    from spatialaudiogen_amd.model import SptAudioGen
    P = init_weights(variable_specs(enc), seed=11, mode='test')
    inp = synth_inputs(G * B, enc, seed=77)
    # make the batches differ in scale as well, so that shared statistics / plane scales / maxima would show
    for g in range(G):
        inp['audio'][g * B:(g + 1) * B] *= (1.0, 0.37, 1.9)[g % 3]
    a = T.as_tensor(inp['audio']).cuda()
    v = f = None
    if 'video' in enc:
        v = _u8(T, inp['video']).cuda() if frames == 'u8' else T.as_tensor(inp['video']).cuda()
    if 'flow' in enc:
        f = T.as_tensor(inp['flow']).cuda()
        for g in range(G):
            f[g * B:(g + 1) * B] *= (1.0, 3.0, 0.25)[g % 3]
    one = SptAudioGen(1, encoders=enc, separation='unet_mask')
    one.load_variables(P)
    grp = SptAudioGen(1, encoders=enc, separation='unet_mask', groups=G)
    grp.load_variables(P)
    sl = lambda t, g: None if t is None else t[g * B:(g + 1) * B]
    want = T.cat([one.inference_ops(sl(a, g), sl(v, g), sl(f, g)) for g in range(G)], 0)
    got = grp.inference_ops(a, v, f)
    assert got.shape == want.shape
    for g in range(G):
        assert T.equal(got[g * B:(g + 1) * B], want[g * B:(g + 1) * B]), (g, float((got - want)[g * B:(g + 1) * B].abs().max()))
    # ... and run to run, with another forward in between
    grp.inference_ops(T.flip(a, [0]), None if v is None else T.flip(v, [0]), None if f is None else T.flip(f, [0]))
    assert T.equal(grp.inference_ops(a, v, f), want)
    assert grp.counter(B, 'fp16x2_saturations') == 0
    if B <= 5 and G <= 3:                        # the groups' intermediates too (sagen_set_option "intermediate_group")
        for g in range(G):
            one.inference_ops(sl(a, g), sl(v, g), sl(f, g))
            grp.set_option(B, 'intermediate_group', g)
            for name in ['bottleneck', 'audio_encoder/conv5'] + (['video_encoder/conv5_2'] if v is not None else []):
                assert T.equal(grp.intermediate(B, name), one.intermediate(B, name)), (g, name)
        grp.set_option(B, 'intermediate_group', 0)
    if B <= 5:                                   # against the oracle too (group 1: neither the first nor the default slot)
        k = {'audio': inp['audio'][B:2 * B]}
        if v is not None:
            k['video'] = (v[B:2 * B].double() / 255.0 - 0.5).float().cpu().numpy() if frames == 'u8' else inp['video'][B:2 * B]
        if f is not None:
            k['flow'] = f[B:2 * B].cpu().numpy()
        ref = SptAudioGenOracle(encoders=enc).inference_ops(k['audio'], P, video=k.get('video'), flow=k.get('flow'))
        err = rms(got[B:2 * B].cpu().numpy() - ref)
        assert err <= 1e-4 and err <= 1e-3 * rms(ref), (err, rms(ref))


def test_grouped_autotune_and_forced_dh_split_keep_the_groups_exact(T):
    """The tuner runs on the grouped context itself (its candidate launches carry the group dimension); the tuned plan, and a plan
    that forces the dh-split on the trunk, give every group the output of an ungrouped context running the same plan."""
    from spatialaudiogen_amd.model import SptAudioGen
    enc, B, G = ['audio', 'video'], 4, 2
    P = init_weights(variable_specs(enc), seed=12, mode='test')
    inp = synth_inputs(G * B, enc, seed=78)
    a, v = T.as_tensor(inp['audio']).cuda(), _u8(T, inp['video']).cuda()
    grp = SptAudioGen(1, encoders=enc, separation='unet_mask', groups=G)
    grp.load_variables(P)
    plan = grp.autotune(a, v)
    assert len(plan) >= 30
    one = SptAudioGen(1, encoders=enc, separation='unet_mask')
    one.load_variables(P)
    one.inference_ops(a[:B], v[:B])
    names = SptAudioGen.tile_names()
    for layer, tile, sk, _ in plan:
        one.plan_set(B, layer, names.index(tile), sk)
    want = T.cat([one.inference_ops(a[g * B:(g + 1) * B], v[g * B:(g + 1) * B]) for g in range(G)], 0)
    assert T.equal(grp.inference_ops(a, v), want)
    tid = names.index('conv3h_kernel<128,64,64,32,1>')
    for net in (one, grp):
        for layer, _, _, _ in plan:
            if '_encoder/conv' in layer and 'video' in layer:
                net.plan_set(B, layer, tid, 3)
    want = T.cat([one.inference_ops(a[g * B:(g + 1) * B], v[g * B:(g + 1) * B]) for g in range(G)], 0)
    assert T.equal(grp.inference_ops(a, v), want)


def test_grouped_contexts_refuse_what_they_cannot_run(T):
    from spatialaudiogen_amd.model import SptAudioGen
    from spatialaudiogen_amd._lib import SagenError
    enc, B, G = ['audio', 'video'], 2, 2
    P = init_weights(variable_specs(enc), seed=13, mode='test')
    inp = synth_inputs(G * B, enc, seed=79)
    a, v = T.as_tensor(inp['audio']).cuda(), _u8(T, inp['video']).cuda()
    grp = SptAudioGen(1, encoders=enc, separation='unet_mask', groups=G)
    grp.load_variables(P)
    y = grp.inference_ops(a, v)
    with pytest.raises(ValueError):
        grp.inference_ops(a[:3], v[:3])                       # 3 windows are not 2 equal batches
    grp.set_option(B, 'fp16x2', 0)                            # bf16 planes: kernels without a group dimension
    with pytest.raises(SagenError, match='grouped'):
        grp.inference_ops(a, v)
    grp.set_option(B, 'fp16x2', 1)
    assert T.equal(grp.inference_ops(a, v), y)
    none = SptAudioGen(1, encoders=['audio'], separation='none', groups=2)
    none.load_variables(init_weights(none.variable_specs(), seed=1, mode='test'))
    with pytest.raises(SagenError, match='FREQ_MASK'):
        none.inference_ops(a)
