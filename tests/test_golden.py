"""Committed golden fixtures (tests/golden/golden_v1.npz, made by tools/make_golden.py from the fp64
oracle; inputs/weights are regenerated from seeds).  CPU: the oracle still reproduces them (pins the
restatement against silent edits).  GPU: the HIP path reproduces them to the parity bar."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import make_golden as G          # noqa: E402  (case table + probe list)
from util import rms             # noqa: E402

GOLD = np.load(os.path.join(ROOT, 'tests', 'golden', 'golden_v1.npz'))
# What TensorFlow 1.4 computed with the REFERENCE's own model.py on the same weights and inputs (tools/tf1_pin_inputs.py +
# tools/tf1_dump_golden.py on a TF1 machine; README "Pinning parity against TF1").  Absent in this image: parity is then UNPINNED
# against the reference itself and the tests below say so instead of passing vacuously.
TF1_FN = os.path.join(ROOT, 'tests', 'golden', 'tf1_v1.npz')
TF1 = np.load(TF1_FN) if os.path.exists(TF1_FN) else None
UNPINNED = 'parity UNPINNED against TF1: tests/golden/tf1_v1.npz absent (python2 tools/tf1_dump_golden.py <reference checkout> tf1_pin_inputs.npz tests/golden/tf1_v1.npz)'


@pytest.mark.parametrize('name', ['a_b2_s0', 'av_b2_s1'])
def test_oracle_reproduces_golden(name):
    out = G.run_case(name)
    for k, v in out.items():
        ref = GOLD[k]
        if k.endswith('/ambix'):
            assert rms(v.astype(np.float64) - ref) < 1e-6 * max(rms(ref), 1e-9), k
        else:
            assert np.allclose(v, ref, rtol=1e-7, atol=1e-9), k


def test_deploy_table_fixture():
    from oracle import np_oracle as O
    rows = np.array(O.deploy_window_table(O.audio_pow_times(12), 0., 10.), dtype=np.float64)
    assert np.array_equal(rows, GOLD['deploy_table_12s'])


@pytest.mark.gpu
@pytest.mark.parametrize('name', sorted(G.CASES))
def test_hip_path_reproduces_golden(name):
    import torch
    from spatialaudiogen_amd.model import SptAudioGen
    from spatialaudiogen_amd.weights import variable_specs, init_weights, synth_inputs
    assert torch.cuda.is_available()
    enc, B, ws, ins = G.CASES[name]
    P = init_weights(variable_specs(enc), seed=ws, mode='test')
    inp = synth_inputs(B, enc, seed=ins)
    net = SptAudioGen(1, encoders=enc, separation='unet_mask')
    net.load_variables(P)
    got = net.inference_ops(inp['audio'], inp.get('video'), inp.get('flow')).cpu().numpy()
    ref = GOLD[name + '/ambix']
    err = rms(got - ref)
    assert err <= 1e-4 and err <= 1e-3 * rms(ref), (name, err, rms(ref))


def tf1_pin_problems(tf1, cases=None):
    """Why a tf1_v1.npz cannot serve as the pin of the CURRENT case table ([] = it can): a case of tools/make_golden.py missing from
    it, or a case whose '<case>/fingerprint' (tools/tf1_pin_inputs.py: checksums of the weights and inputs the TF1 run was fed)
    differs from the one the current seeds / initialisers give."""
    import tf1_pin_inputs as K
    from spatialaudiogen_amd.weights import variable_specs, init_weights, synth_inputs
    bad = []
    for name, (enc, B, ws, ins) in sorted((cases or G.CASES).items()):
        if name + '/ambix' not in tf1 or name + '/fingerprint' not in tf1:
            bad.append('%s: no output / no fingerprint in the file' % name)
            continue
        want = K.fingerprint(init_weights(variable_specs(enc), seed=ws, mode='test'), synth_inputs(B, enc, seed=ins))
        got = np.asarray(tf1[name + '/fingerprint'], np.float64)
        if got.shape != want.shape or not np.allclose(got, want, rtol=1e-9, atol=1e-9):
            bad.append('%s: computed from other weights / inputs than the current case table gives' % name)
    return bad


def test_a_tf1_pin_that_is_present_must_not_be_stale():
    """tests/golden/tf1_v1.npz absent = parity UNPINNED (the tests below skip and say so).  PRESENT but made from another case table
    (seeds, initialisers, batch sizes changed since tools/tf1_pin_inputs.py wrote its inputs) = FAIL, never a silent skip or a
    comparison against outputs of other inputs."""
    if TF1 is None:
        pytest.skip(UNPINNED)
    bad = tf1_pin_problems(TF1)
    assert not bad, 'tests/golden/tf1_v1.npz is STALE against tools/tf1_pin_inputs.py: %s - rerun the pin kit' % '; '.join(bad)


def test_the_staleness_check_tells_a_current_pin_from_a_stale_one():
    """The check itself, on synthetic files (audio-only case: small): a pin whose fingerprint matches passes, one made with another
    weight seed, one without a fingerprint and one lacking the case are all reported."""
    import tf1_pin_inputs as K
    from spatialaudiogen_amd.weights import variable_specs, init_weights, synth_inputs
    name = 'a_b2_s0'
    enc, B, ws, ins = G.CASES[name]
    cases = {name: G.CASES[name]}
    amb = np.zeros((B, 4800, 3), np.float32)
    good = {name + '/ambix': amb, name + '/fingerprint': K.fingerprint(init_weights(variable_specs(enc), seed=ws, mode='test'), synth_inputs(B, enc, seed=ins))}
    assert tf1_pin_problems(good, cases) == []
    other = dict(good)
    other[name + '/fingerprint'] = K.fingerprint(init_weights(variable_specs(enc), seed=ws + 7, mode='test'), synth_inputs(B, enc, seed=ins))
    assert len(tf1_pin_problems(other, cases)) == 1 and 'other weights' in tf1_pin_problems(other, cases)[0]
    assert len(tf1_pin_problems({name + '/ambix': amb}, cases)) == 1
    assert len(tf1_pin_problems({}, cases)) == 1


@pytest.mark.parametrize('name', sorted(G.CASES))
def test_oracle_matches_what_tf1_computed(name):
    """The pin: the numpy oracle against the outputs of the reference's own graph (BASELINE.json: <= 1e-4 RMS vs TF1)."""
    if TF1 is None:
        pytest.skip(UNPINNED)
    assert not tf1_pin_problems(TF1, {name: G.CASES[name]}), 'stale tf1_v1.npz'
    out = G.run_case(name)
    ref = TF1[name + '/ambix'].astype(np.float64)
    err = rms(out[name + '/ambix'].astype(np.float64) - ref)
    assert err <= 1e-4 and err <= 1e-3 * rms(ref), (name, err, rms(ref))
    for p in ('video_encoder/conv5_2', 'flow_encoder/conv5_2', 'decoder/ambix'):
        k = '%s/chk/%s' % (name, p)
        if k in TF1.files and k in out:                     # [sum, sumsq, 16 probes]: the trunk output and the final mix
            scale = np.sqrt(TF1[k][1] / max(np.prod(TF1['%s/shape/%s' % (name, p)]), 1))
            assert np.allclose(out[k][2:], TF1[k][2:], atol=1e-3 * scale + 1e-6), (k, out[k][2:6], TF1[k][2:6])
            assert abs(out[k][1] - TF1[k][1]) <= 2e-3 * TF1[k][1] + 1e-9, k


@pytest.mark.gpu
@pytest.mark.parametrize('name', sorted(G.CASES))
def test_hip_path_matches_what_tf1_computed(name):
    if TF1 is None:
        pytest.skip(UNPINNED)
    assert not tf1_pin_problems(TF1, {name: G.CASES[name]}), 'stale tf1_v1.npz'
    import torch
    from spatialaudiogen_amd.model import SptAudioGen
    from spatialaudiogen_amd.weights import variable_specs, init_weights, synth_inputs
    assert torch.cuda.is_available()
    enc, B, ws, ins = G.CASES[name]
    P = init_weights(variable_specs(enc), seed=ws, mode='test')
    inp = synth_inputs(B, enc, seed=ins)
    net = SptAudioGen(1, encoders=enc, separation='unet_mask')
    net.load_variables(P)
    got = net.inference_ops(inp['audio'], inp.get('video'), inp.get('flow')).cpu().numpy()
    ref = TF1[name + '/ambix'].astype(np.float64)
    err = rms(got - ref)
    assert err <= 1e-4 and err <= 1e-3 * rms(ref), (name, err, rms(ref))


def test_pin_kit_inputs_carry_every_variable_under_its_checkpoint_name():
    """tools/tf1_pin_inputs.py (step 1 of the pin kit): every variable of a case under '<case>/var/<TF name>', the inputs under
    '<case>/in/*' - what tools/tf1_dump_golden.py assigns by name on the TF1 side (audio-only case: small)."""
    import tf1_pin_inputs as K
    from spatialaudiogen_amd.weights import variable_specs
    saved = dict(G.CASES)
    try:
        for k in list(G.CASES):
            if k != 'a_b2_s0':
                del G.CASES[k]
        d = K.build()
    finally:
        G.CASES.clear(); G.CASES.update(saved)
    specs = variable_specs(['audio'])
    for k, shape in specs.items():
        assert tuple(d['a_b2_s0/var/' + k].shape) == tuple(shape) and d['a_b2_s0/var/' + k].dtype == np.float32
    assert d['a_b2_s0/in/audio'].shape == (2, 52799, 1) and str(d['a_b2_s0/encoders']) == 'audio'
    assert sum(1 for k in d if k.startswith('a_b2_s0/var/')) == len(specs)
    assert d['a_b2_s0/fingerprint'].shape == (5,) and d['a_b2_s0/fingerprint'][0] == len(specs)
